"""Frame time of the reference's own scenes (tests/golden/scenes) at the golden-test profile scaled up
(1920x1080, 64 spp, 4 bounces), origin grids against KD-tree only:  python tools/scene_times.py"""
import sys, time
sys.path.insert(0, '.')
import os
os.environ.setdefault('PT_ESCAPE_AFTER', '0')   # (a measurement: the escape masks from the scene's first frame, not its third)
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
prof = pta.Profile.make(1920, 1080, 64, 4)
n = prof.width * prof.height
rgb = torch.empty(n * 3, dtype=torch.uint8, device='cuda'); acc = torch.empty(n * 3, dtype=torch.float32, device='cuda')
for name in ("cube", "reflection", "head", "spheres", "alpha_transparency", "white_furnace_indirect", "white_furnace_direct"):
    g = pta.GpuScene(pta.HostScene.load_isf(f"tests/golden/scenes/{name}/scene.isf"), 0)
    out = []
    for flags in (0, pta.PT_FLAG_NO_GRIDS):
        o = pta.Opts.make(flags=flags)
        for _ in range(3):   # (first frame of the configuration, the frame that allocates the planned queues, one more)
            g.render_device(prof, o, rgb.data_ptr(), acc.data_ptr(), 0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g.render_device(prof, o, rgb.data_ptr(), acc.data_ptr(), 0)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 3 * 1e3)
    i = g.info().as_dict()
    print(f"{name:24s} grids {out[0]:7.2f} ms ({n * 64 / out[0] / 1e3:7.0f} Msamples/s)   KD only {out[1]:7.2f} ms ({n * 64 / out[1] / 1e3:7.0f})   "
          f"x{out[1] / out[0]:.2f}   lights with grids {i['light_grids']}, translucent {i['has_translucent']}")

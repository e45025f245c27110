"""Per-launch stage times (PT_DEBUG_TIMES) of one frame of a generated scene:
    PT_DEBUG_TIMES=1 python tools/launch_times.py --tris 1000000 --flags 1 --width 3840 --height 2160 --spp 16 --bounces 8"""
import argparse, sys
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=500000); ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bounces", type=int, default=5); ap.add_argument("--flags", type=int, default=0)
a = ap.parse_args()
sc = pta.HostScene.generate_ps5(a.tris, 0, a.flags)
g = pta.GpuScene(sc, 0)
prof = pta.Profile.make(a.width, a.height, a.spp, a.bounces, "ACES")
n = a.width * a.height
rgb = torch.empty(n * 3, dtype=torch.uint8, device='cuda'); acc = torch.empty(n * 3, dtype=torch.float32, device='cuda')
for flags in (pta.PT_FLAG_COUNTERS, pta.PT_FLAG_TIMING):
    g.render_device(prof, pta.Opts.make(flags=flags), rgb.data_ptr(), acc.data_ptr(), 0); torch.cuda.synchronize()
    if flags == pta.PT_FLAG_COUNTERS:
        c = g.counters().as_dict(); print({k: c[k] for k in ("samples", "segments", "shadow_rays", "shaded_hits", "restarts", "rng_draws", "nodes_visited", "trace_nodes")}, file=sys.stderr)

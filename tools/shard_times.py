"""Per-rank frame time of an N-way tile-sharded render, emulated on one GPU (EVERY rank's shard, one after the other):
    python tools/shard_times.py [--spp 128] [--tris 500000] [--width 1920 --height 1080 --bounces 5 --tonemap FILMIC
                                 --scene-flags 0 --shards 1,2,4,8]
Prints ms per frame for every shard count and the strong-scaling efficiency those imply (all-gather not included):
a PROJECTION for the N-GPU job from one GPU, not an N-GPU measurement.
  config 3: defaults          config 4: --spp 512 --bounces 8
  config 5: --tris 4000000 --width 3840 --height 2160 --spp 1024 --bounces 8 --tonemap ACES --scene-flags 1
            (--spp 256 renders a quarter of the samples: the per-rank frame is linear in spp at that size)"""
import argparse, sys, time
sys.path.insert(0, '.')
import os
os.environ.setdefault('PT_ESCAPE_AFTER', '0')   # (a measurement: the escape masks from the scene's first frame, not its third)
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
ap = argparse.ArgumentParser()
ap.add_argument('--spp', type=int, default=128)
ap.add_argument('--tris', type=int, default=500000)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--width', type=int, default=1920)
ap.add_argument('--height', type=int, default=1080)
ap.add_argument('--bounces', type=int, default=5)
ap.add_argument('--tonemap', default='FILMIC')
ap.add_argument('--scene-flags', type=int, default=8)
ap.add_argument('--shards', default='1,2,4,8')
a = ap.parse_args()
sc = pta.HostScene.generate_ps5(a.tris, 0, a.scene_flags)
g = pta.GpuScene(sc, 0)
prof = pta.Profile.make(a.width, a.height, a.spp, a.bounces, a.tonemap)
print(f"workload: {sc.n_triangles} triangles (flags {a.scene_flags}), {a.width}x{a.height}, {a.spp} spp, {a.bounces} bounces, {a.tonemap}")
base = None
for n in [int(x) for x in a.shards.split(',')]:
    worst = 0.0
    for r in range(n):
        opts = pta.Opts.make(flags=pta.PT_FLAG_TIMING, shard_rank=r, shard_count=n, tile_w=32, tile_h=32)
        npx = len(pta.local_pixel_map(prof, opts))
        rgb = torch.empty(npx * 3, dtype=torch.uint8, device='cuda')
        acc = torch.empty(npx * 3, dtype=torch.float32, device='cuda')
        for _ in range(2):   # (the first frame of the configuration counts, the second allocates the planned queues)
            g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        worst = max(worst, ms)
        t = g.timing().as_dict()
    base = base or worst * n   # (the first shard count listed is the reference; normally 1)
    print(f"shards {n}: {worst:8.3f} ms per frame (slowest rank)  efficiency {base / (n * worst):.3f}  "
          f"stages {{'rng': {t['generate_ms']:.2f}, 'trace': {t['trace_ms']:.2f}, 'shade': {t['shade_ms']:.2f}, 'shadow': {t['shadow_ms']:.2f}}} launches {t['stage_launches']}")

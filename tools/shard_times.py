"""Per-rank frame time of an N-way tile-sharded render, emulated on one GPU (EVERY rank's shard, one after the other):
    python tools/shard_times.py [--spp 128] [--tris 500000]
Prints ms per frame for shard_count = 1, 2, 4, 8 and the strong-scaling efficiency those imply
(all-gather not included)."""
import argparse, sys, time
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
ap = argparse.ArgumentParser()
ap.add_argument('--spp', type=int, default=128)
ap.add_argument('--tris', type=int, default=500000)
ap.add_argument('--reps', type=int, default=3)
a = ap.parse_args()
sc = pta.HostScene.generate_ps5(a.tris, 0)
g = pta.GpuScene(sc, 0)
prof = pta.Profile.make(1920, 1080, a.spp, 5, "FILMIC")
base = None
for n in (1, 2, 4, 8):
    worst = 0.0
    for r in range(n):
        opts = pta.Opts.make(flags=pta.PT_FLAG_TIMING, shard_rank=r, shard_count=n, tile_w=32, tile_h=32)
        npx = len(pta.local_pixel_map(prof, opts))
        rgb = torch.empty(npx * 3, dtype=torch.uint8, device='cuda')
        acc = torch.empty(npx * 3, dtype=torch.float32, device='cuda')
        g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.reps * 1e3
        worst = max(worst, ms)
        t = g.timing().as_dict()
    base = base or worst
    print(f"shards {n}: {worst:8.3f} ms per frame (slowest rank)  efficiency {base / (n * worst):.3f}  "
          f"stages {{'rng': {t['generate_ms']:.2f}, 'trace': {t['trace_ms']:.2f}, 'shade': {t['shade_ms']:.2f}, 'shadow': {t['shadow_ms']:.2f}}} launches {t['stage_launches']}")

"""Per-launch stage times (PT_DEBUG_TIMES) of one shard of an N-way tile-sharded config-3 frame:
    PT_DEBUG_TIMES=1 python tools/shard_launches.py --shards 8 [--rank 0]"""
import argparse, sys
sys.path.insert(0, '.')
import os
os.environ.setdefault('PT_ESCAPE_AFTER', '0')   # (a measurement: the escape masks from the scene's first frame, not its third)
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
ap = argparse.ArgumentParser()
ap.add_argument('--shards', type=int, default=8); ap.add_argument('--rank', type=int, default=0)
ap.add_argument('--spp', type=int, default=128); ap.add_argument('--tris', type=int, default=500000)
a = ap.parse_args()
sc = pta.HostScene.generate_ps5(a.tris, 0)
g = pta.GpuScene(sc, 0)
prof = pta.Profile.make(1920, 1080, a.spp, 5, "FILMIC")
opts = pta.Opts.make(flags=pta.PT_FLAG_TIMING, shard_rank=a.rank, shard_count=a.shards, tile_w=32, tile_h=32)
npx = len(pta.local_pixel_map(prof, opts))
rgb = torch.empty(npx * 3, dtype=torch.uint8, device='cuda'); acc = torch.empty(npx * 3, dtype=torch.float32, device='cuda')
for _ in range(2):
    print("---- frame", file=sys.stderr)
    g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0)
print(g.timing().as_dict())
g.render_device(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS, shard_rank=a.rank, shard_count=a.shards, tile_w=32, tile_h=32), rgb.data_ptr(), acc.data_ptr(), 0)
torch.cuda.synchronize()
print(g.counters().as_dict())

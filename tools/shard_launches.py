"""Per-launch stage times (PT_DEBUG_TIMES) of one frame: python tools/shard_launches.py <shards> [spp]"""
import sys, os
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
n = int(sys.argv[1]); spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sc = pta.HostScene.generate_ps5(500000, 0)
g = pta.GpuScene(sc, 0)
prof = pta.Profile.make(1920, 1080, spp, 5, "FILMIC")
opts = pta.Opts.make(flags=pta.PT_FLAG_TIMING, shard_rank=0, shard_count=n, tile_w=32, tile_h=32)
npx = len(pta.local_pixel_map(prof, opts))
rgb = torch.empty(npx * 3, dtype=torch.uint8, device='cuda'); acc = torch.empty(npx * 3, dtype=torch.float32, device='cuda')
for _ in range(2):
    g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0); torch.cuda.synchronize()
print("items", npx * spp, file=sys.stderr)

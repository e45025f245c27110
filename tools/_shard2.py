import sys, time
sys.path.insert(0, '.')
import torch; torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
sc = pta.HostScene.generate_ps5(500000, 0); g = pta.GpuScene(sc, 0)
prof = pta.Profile.make(1920, 1080, 128, 5, "FILMIC")
for n, tile in ((1, 32), (2, 32), (2, 16), (2, 64), (4, 32), (8, 32), (8, 16), (8, 64)):
    res = []
    for r in range(n):
        opts = pta.Opts.make(flags=pta.PT_FLAG_TIMING, shard_rank=r, shard_count=n, tile_w=tile, tile_h=tile)
        npx = len(pta.local_pixel_map(prof, opts))
        rgb = torch.empty(npx * 3, dtype=torch.uint8, device='cuda'); acc = torch.empty(npx * 3, dtype=torch.float32, device='cuda')
        g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 3 * 1e3)
    print(f"shards {n} tile {tile}: " + " ".join(f"{v:.2f}" for v in res) + f"  max {max(res):.2f} sum {sum(res):.2f}")

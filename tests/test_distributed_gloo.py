"""N > 1 path on CPU: world_size-2 (and 3) gloo ranks shard the frame with the product's tile map
(pt_local_pixel_map — host code, no GPU), each rank RENDERS ITS OWN SHARD with the oracle (tile row by tile row,
global pixel index in the seed), all-gathers the padded u8 slices exactly as bench.py does, and rebuilds the
row-major image.  It must equal the unsharded render."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys, ctypes as C
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, sys.argv[1])
    import __graft_entry__ as entry
    pta, oracle = entry.load_package(), entry.load_oracle()
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    scene = pta.HostScene.load_isf(os.path.join(sys.argv[1], "tests/golden/scenes/alpha_transparency/scene.isf"))
    prof = pta.Profile.make(150, 70, 3, 2)
    tile = 16
    opts = pta.Opts.make(shard_rank=rank, shard_count=world, tile_w=tile, tile_h=tile)
    idx = pta.local_pixel_map(prof, opts)
    osc = oracle.OracleScene(scene.desc, oracle.PTO_BVH)
    # this rank's pixels, rendered by this rank only: the packed order is tile after tile, row by row inside a tile,
    # so every run of consecutive global indices (one tile row) is one oracle call over [begin, end)
    local_rgb = np.zeros((len(idx), 3), np.uint8)
    breaks = np.nonzero(np.diff(idx.astype(np.int64)) != 1)[0] + 1
    n_runs = 0
    for run in np.split(np.arange(len(idx)), breaks):
        begin, end = int(idx[run[0]]), int(idx[run[-1]]) + 1
        rgb_run, _, _ = osc.render(prof, begin, end, 1)
        local_rgb[run] = rgb_run
        n_runs += 1
    assert n_runs >= len(idx) // tile                   # (really tile rows, not one big range)
    full_rgb, _, _ = osc.render(prof)                   # the unsharded truth, to compare the assembled frame with
    n_local = torch.tensor([len(idx)])
    dist.all_reduce(n_local, op=dist.ReduceOp.MAX)
    slice_pixels = int(n_local.item())
    padded = torch.zeros(slice_pixels * 3, dtype=torch.uint8)
    padded[: len(idx) * 3] = torch.from_numpy(local_rgb.reshape(-1))
    gathered = torch.zeros(world * slice_pixels * 3, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, padded)
    image = np.zeros((prof.width * prof.height, 3), np.uint8)
    g = gathered.numpy().reshape(world, slice_pixels, 3)
    for r in range(world):
        ridx = pta.local_pixel_map(prof, pta.Opts.make(shard_rank=r, shard_count=world, tile_w=tile, tile_h=tile))
        image[ridx] = g[r, : len(ridx)]
    assert np.array_equal(image, full_rgb), "assembled image differs from the unsharded render"
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
''')


@pytest.mark.parametrize("world", [2, 3])
def test_tile_sharded_render_with_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world), str(script), str(ROOT)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert res.stdout.count("ok") == world

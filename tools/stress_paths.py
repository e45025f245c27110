"""Randomised differential test of the integrator paths on the GPU: generated scenes (every flag combination the generator
knows, several sizes and seeds) x random profiles (odd sizes, 1-9 samples, 0-7 bounces, every tone map) x random options
(shards, tile sizes, sample batches, queue budgets): the default pipeline (origin grids, camera-grid cull, split shade pass, hand-over
kernel) must equal the KD-tree pipeline and the one-lane-per-pixel megakernel bit for bit, image and f32 accumulation;
so must the second frame of each pipeline, whose queues are sized by the first frame's counts (frame plan).
    python tools/stress_paths.py [seconds] [seed]"""
import os, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
os.environ.setdefault("PT_ESCAPE_AFTER", "0")   # (escape masks from a scene's first frame: the cases render two frames per pipeline)
import __graft_entry__ as e
pta = e.load_package()
orc = e.load_oracle()


def run(budget, seed, max_cases=None, verbose=True):
    """Returns (cases, cases with empty blocks, cases also checked against the oracle); raises AssertionError on a mismatch."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    cases = culled = with_oracle = planned = 0
    last = t0
    while time.time() - t0 < budget and (max_cases is None or cases < max_cases):
        if verbose and time.time() - last > 30:   # (a run on the GPU box must not stay silent)
            print(f"... {cases} cases after {time.time() - t0:.0f} s", flush=True)
            last = time.time()
        tris = int(rng.choice([600, 2000, 9000, 40000, 150000]))
        flags = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7]))
        seed_s = int(rng.integers(0, 1000))
        host_scene = pta.HostScene.generate_ps5(tris, seed_s, flags)
        g = pta.GpuScene(host_scene, 0)
        osc = None
        for _ in range(4):
            w, h = int(rng.integers(9, 700)), int(rng.integers(9, 400))
            prof = pta.Profile.make(w, h, int(rng.integers(1, 10)), int(rng.integers(0, 8)), str(rng.choice(["REINHARD", "FILMIC", "ACES"])))
            count = int(rng.choice([1, 1, 2, 3, 5]))
            rank = int(rng.integers(0, count))
            tile_w, tile_h = [(32, 32), (16, 16), (64, 8), (48, 16), (24, 32), (8, 32)][int(rng.integers(0, 6))]   # (tile_w * tile_h: a multiple of 256)
            kw = dict(shard_rank=rank, shard_count=count, tile_w=tile_w, tile_h=tile_h, sample_batch=int(rng.choice([0, 0, 1, 3])))
            # queue budgets of the first / the later frames: the defaults, or so little that a frame takes several chunks of 1 Mi items
            budgets = [(None, None), ("0.001", None), ("0.001", "0.001"), (None, "0.001")][int(rng.integers(0, 4))]
            for name, val in zip(("PT_QUEUE_GIB", "PT_QUEUE_STEADY_GIB", "PT_QUEUE_ONE_PASS_GIB"), budgets + (budgets[1],)):
                if val is None:
                    os.environ.pop(name, None)
                else:
                    os.environ[name] = val
            what = dict(tris=tris, flags=flags, seed=seed_s, w=w, h=h, spp=prof.samples, bounces=prof.bounces, budgets=budgets, **kw)
            rgb, acc = g.render(prof, pta.Opts.make(**kw))
            blocks, empty = g.cull_stats()
            culled += empty > 0
            for f in (pta.PT_FLAG_NO_GRIDS, pta.PT_FLAG_MEGAKERNEL):
                rgb2, acc2 = g.render(prof, pta.Opts.make(flags=f, **kw))
                same = np.array_equal(acc.view(np.uint32), acc2.view(np.uint32)) and np.array_equal(rgb, rgb2)
                assert same, ("MISMATCH between the default pipeline and path", f, what,
                              np.flatnonzero((acc.view(np.uint32) != acc2.view(np.uint32)).reshape(len(acc), -1).any(1))[:10])
            # the second frame of a configuration: queues as long as the first frame's counts say
            for f in (0, pta.PT_FLAG_NO_GRIDS):
                rgb2, acc2 = g.render(prof, pta.Opts.make(flags=f, **kw))
                same = np.array_equal(acc.view(np.uint32), acc2.view(np.uint32)) and np.array_equal(rgb, rgb2)
                assert same, ("MISMATCH between the first and the second frame of path", f, what,
                              np.flatnonzero((acc.view(np.uint32) != acc2.view(np.uint32)).reshape(max(1, len(acc)), -1).any(1))[:10])
                # (not planned: an empty shard renders nothing; a scene that builds its escape masks for this frame counts again)
                planned += int(g.info().frame_planned) if len(acc) else 0
            # small whole frames also against the CPU oracle (the restatement of the reference, pinned by its goldens)
            if count == 1 and w * h * prof.samples <= 60000:
                osc = osc or orc.OracleScene(host_scene.desc, orc.PTO_BVH)
                o_rgb, o_acc, _ = osc.render(prof)
                same = np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)) and np.array_equal(rgb, o_rgb)
                assert same, ("MISMATCH between the GPU and the oracle", what,
                              np.flatnonzero((acc.view(np.uint32) != o_acc.view(np.uint32)).reshape(len(acc), -1).any(1))[:10])
                with_oracle += 1
            cases += 1
    for name in ("PT_QUEUE_GIB", "PT_QUEUE_STEADY_GIB", "PT_QUEUE_ONE_PASS_GIB"):
        os.environ.pop(name, None)
    if verbose:
        print(f"{planned} second frames ran with planned queues")
    return cases, culled, with_oracle, time.time() - t0


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    try:
        cases, culled, with_oracle, secs = run(budget, seed)
    except AssertionError as err:
        print(*err.args[0] if isinstance(err.args[0], tuple) else err.args)
        sys.exit(1)
    print(f"{with_oracle} of the cases also equal to the CPU oracle's frame")
    print(f"{cases} cases in {secs:.0f} s, all three paths bit-identical; the camera-grid cull found empty blocks in {culled} of them")

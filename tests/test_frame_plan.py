"""The frame plan (csrc/pt_gpu.hip, render_device): queues sized by what a frame is known to produce.

A frame is a pure function of (scene, profile, options) - the reference seeds every pixel's generator from the pixel's
index (renderer/mod.rs:110-112) - so the number of records each bounce puts into each queue is the same in every frame of
a configuration.  The first frame runs in chunks of a fixed budget with every queue as long as the chunk and counts; the
later ones get queues of exactly the counted lengths (the reference's whole working state is one W*H*12 B buffer,
renderer/mod.rs:76-81).  The bits must not depend on any of it.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name,flags", [("cube", 0), ("alpha_transparency", 0), ("head", 0), ("head", 4)])
@pytest.mark.parametrize("steady_gib", ["0.02", "16"])
def test_planned_frames_render_the_same_bits_in_less_memory(pta, scene_cache, monkeypatch, name, flags, steady_gib):
    scene = scene_cache(name)
    prof = pta.Profile.make(640, 360, 16, 4)             # 3.7 M work items
    opts = pta.Opts.make(flags=flags)
    rgb, acc = pta.GpuScene(scene).render(prof, opts)    # default budgets: one chunk
    monkeypatch.setenv("PT_QUEUE_GIB", "0.01")           # first frame: chunks of 1 Mi work items (the minimum)
    monkeypatch.setenv("PT_QUEUE_STEADY_GIB", steady_gib)
    monkeypatch.setenv("PT_QUEUE_ONE_PASS_GIB", steady_gib)   # (a frame that fits this in one pass may take it: default 32)
    g = pta.GpuScene(scene)
    seen = []
    for frame in range(4):
        rgb2, acc2 = g.render(prof, opts)
        assert np.array_equal(rgb2, rgb) and np.array_equal(bits(acc2), bits(acc)), frame
        seen.append(g.info().as_dict())
    first, second, last = seen[0], seen[1], seen[-1]
    assert first["frame_planned"] == 0 and first["queue_chunk_items"] == 1 << 20
    assert second["frame_planned"] == 1 and last["frame_planned"] == 1
    assert last["queue_bytes"] == second["queue_bytes"]
    if steady_gib == "16":
        assert second["queue_chunk_items"] == 640 * 384 * 16          # the whole frame (32 x 32 tiles) in one pass
    else:
        assert second["queue_chunk_items"] % (1 << 20) == 0 and second["queue_chunk_items"] < 640 * 384 * 16
    # bounce 0 of the fused pipeline uses no queue: one 16-byte RNG plane per work item + the survivors' records
    per_item = [i["queue_bytes"] / i["queue_chunk_items"] for i in (first, second)]
    # (buffers within 64 MiB of what the plan needs are kept: the small chunks of the 0.02 GiB case show nothing)
    assert per_item[0] >= 256 and (per_item[1] < per_item[0] - 32 if steady_gib == "16" else per_item[1] <= per_item[0]), per_item


def test_two_configurations_share_one_scene(pta, scene_cache, monkeypatch):
    """Plans are kept per configuration; the buffers are the scene's and only grow once every plan is made."""
    scene = scene_cache("reflection")
    g = pta.GpuScene(scene)
    a, b = pta.Profile.make(320, 200, 8, 3), pta.Profile.make(640, 360, 4, 5)
    ref_a, ref_b = g.render(a), g.render(b)
    sizes = []
    for _ in range(3):
        for prof, ref in ((a, ref_a), (b, ref_b)):
            rgb, acc = g.render(prof)
            assert np.array_equal(rgb, ref[0]) and np.array_equal(bits(acc), bits(ref[1]))
            sizes.append(g.info().as_dict()["queue_bytes"])
    assert g.info().as_dict()["frame_planned"] == 1
    assert sizes[-1] == sizes[-2] == sizes[-3]


def test_frames_in_flight_do_not_wait_for_the_counts(pta, scene_cache):
    """pt_render_device is asynchronous: frames issued before the first frame's counts have arrived run like a first frame."""
    import torch
    scene = scene_cache("cube")
    prof = pta.Profile.make(640, 360, 16, 4)
    rgb, acc = pta.GpuScene(scene).render(prof)
    g = pta.GpuScene(scene)
    n = 640 * 360
    outs = [(torch.empty(n * 3, dtype=torch.uint8, device="cuda"), torch.empty(n * 3, dtype=torch.float32, device="cuda"))
            for _ in range(6)]
    opts = pta.Opts.make()
    stream = torch.cuda.current_stream().cuda_stream
    for k, (r, a) in enumerate(outs):
        g.render_device(prof, opts, r.data_ptr(), a.data_ptr(), stream)
        if k == 3:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    assert g.info().as_dict()["frame_planned"] == 1
    for r, a in outs:
        assert np.array_equal(r.cpu().numpy().reshape(-1, 3), rgb)
        assert np.array_equal(bits(a.cpu().numpy().reshape(-1, 3)), bits(acc))


def test_escape_masks_arrive_with_the_third_frame(pta, monkeypatch):
    """A scene builds its escape masks when it is about to render its third frame of the default pipeline (a one-shot render is
    better off without them); the frame after counts again, the one after that is planned - same bits throughout."""
    monkeypatch.setenv("PT_ESCAPE_AFTER", "2")
    host = pta.HostScene.generate_ps5(60000, 3, 8)
    prof = pta.Profile.make(480, 270, 8, 4)
    rgb, acc = pta.GpuScene(host).render(prof, pta.Opts.make(flags=pta.PT_FLAG_NO_GRIDS))
    g = pta.GpuScene(host)
    seen = []
    for frame in range(5):
        rgb2, acc2 = g.render(prof)
        assert np.array_equal(rgb2, rgb) and np.array_equal(bits(acc2), bits(acc)), frame
        seen.append(g.info().as_dict())
    assert [i["escape_prims"] > 0 for i in seen] == [False, False, True, True, True]
    assert [i["frame_planned"] for i in seen] == [0, 1, 0, 1, 1]


@pytest.mark.parametrize("name,flags", [("head", 0), ("alpha_transparency", 0), ("head", 4)])
def test_a_wrong_plan_is_reported_not_rendered_silently(pta, scene_cache, monkeypatch, name, flags):
    """What cannot happen (a queue sized from an earlier frame's counts runs full) made to happen: PT_PLAN_TEST_SHRINK cuts every
    planned array to 60 %.  The frame must complete without touching memory it does not own, the NEXT call of the configuration
    must fail loudly, and the one after - counted again - is the right frame."""
    scene = scene_cache(name)
    prof = pta.Profile.make(640, 360, 16, 4)
    opts = pta.Opts.make(flags=flags)
    g = pta.GpuScene(scene)
    rgb, acc = g.render(prof, opts)
    g.render(prof, opts)                                  # planned
    monkeypatch.setenv("PT_PLAN_TEST_SHRINK", "0.6")
    g.render(prof, opts)                                  # arrays too short: records dropped, frame flagged (image not checked)
    monkeypatch.delenv("PT_PLAN_TEST_SHRINK")
    with pytest.raises(pta.PtError, match="ran full"):
        g.render(prof, opts)
    rgb2, acc2 = g.render(prof, opts)                     # first-frame mode again
    assert np.array_equal(rgb2, rgb) and np.array_equal(bits(acc2), bits(acc))
    rgb3, acc3 = g.render(prof, opts)                     # and planned again
    assert np.array_equal(rgb3, rgb) and np.array_equal(bits(acc3), bits(acc)) and g.info().frame_planned == 1

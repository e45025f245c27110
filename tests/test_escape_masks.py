"""Escape masks (csrc/pt_escape.h): per primitive a cube map of directions (8 x 8 cells per face) whose clear bits PROVE that a
ray leaving the primitive in that cell hits nothing (the reference casts it and gets an empty list, renderer/mod.rs:180-186).
The masks are a filter in front of the reference's arithmetic; these tests check that they are conservative - rays aimed
through clear cells, from every place an origin can be, against the brute-force oracle and the exact scalar walker - and that
the frame does not change by a bit when they are switched off."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
H_LO, H_HI = 5e-6, 1e-3


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rays_through_clear_cells(scene, masks, n, seed):
    """n rays: a primitive with a mask (weighted by its number of clear cells), a clear cell of it, a direction inside the
    cell, an origin above a random point of the triangle - foot slightly outside included - at a height in [H_LO, H_HI]."""
    normals, v0, blocked = masks
    rng = np.random.default_rng(seed)
    d = scene.desc.contents
    tris = np.ctypeslib.as_array(d.triangles, (int(d.n_triangles) * 24,)).reshape(-1, 3, 8)[:, :, :3]
    # primitive ids: one per triangle and per sphere in model order - map triangle -> primitive
    prim_of_tri = np.zeros(len(tris), np.int64)
    q = 0
    for m in range(int(d.n_models)):
        mo = d.models[m]
        if mo.kind == 0:   # mesh
            prim_of_tri[mo.tri_first:mo.tri_first + mo.tri_count] = np.arange(q, q + mo.tri_count)
            q += mo.tri_count
        else:
            q += 1
    tri_of_prim = -np.ones(q, np.int64)
    tri_of_prim[prim_of_tri] = np.arange(len(tris))
    clear_count = (~blocked).reshape(len(blocked), -1).sum(axis=1) * (np.abs(normals).sum(axis=1) > 0)
    if clear_count.sum() == 0:
        return None, None
    prims = rng.choice(len(clear_count), n, p=clear_count / clear_count.sum())
    # a clear cell of each
    cells = np.array([rng.choice(np.nonzero(~blocked[p].reshape(-1))[0]) for p in prims])
    face, cv, cu = cells // 64, (cells % 64) // 8, cells % 8
    u = (cu + rng.uniform(0.0, 1.0, n)) * 0.25 - 1.0
    v = (cv + rng.uniform(0.0, 1.0, n)) * 0.25 - 1.0
    axis, neg = face // 2, face % 2
    dirs = np.zeros((n, 3))
    i = np.arange(n)
    dirs[i, axis] = np.where(neg == 1, -1.0, 1.0)
    dirs[i, (axis + 1) % 3] = u
    dirs[i, (axis + 2) % 3] = v
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    t = tris[tri_of_prim[prims]]
    b = rng.random((n, 2))
    s = np.sqrt(b[:, 0])
    w = np.stack([1 - s, s * (1 - b[:, 1]), s * b[:, 1]], axis=1)
    w = w * rng.uniform(1.0, 1.002, (n, 1)) - rng.uniform(0.0, 0.0007, (n, 3))   # feet a little outside the triangle too
    foot = (t * w[:, :, None]).sum(axis=1)
    height = np.where(rng.random(n) < 0.7, rng.uniform(H_LO * 1.01, 3e-5, n), rng.uniform(H_LO * 1.01, H_HI * 0.99, n))
    o = foot + normals[prims] * height[:, None]
    return np.concatenate([o, dirs], axis=1).astype(np.float32), prims


@pytest.mark.parametrize("what", ["gen8", "gen0", "gen12", "cube", "reflection", "head", "alpha_transparency", "white_furnace_direct"])
def test_rays_through_clear_cells_hit_nothing(pta, oracle, scene_cache, what):
    if what.startswith("gen"):
        scene = pta.HostScene.generate_ps5(60000, seed=3, flags=int(what[3:]))
    else:
        scene = scene_cache(what)
    g = pta.GpuScene(scene)
    masks = g.escape_masks()    # (builds them if the scene has not rendered yet)
    info = g.info()
    assert info.escape_prims == int((np.abs(masks[0]).sum(axis=1) > 0).sum())
    rays, prims = rays_through_clear_cells(scene, masks, 60000, seed=9)
    if rays is None:
        pytest.skip("no primitive of this scene has a mask with a clear cell")
    # the heights the float32 origin really has (the kernel checks exactly this number)
    h = ((rays[:, :3] - masks[1][prims]) * masks[0][prims]).sum(axis=1)
    ok = (h >= H_LO) & (h <= H_HI)
    hits, counts = g.trace_all(rays[ok], 2)                       # the exact scalar walker (kd_traverse)
    assert counts.sum() == 0, (what, int(counts.sum()), rays[ok][counts > 0][:3], hits[counts > 0][:3])
    sub = rays[ok][:4000]
    o_hits, o_counts = oracle.OracleScene(scene.desc, oracle.PTO_BRUTE_FORCE if scene.n_prims < 5000 else oracle.PTO_BVH).trace_all(sub, 2)
    assert o_counts.sum() == 0


def test_the_ground_of_the_stand_in_is_covered(pta):
    """What the masks are for: on the benchmark scene the ground's triangles have masks, and most of the sky above them is clear."""
    scene = pta.HostScene.generate_ps5(500000, seed=0, flags=8)   # (the benchmark scene: its ground cells are 0.21 wide)
    g = pta.GpuScene(scene)
    normals, v0, blocked = g.escape_masks()
    m0 = scene.desc.contents.models[0]
    ground = slice(m0.tri_first, m0.tri_first + m0.tri_count)
    has = np.abs(normals[ground]).sum(axis=1) > 0
    assert has.mean() > 0.95
    clear = (~blocked[ground][has]).reshape(has.sum(), -1).mean()
    assert clear > 0.25          # (the lower hemisphere and the grazing band are blocked: at most ~0.45 can be clear)
    prof = pta.Profile.make(480, 270, 4, 5, "FILMIC")
    g.render(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS))
    c = g.counters().as_dict()
    assert c["masked_casts"] > 0.35 * (c["segments"] - c["samples"])   # (config 3: 45 % of the casts of bounces >= 1)


@pytest.mark.parametrize("flags", [8, 0, 9, 10, 12])
def test_frames_with_and_without_masks_are_the_same_bits(pta, flags):
    """The masked pipeline against itself without masks (PT_ESCAPE=0 at scene creation), the KD-tree pipeline and the megakernel
    (which never consult a mask), and the masks did remove casts."""
    scene = pta.HostScene.generate_ps5(40000, seed=2, flags=flags)
    prof = pta.Profile.make(320, 180, 6, 5, "FILMIC")
    g = pta.GpuScene(scene)
    rgb, acc = g.render(prof)
    g.render(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS))
    c = g.counters().as_dict()
    os.environ["PT_ESCAPE"] = "0"
    try:
        g0 = pta.GpuScene(scene)
    finally:
        del os.environ["PT_ESCAPE"]
    assert g0.info().escape_prims == 0
    rgb0, acc0 = g0.render(prof)
    assert np.array_equal(bits(acc), bits(acc0)) and np.array_equal(rgb, rgb0)
    g0.render(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS))
    c0 = g0.counters().as_dict()
    assert c0["masked_casts"] == 0 and c["segments"] == c0["segments"] and c["shadow_rays"] == c0["shadow_rays"]
    if flags in (8, 0, 9, 10):
        assert c["masked_casts"] > 0      # (a 40 000-triangle stand-in has a coarse ground: few cells can be cleared)
    for f in (pta.PT_FLAG_NO_GRIDS, pta.PT_FLAG_MEGAKERNEL):
        rgb2, acc2 = g.render(prof, pta.Opts.make(flags=f))
        assert np.array_equal(bits(acc), bits(acc2)) and np.array_equal(rgb, rgb2), f


def test_non_black_background_is_added_in_the_reference_order(pta, oracle, scene_cache):
    """spheres / white_furnace scenes have a non-black background: a masked miss adds throughput x background AFTER the bounce's
    lights (renderer/mod.rs:184-186 follows :248-262) - only where the lights are added in the same kernel; the frame against the oracle."""
    for name in ("cube", "reflection", "alpha_transparency"):
        scene = scene_cache(name)
        d = scene.desc.contents
        old = [d.background[k] for k in range(3)]
        try:
            for k, v in enumerate((0.3, 0.5, 0.9)):
                d.background[k] = v
            g = pta.GpuScene(scene)
            prof = pta.Profile.make(160, 120, 8, 4, "FILMIC")
            rgb, acc = g.render(prof)
            o_rgb, o_acc, _ = oracle.OracleScene(scene.desc, oracle.PTO_BVH).render(prof)
            assert np.array_equal(bits(acc), bits(o_acc)) and np.array_equal(rgb, o_rgb), name
        finally:
            for k in range(3):
                d.background[k] = old[k]

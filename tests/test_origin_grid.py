"""Origin grids (host/origin_grid.cpp) are a candidate filter like kdtree-ray: they may list too much, never too
little.  CPU-only check against the oracle's literal brute-force ray_cast(): for rays through the grid's origin,
every primitive of the sorted hit list must sit in the looked-up cell (or the global block), and its stored
distance must not exceed the distance of the hit."""
import ctypes as C

import numpy as np
import pytest

from conftest import SCENES

MESH_SCENES = ["cube", "reflection", "head", "alpha_transparency", "white_furnace_direct"]
SPHERE_SCENES = ["spheres", "white_furnace_indirect"]


def camera_origin(scene):
    t = scene.desc.contents.camera.transform
    return np.array([t[12], t[13], t[14]], np.float32)


def primary_rays(osc, prof, n, seed):
    rng = np.random.default_rng(seed)
    pix = rng.integers(0, prof.width * prof.height, n)
    smp = rng.integers(1, prof.samples + 1, n)
    return np.stack([osc.primary_ray(prof, int(p), int(s)) for p, s in zip(pix, smp)])


def check_rays(grid, rays, lookup_dirs, hits, counts, max_dist=None, origin=None):
    """hits/counts: brute-force sorted hit lists of `rays`; every hit (within max_dist of the grid origin when given)
    must be a candidate of cell(lookup_dir)."""
    cells = grid.cells(lookup_dirs)
    checked = 0
    for i in range(len(rays)):
        if counts[i] == 0:
            continue
        prims, mind = grid.candidates(int(cells[i]))
        ids = prims & 0x7fffffff
        o, d = rays[i, :3].astype(np.float64), rays[i, 3:].astype(np.float64)
        for k in range(min(int(counts[i]), hits.shape[1])):
            h = hits[i, k]
            sphere = bool(h["flags"] & 2)
            # hit point: triangles are o + d * dist (triangle.rs:77); spheres report the Euclidean distance
            x = o + d * (float(h["dist"]) / (np.linalg.norm(d) if sphere else 1.0))
            dist_o = float(np.linalg.norm(x - origin.astype(np.float64)))
            if max_dist is not None and dist_o > max_dist[i]:
                continue
            sel = np.nonzero(ids == int(h["prim"]))[0]
            assert len(sel) > 0, f"ray {i}: primitive {int(h['prim'])} hit by brute force is not in cell {int(cells[i])}"
            assert float(mind[sel[0]]) <= dist_o * (1 + 1e-6), (i, int(h["prim"]), float(mind[sel[0]]), dist_o)
            checked += 1
    return checked


@pytest.mark.parametrize("name", MESH_SCENES + SPHERE_SCENES)
@pytest.mark.parametrize("res", [0, 256])
def test_camera_grid_is_conservative(pta, oracle, scene_cache, name, res):
    scene = scene_cache(name)
    osc = oracle.OracleScene(scene.desc, oracle.PTO_BRUTE_FORCE)
    prof = pta.Profile.make(256, 256, 4, 2)
    rays = primary_rays(osc, prof, 1500, seed=11)
    grid = pta.OriginGrid(scene, camera_origin(scene), res)
    assert grid.enabled
    hits, counts = osc.trace_all(rays, 16)
    assert check_rays(grid, rays, rays[:, 3:], hits, counts, origin=camera_origin(scene)) > 300


@pytest.mark.parametrize("name", ["cube", "reflection", "head", "spheres"])
def test_light_grid_is_conservative(pta, oracle, scene_cache, name):
    """Shadow rays of get_light_info (mod.rs:301-331): origin = hit + n * 1e-5 (|n| up to 1.5 here), direction =
    -normalize(hit - light); the cell is looked up with (hit - light)."""
    scene = scene_cache(name)
    osc = oracle.OracleScene(scene.desc, oracle.PTO_BRUTE_FORCE)
    prof = pta.Profile.make(256, 256, 4, 2)
    rays = primary_rays(osc, prof, 1500, seed=12)
    first, cnt = osc.trace_all(rays, 1)
    ok = cnt > 0
    rng = np.random.default_rng(13)
    lights = scene.desc.contents.lights
    n_checked = 0
    for li in range(scene.n_lights):
        if lights[li].kind != pta.PT_LIGHT_POINT:
            continue
        L = np.array(list(lights[li].vec), np.float32)
        grid = pta.OriginGrid(scene, L, 0, 2e-5 * 1.5)
        assert grid.enabled
        sphere = (first["flags"][ok, 0] & 2) != 0
        dist = first["dist"][ok, 0]
        dn = np.linalg.norm(rays[ok, 3:], axis=1).astype(np.float32)
        tpar = np.where(sphere, dist / dn, dist).astype(np.float32)
        pos = (rays[ok, :3] + rays[ok, 3:] * tpar[:, None]).astype(np.float32)
        gn = rng.normal(size=pos.shape).astype(np.float32)
        gn *= (rng.uniform(0.2, 1.5, len(pos)).astype(np.float32) / np.linalg.norm(gn, axis=1).astype(np.float32))[:, None]
        direction = (pos - L).astype(np.float32)
        ldist = np.sqrt((direction * direction).sum(axis=1, dtype=np.float32)).astype(np.float32)
        sd = (-(direction * (np.float32(1.0) / ldist)[:, None])).astype(np.float32)
        so = (pos + gn * np.float32(0.00001)).astype(np.float32)
        srays = np.concatenate([so, sd], axis=1).astype(np.float32)
        hits, counts = osc.trace_all(srays, 16)
        # only hits that pass the range test |shadow_pos - hit_pos| <= dist matter; they lie within ldist (+ offset) of L
        n_checked += check_rays(grid, srays, direction, hits, counts, max_dist=ldist.astype(np.float64) + 4e-5, origin=L)
    assert n_checked > 100


@pytest.mark.parametrize("name", ["head", "alpha_transparency", "white_furnace_direct", "spheres"])
def test_directional_grid_is_conservative(pta, oracle, scene_cache, name):
    """Shadow rays of a directional light (mod.rs:283-299): origin = hit + n * 1e-5, direction = -light.direction as it
    is (not normalised), no distance limit.  The orthographic grid is looked up with the ray's ORIGIN; every primitive
    of the brute-force hit list must be a candidate whose key (minus its depth bound) does not exceed minus the
    depth of the origin."""
    scene = scene_cache(name)
    osc = oracle.OracleScene(scene.desc, oracle.PTO_BRUTE_FORCE)
    prof = pta.Profile.make(256, 256, 4, 2)
    rays = primary_rays(osc, prof, 1500, seed=21)
    first, cnt = osc.trace_all(rays, 1)
    ok = cnt > 0
    lights = scene.desc.contents.lights
    dirs = [np.array(list(lights[i].vec), np.float32) for i in range(scene.n_lights) if lights[i].kind == pta.PT_LIGHT_DIRECTIONAL]
    dirs += [np.array(v, np.float32) for v in ([0.3, -2.0, 0.4], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0])]   # (+ arbitrary ones)
    rng = np.random.default_rng(22)
    n_checked = 0
    for ldir in dirs:
        sd = (-ldir).astype(np.float32)
        grid = pta.OriginGrid(scene, direction=sd)
        assert grid.enabled and grid.c.kind == 1
        sphere = (first["flags"][ok, 0] & 2) != 0
        dn = np.linalg.norm(rays[ok, 3:], axis=1).astype(np.float32)
        tpar = np.where(sphere, first["dist"][ok, 0] / dn, first["dist"][ok, 0]).astype(np.float32)
        pos = (rays[ok, :3] + rays[ok, 3:] * tpar[:, None]).astype(np.float32)
        gn = rng.normal(size=pos.shape).astype(np.float32)
        so = (pos + gn * np.float32(0.00001)).astype(np.float32)
        srays = np.concatenate([so, np.broadcast_to(sd, so.shape)], axis=1).astype(np.float32)
        hits, counts = osc.trace_all(srays, 16)
        cells = grid.cells(so)
        cut = -grid.depth(so)
        for i in range(len(srays)):
            if counts[i] == 0:
                continue
            prims, key = grid.candidates(int(cells[i]))
            ids = prims & 0x7fffffff
            for k in range(min(int(counts[i]), hits.shape[1])):
                sel = np.nonzero(ids == int(hits[i, k]["prim"]))[0]
                assert len(sel) > 0, (name, ldir, i, int(hits[i, k]["prim"]))
                assert sel[0] < grid.n_global or key[sel[0]] <= cut[i], (name, i, float(key[sel[0]]), float(cut[i]))
                n_checked += 1
    assert n_checked > 50


def test_generated_scene_grids(pta, oracle):
    scene = pta.HostScene.generate_ps5(20000, seed=0)
    osc = oracle.OracleScene(scene.desc, oracle.PTO_BVH)
    prof = pta.Profile.make(320, 180, 2, 2)
    rays = primary_rays(osc, prof, 3000, seed=5)
    grid = pta.OriginGrid(scene, camera_origin(scene))
    assert grid.enabled and grid.n_global <= 2
    hits, counts = osc.trace_all(rays, 16)
    assert check_rays(grid, rays, rays[:, 3:], hits, counts, origin=camera_origin(scene)) > 1500
    # lists stay short: that is the point of the structure
    lens = np.diff(grid.cell_off.astype(np.int64))
    assert lens.max() == grid.c.max_cell_refs
    assert lens[lens > 0].mean() < 40


def test_grid_rejects_bad_input(pta, scene_cache):
    scene = scene_cache("cube")
    g = pta.OriginGrid(scene, [np.nan, 0, 0])
    assert not g.enabled
    with pytest.raises(pta.PtError):
        pta.OriginGrid(scene, [0, 0, 0], 0, -1.0)
    # an origin ON a triangle: that primitive must be global, the grid still works
    v = scene.desc.contents.triangles
    g2 = pta.OriginGrid(scene, [v[0], v[1], v[2]])
    assert (not g2.enabled) or g2.n_global >= 1

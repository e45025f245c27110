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


# ---------------------------------------------------------------------------------------------
# The grids the DEVICE builds (csrc/pt_grid_build.h): the same lists as the host builder's, byte for byte (both compile
# host/og_raster.h), and conservative in their own right.
# ---------------------------------------------------------------------------------------------
def _host_twin(pta, scene, which, res):
    """The host-built grid with the parameters pt_scene_create derives for the device-built one."""
    d = scene.desc.contents
    if which == 0:
        m = np.array(list(d.camera.transform), np.float64).reshape(4, 4)[:3, :3]
        fro = float(np.sqrt((m * m).sum()))
        return pta.OriginGrid(scene, camera_origin(scene), res, 0.0, np.float32(fro * 1.001))
    light = d.lights[which - 1]
    vec = np.array(list(light.vec), np.float32)
    if light.kind == pta.PT_LIGHT_POINT:
        return pta.OriginGrid(scene, vec, res, np.float32(1.05e-5) * np.float32(1.5), np.float32(1.001))
    return pta.OriginGrid(scene, None, res, direction=(np.float32(-1.0) * vec))


@pytest.mark.gpu
@pytest.mark.parametrize("name", MESH_SCENES + SPHERE_SCENES)
def test_device_built_grids_equal_the_host_builders(pta, oracle, scene_cache, gpu_scene_cache, name):
    scene, g = scene_cache(name), gpu_scene_cache(name)
    info = g.info().as_dict()
    assert info["cam_grid_res"] > 0 and info["light_grids"] == scene.n_lights
    n_refs = 0
    for which in range(1 + scene.n_lights):
        dev = pta.OriginGrid.from_device(g, which)
        assert dev.enabled
        host = _host_twin(pta, scene, which, dev.res)
        assert host.enabled and host.res == dev.res and host.n_global == dev.n_global and host.n_refs == dev.n_refs
        assert np.array_equal(host.cell_off, dev.cell_off)
        assert np.array_equal(host.ref_prim[:host.n_refs], dev.ref_prim[:dev.n_refs])
        assert np.array_equal(host.ref_mindist[:host.n_refs].view(np.uint32), dev.ref_mindist[:dev.n_refs].view(np.uint32))
        assert int(host.c.max_cell_refs) == int(dev.c.max_cell_refs)
        n_refs += dev.n_refs
        host.close()
    assert n_refs == info["grid_refs"]
    # ... and the conservativeness check of the camera grid on the device-built lists themselves
    osc = oracle.OracleScene(scene.desc, oracle.PTO_BRUTE_FORCE)
    prof = pta.Profile.make(256, 256, 4, 2)
    rays = primary_rays(osc, prof, 800, seed=21)
    hits, counts = osc.trace_all(rays, 16)
    cam = pta.OriginGrid.from_device(g, 0)
    assert check_rays(cam, rays, rays[:, 3:], hits, counts, origin=camera_origin(scene)) > 150


@pytest.mark.gpu
def test_device_built_grids_of_the_generated_scene(pta):
    """The 30 k-triangle stand-in (opaque and translucent): device == host, camera and light grid."""
    for flags in (0, 5):
        scene = pta.HostScene.generate_ps5(30000, seed=0, flags=flags)
        g = pta.GpuScene(scene)
        for which in (0, 1):
            dev = pta.OriginGrid.from_device(g, which)
            host = _host_twin(pta, scene, which, dev.res)
            assert dev.enabled and host.enabled and host.n_refs == dev.n_refs and host.n_global == dev.n_global
            assert np.array_equal(host.cell_off, dev.cell_off)
            assert np.array_equal(host.ref_prim[:host.n_refs], dev.ref_prim[:dev.n_refs])
            assert np.array_equal(host.ref_mindist[:host.n_refs].view(np.uint32), dev.ref_mindist[:dev.n_refs].view(np.uint32))
            host.close()

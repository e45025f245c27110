"""C-ABI surface (no compute without a GPU) and KD-tree builder invariants."""
import ctypes as C
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def declared_functions(header):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pth?_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pta):
    host, gpu = pta.host_lib(), pta.gpu_lib()
    host_decl = declared_functions("pthost.h")
    gpu_decl = declared_functions("ptgpu.h")
    assert sorted(pta.HOST_SYMBOLS) == host_decl
    assert sorted(pta.GPU_SYMBOLS) == gpu_decl
    for n in host_decl:
        getattr(host, n)
    for n in gpu_decl:
        getattr(gpu, n)
    # the GPU library also carries the host entry points (the CLI links only libptgpu.so)
    for n in host_decl:
        getattr(gpu, n)


def test_struct_sizes_match_the_header(pta):
    src = r'''
#include <stdio.h>
#include "ptgpu.h"
#include "pthost.h"
int main(){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(pt_texture), sizeof(pt_material),
  sizeof(pt_model), sizeof(pt_light), sizeof(pt_camera), sizeof(pt_scene_desc), sizeof(pt_profile), sizeof(pt_opts),
  sizeof(pt_hit), sizeof(pt_timing), sizeof(pt_counters), sizeof(pt_scene_info), sizeof(pth_kdtree)); return 0; }'''
    exe = ROOT / "build" / "abi_sizes"
    exe.parent.mkdir(exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I", str(ROOT / "include"), "-o", str(exe)], input=src.encode(), check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, check=True).stdout.split()]
    py = [C.sizeof(t) for t in (pta.Texture, pta.Material, pta.Model, pta.Light, pta.Camera, pta.SceneDesc, pta.Profile,
                                pta.Opts, pta.Hit, pta.Timing, pta.Counters, pta.SceneInfo, pta.KdTree)]
    assert sizes == py


def test_gpu_entry_points_fail_loudly_without_a_gpu(pta, scene_cache):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = pta.gpu_lib().pt_scene_create(scene_cache("cube").desc, 0, C.byref(h))
    assert rc == -4  # PT_ERR_DEVICE: no CPU fallback
    assert b"HIP" in pta.gpu_lib().pt_last_error() or b"device" in pta.gpu_lib().pt_last_error()


def test_pixel_maps_partition_the_image(pta):
    for (w, h, count, tile) in ((150, 70, 3, 16), (1920, 1080, 8, 32), (33, 9, 2, 16), (64, 64, 5, 32)):
        prof = pta.Profile.make(w, h, 1, 1)
        seen = np.zeros(w * h, int)
        sizes = []
        for r in range(count):
            opts = pta.Opts.make(shard_rank=r, shard_count=count, tile_w=tile, tile_h=tile)
            idx = pta.local_pixel_map(prof, opts)
            sizes.append(len(idx))
            seen[idx] += 1
            # packed order: tiles ascending, row-major inside a tile; tile (tx, ty) belongs to rank (tx + ty * s) % count
            # (diagonal stripes: s = smallest odd number >= 3 coprime to count, 1 for count <= 2)
            tiles_x = (w + tile - 1) // tile
            tx, ty = (idx % w) // tile, idx // w // tile
            k = ty * tiles_x + tx
            s = 1 if count <= 2 else next(v for v in range(3, 99, 2) if np.gcd(v, count) == 1)
            assert (np.diff(k.astype(np.int64)) >= 0).all() and ((tx + ty * s) % count == r).all()
        assert (seen == 1).all()
        assert max(sizes) - min(sizes) <= 2 * tile * tile
    prof = pta.Profile.make(40, 30, 1, 1)
    assert np.array_equal(pta.local_pixel_map(prof, pta.Opts.make()), np.arange(1200))
    with pytest.raises(pta.PtError):
        pta.local_pixel_map(prof, pta.Opts.make(shard_count=2, tile_w=12, tile_h=12))


# ----------------------------------------------------------------------------- KD-tree
def build_kd(pta, scene):
    kd = pta.KdTree()
    pta.check_host(pta.host_lib().pth_kd_build(scene.desc, C.byref(kd)))
    nodes = np.ctypeslib.as_array(C.cast(kd.nodes, C.POINTER(C.c_uint32)), (kd.n_nodes, 2)).copy()
    refs = np.ctypeslib.as_array(kd.refs, (max(1, kd.n_refs),)).copy()[: kd.n_refs]
    info = dict(n_nodes=kd.n_nodes, n_refs=kd.n_refs, n_leaves=kd.n_leaves, depth=kd.depth,
                bmin=np.array(kd.bounds_min), bmax=np.array(kd.bounds_max))
    pta.host_lib().pth_kd_free(C.byref(kd))
    return nodes, refs, info


def visited_prims(nodes, refs, info, o, d):
    """The device traversal (csrc/pt_integrator.h kd_traverse) without the early exit: every primitive in
    every leaf the ray is allowed to visit."""
    o, d = o.astype(np.float32), d.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.float32(1) / d
        t0, t1 = (info["bmin"].astype(np.float32) - o) * inv, (info["bmax"].astype(np.float32) - o) * inv
    tmin, tmax = np.float32(0), np.float32(np.inf)
    for a in range(3):
        tn, tf = (t0[a], t1[a]) if not t0[a] > t1[a] else (t1[a], t0[a])
        tmin = tn if tn > tmin else tmin
        tmax = tf if tf < tmax else tmax
    out = set()
    if tmin > tmax * np.float32(1.0001) + np.float32(1e-4):
        return out
    stack, node = [], 0
    REL, ABS = np.float32(1.00001), np.float32(1e-6)
    while True:
        w0, w1 = nodes[node]
        ax = int(w1 & 3)
        if ax != 3:
            split = np.array([w0], np.uint32).view(np.float32)[0]
            tp = (split - o[ax]) * inv[ax]
            below_first = o[ax] < split or (o[ax] == split and d[ax] <= 0)
            below, above = node + 1, int(w1 >> 2)
            first, second = (below, above) if below_first else (above, below)
            if tp > tmax * REL + ABS or tp <= 0:
                node = first
            elif tp < tmin * (np.float32(2) - REL) - ABS:
                node = second
            else:
                stack.append((second, tmax))
                node, tmax = first, tp
            continue
        n = int(w1 >> 2)
        out.update(int(p) for p in refs[int(w0): int(w0) + n])
        if not stack:
            return out
        tmin = tmax
        node, tmax = stack.pop()


@pytest.mark.parametrize("name", ["cube", "head", "alpha_transparency", "spheres", "white_furnace_direct"])
def test_kd_tree_structure_and_coverage(pta, oracle, scene_cache, name):
    scene = scene_cache(name)
    nodes, refs, info = build_kd(pta, scene)
    n_prims = scene.n_prims
    # structure: DFS layout, every child index valid, every leaf range valid, every primitive referenced
    assert info["n_nodes"] == len(nodes) and (nodes[:, 1] & 3 != 3).sum() + info["n_leaves"] == len(nodes)
    interior = nodes[:, 1] & 3 != 3
    above = nodes[interior, 1] >> 2
    assert (above > np.nonzero(interior)[0] + 1).all() and (above < len(nodes)).all()
    leaf = ~interior
    assert ((nodes[leaf, 0] + (nodes[leaf, 1] >> 2)) <= info["n_refs"]).all()
    assert set(refs.tolist()) == set(range(n_prims))
    assert info["depth"] < 64
    # coverage: every hit the brute-force oracle finds lies in a leaf the traversal visits
    osc = oracle.OracleScene(scene.desc, oracle.PTO_BRUTE_FORCE)
    prof = pta.Profile.make(64, 48, 2, 1)
    rng = np.random.default_rng(11)
    rays = np.stack([osc.primary_ray(prof, int(p), 1) for p in rng.integers(0, 64 * 48, 120)])
    hits, counts = osc.trace_all(rays, 16)
    for r in range(len(rays)):
        vis = visited_prims(nodes, refs, info, rays[r, :3], rays[r, 3:])
        for j in range(min(int(counts[r]), 16)):
            assert int(hits["prim"][r, j]) in vis


def test_kd_tree_peels_planar_ground(pta):
    """A ground plane on the scene's bounding-box face must not be tested by rays that never reach it."""
    scene = pta.HostScene.generate_ps5(4000, 0)
    nodes, refs, info = build_kd(pta, scene)
    n_ground = scene.desc.contents.models[0].tri_count
    # a ray skimming one unit above the whole ground plane
    vis = visited_prims(nodes, refs, info, np.array([-11.9, 1.0, -11.9]), np.array([0.7, 1e-4, 0.7]))
    assert sum(1 for p in vis if p < n_ground) == 0
    assert info["n_refs"] < 16 * scene.n_prims

"""ctypes binding of the CPU oracle (oracle/libptoracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product never imports this module.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
PTO_BRUTE_FORCE, PTO_BVH, PTO_NO_SCENE_SLAB = 0, 1, 2
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = HERE / "libptoracle.so"
        if not path.exists():
            raise RuntimeError(f"{path} missing: run `make oracle` (or __graft_entry__.build())")
        L = C.CDLL(str(path))
        vp = C.c_void_p
        L.pto_scene_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.pto_scene_destroy.argtypes = [vp]
        L.pto_scene_destroy.restype = None
        L.pto_render.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_int, vp, vp, vp]
        L.pto_render_partial.argtypes = [vp, vp, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, vp, vp, vp]
        L.pto_rng_key.argtypes = [C.c_uint64, vp]
        L.pto_chacha_block.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
        L.pto_post_process.argtypes = [vp, vp, C.c_uint64, vp]
        L.pto_debug_render.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_int)]
        L.pto_trace_rays_all.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp, vp]
        L.pto_intersect_triangles.argtypes = [vp, vp, C.c_uint64, vp]
        L.pto_rng_words.argtypes = [vp, C.c_uint64, C.c_uint32, vp]
        L.pto_eval_math.argtypes = [C.c_int, vp, C.c_uint64, vp]
        L.pto_primary_ray.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp]
        L.pto_path_rays.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_uint32, vp]
        L.pto_scene_slab_study_begin.argtypes = [vp, C.c_int]
        L.pto_scene_slab_study.argtypes = [vp, vp]
        L.pto_max_threads.restype = C.c_int
        L.pto_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def max_threads():
    return int(lib().pto_max_threads())


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"oracle error [{rc}]: {lib().pto_last_error().decode(errors='replace')}")


HIT_DTYPE = np.dtype([("prim", "<i4"), ("flags", "<i4"), ("dist", "<f4"), ("u", "<f4"), ("v", "<f4")])
STAT_NAMES = ("samples", "segments", "shadow_rays", "shaded_hits", "rng_draws", "max_draws_per_sample",
              "numeric_errors")


class OracleScene:
    """pto_scene built from a pt_scene_desc pointer (ctypes POINTER(SceneDesc))."""

    def __init__(self, desc_ptr, mode=PTO_BVH, keepalive=None):
        self.handle = C.c_void_p()
        self._keep = keepalive
        _check(lib().pto_scene_create(C.cast(desc_ptr, C.c_void_p), mode, C.byref(self.handle)))

    def render(self, profile, pixel_begin=0, pixel_end=0, threads=0, sample_count=0):
        """Returns (rgb8 [n,3], accum [n,3] f32 = SUM over samples, stats dict).  sample_count: only the first
        that many sample passes (the viewer feed after sample_count passes)."""
        npix = profile.width * profile.height
        end = pixel_end or npix
        n = end - pixel_begin
        rgb = np.empty((n, 3), np.uint8)
        acc = np.empty((n, 3), np.float32)
        stats = (C.c_uint64 * len(STAT_NAMES))()
        _check(lib().pto_render_partial(self.handle, C.byref(profile), sample_count or profile.samples, pixel_begin, end,
                                        threads, rgb.ctypes.data, acc.ctypes.data, C.byref(stats)))
        return rgb, acc, dict(zip(STAT_NAMES, (int(v) for v in stats)))

    def debug_render(self, width, height):
        planes = np.zeros((7, width * height, 3), np.uint8)
        any_hit = C.c_int(0)
        _check(lib().pto_debug_render(self.handle, width, height, planes.ctypes.data, C.byref(any_hit)))
        names = ("normal", "albedo", "opacity", "metalness", "roughness", "emissive", "ior")
        return dict(zip(names, planes)) if any_hit.value else {}

    def trace_all(self, rays, max_hits=8):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), max_hits), dtype=HIT_DTYPE)
        counts = np.zeros(len(rays), np.uint32)
        _check(lib().pto_trace_rays_all(self.handle, rays.ctypes.data, len(rays), max_hits, out.ctypes.data,
                                        counts.ctypes.data))
        return out, counts

    def slab_study_begin(self, on=True):
        """Study hook: casts the slab test rejects are searched for hits anyway and counted (slab_study)."""
        _check(lib().pto_scene_slab_study_begin(self.handle, 1 if on else 0))

    def slab_study(self):
        """(casts the slab test rejected although they had hits, of those: origin strictly inside the scene's box)."""
        out = np.zeros(2, np.uint64)
        _check(lib().pto_scene_slab_study(self.handle, out.ctypes.data))
        return int(out[0]), int(out[1])

    def primary_ray(self, profile, pixel, sample):
        out = np.zeros(6, np.float32)
        _check(lib().pto_primary_ray(self.handle, C.byref(profile), pixel, sample, out.ctypes.data))
        return out

    def path_rays(self, profile, pixel, sample, max_rays=4096):
        """Study hook: every ray ray_cast sees while one sample is rendered (camera ray, shadow rays, bounce rays)."""
        out = np.zeros((max_rays, 6), np.float32)
        n = C.c_uint32(0)
        _check(lib().pto_path_rays(self.handle, C.byref(profile), pixel, sample, out.ctypes.data, max_rays, C.byref(n)))
        return out[: min(n.value, max_rays)].copy()

    def close(self):
        if self.handle:
            lib().pto_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def post_process(profile, accum):
    accum = np.ascontiguousarray(accum, np.float32).reshape(-1, 3)
    out = np.empty((len(accum), 3), np.uint8)
    _check(lib().pto_post_process(C.byref(profile), accum.ctypes.data, len(accum), out.ctypes.data))
    return out


def intersect_triangles(rays, tris):
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    out = np.zeros(len(rays), dtype=HIT_DTYPE)
    _check(lib().pto_intersect_triangles(rays.ctypes.data, tris.ctypes.data, len(rays), out.ctypes.data))
    return out


def rng_words(seeds, n_words):
    seeds = np.ascontiguousarray(seeds, np.uint64)
    out = np.zeros((len(seeds), n_words), np.uint32)
    _check(lib().pto_rng_words(seeds.ctypes.data, len(seeds), n_words, out.ctypes.data))
    return out


def rng_key(seed):
    out = np.zeros(8, np.uint32)
    _check(lib().pto_rng_key(int(seed), out.ctypes.data))
    return out


def chacha_block(key8, counter=0, rounds=12):
    key8 = np.ascontiguousarray(key8, np.uint32)
    out = np.zeros(16, np.uint32)
    _check(lib().pto_chacha_block(key8.ctypes.data, int(counter), rounds, out.ctypes.data))
    return out


MATH_FN = {"pow_inv_gamma": 0, "acos": 1, "sin": 2, "cos": 3, "pow_gamma": 4, "tan": 5}


def eval_math(fn, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    _check(lib().pto_eval_math(MATH_FN[fn] if isinstance(fn, str) else fn, x.ctypes.data, x.size, out.ctypes.data))
    return out

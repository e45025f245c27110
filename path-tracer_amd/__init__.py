"""ctypes view of the C ABI (include/ptgpu.h, include/pthost.h).

The product is the C/HIP code under ``csrc/`` and ``host/``; this module only
mirrors the structs and loads the shared libraries so that tests, bench.py and
__graft_entry__ can call through the same C ABI a Rust host would bind
(INTEGRATION.md).  The directory name contains a hyphen, so it is imported as
``path_tracer_amd`` through ``__graft_entry__.load_package()``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent

PT_OK = 0
PT_MODEL_MESH, PT_MODEL_SPHERE = 0, 1
PT_LIGHT_POINT, PT_LIGHT_DIRECTIONAL = 0, 1
PT_TONEMAP_REINHARD, PT_TONEMAP_FILMIC, PT_TONEMAP_ACES = 0, 1, 2
PT_FLAG_TIMING, PT_FLAG_COUNTERS, PT_FLAG_NO_GRIDS, PT_FLAG_MEGAKERNEL = 1, 2, 4, 8
TONEMAPS = {"REINHARD": 0, "FILMIC": 1, "ACES": 2}
DEBUG_PLANES = ("normal", "albedo", "opacity", "metalness", "roughness", "emissive", "ior")


class Texture(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32),
                ("channels", C.c_uint32), ("_pad", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("emissive", C.c_float * 3), ("opacity", C.c_float),
                ("metalness", C.c_float), ("roughness", C.c_float), ("ior", C.c_float),
                ("tex_albedo", C.c_int32), ("tex_emissive", C.c_int32), ("tex_opacity", C.c_int32),
                ("tex_metalness", C.c_int32), ("tex_roughness", C.c_int32), ("tex_normal", C.c_int32)]


class Model(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("tri_first", C.c_uint32),
                ("tri_count", C.c_uint32), ("center", C.c_float * 3), ("radius", C.c_float)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("vec", C.c_float * 3), ("color", C.c_float * 3), ("size", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("fov", C.c_float), ("zfar", C.c_float), ("znear", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("n_models", C.c_uint32), ("n_materials", C.c_uint32), ("n_textures", C.c_uint32),
                ("n_lights", C.c_uint32), ("n_triangles", C.c_uint64), ("n_texel_bytes", C.c_uint64),
                ("models", C.POINTER(Model)), ("materials", C.POINTER(Material)),
                ("textures", C.POINTER(Texture)), ("lights", C.POINTER(Light)),
                ("triangles", C.POINTER(C.c_float)), ("texels", C.POINTER(C.c_uint8)),
                ("camera", Camera), ("background", C.c_float * 3), ("_pad", C.c_uint32)]


class Profile(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples", C.c_uint32),
                ("bounces", C.c_uint32), ("brdf", C.c_int32), ("tonemap", C.c_int32)]

    @classmethod
    def make(cls, width=1920, height=1080, samples=64, bounces=4, tonemap="FILMIC"):
        return cls(width, height, samples, bounces, 0, TONEMAPS[tonemap] if isinstance(tonemap, str) else tonemap)


PROGRESS_FN = C.CFUNCTYPE(None, C.c_uint32, C.c_uint32, C.c_void_p)
PREVIEW_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p)


class Opts(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("device", C.c_int32), ("shard_rank", C.c_uint32),
                ("shard_count", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                ("sample_batch", C.c_uint32), ("_pad", C.c_uint32), ("progress", PROGRESS_FN),
                ("progress_user", C.c_void_p), ("preview", PREVIEW_FN), ("preview_user", C.c_void_p)]

    @classmethod
    def make(cls, flags=0, device=-1, shard_rank=0, shard_count=1, tile_w=0, tile_h=0, sample_batch=0, preview=None):
        o = cls(flags, device, shard_rank, shard_count, tile_w, tile_h, sample_batch, 0,
                C.cast(None, PROGRESS_FN), None, C.cast(None, PREVIEW_FN), None)
        if preview is not None:
            o._keep = PREVIEW_FN(preview)  # keep the trampoline alive with the struct
            o.preview = o._keep
        return o


class Hit(C.Structure):
    _fields_ = [("prim", C.c_int32), ("flags", C.c_int32), ("dist", C.c_float), ("u", C.c_float),
                ("v", C.c_float)]


class Timing(C.Structure):
    _fields_ = [("launches", C.c_uint32), ("integrate_ms", C.c_float), ("postprocess_ms", C.c_float),
                ("total_ms", C.c_float), ("generate_ms", C.c_float), ("trace_ms", C.c_float),
                ("shade_ms", C.c_float), ("shadow_ms", C.c_float), ("accumulate_ms", C.c_float),
                ("stage_launches", C.c_uint32), ("bounce0_launches", C.c_uint32), ("bounce0_ms", C.c_float)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "segments", "shadow_rays", "nodes_visited",
                                          "tris_tested", "shaded_hits", "rng_draws", "restarts",
                                          "max_nodes_per_cast", "casts_over_1k_nodes", "trace_nodes", "trace_tris",
                                          "shadow_skipped", "bounce0_hits", "bounce0_shadow_rays", "bounce0_tris",
                                          "grid_tris", "bounce0_cam_tris", "deferred_casts", "exact_casts", "masked_casts", "bounce0_masked")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SceneInfo(C.Structure):
    _fields_ = [("n_prims", C.c_uint64), ("n_kd_nodes", C.c_uint64), ("n_kd_leaves", C.c_uint64),
                ("n_leaf_refs", C.c_uint64), ("kd_depth", C.c_uint32), ("has_translucent", C.c_uint32),
                ("kd_build_seconds", C.c_float), ("upload_seconds", C.c_float), ("device_bytes", C.c_uint64),
                ("cam_grid_res", C.c_uint32), ("light_grids", C.c_uint32), ("grid_refs", C.c_uint64),
                ("grid_build_seconds", C.c_float), ("escape_build_seconds", C.c_float), ("escape_prims", C.c_uint32),
                ("escape_clear_fraction", C.c_float), ("queue_bytes", C.c_uint64), ("queue_chunk_items", C.c_uint32),
                ("frame_planned", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class KdNode(C.Structure):
    _fields_ = [("w0", C.c_uint32), ("w1", C.c_uint32)]


class KdTree(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_refs", C.c_uint64), ("n_leaves", C.c_uint64),
                ("depth", C.c_uint32), ("_pad", C.c_uint32), ("bounds_min", C.c_float * 3),
                ("bounds_max", C.c_float * 3), ("nodes", C.POINTER(KdNode)), ("refs", C.POINTER(C.c_uint32)),
                ("build_seconds", C.c_double), ("expected_nodes", C.c_double), ("expected_tests", C.c_double)]


class GridRef(C.Structure):
    _fields_ = [("prim", C.c_uint32), ("mindist", C.c_float)]


class OriginGridC(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("res", C.c_uint32), ("n_cells", C.c_uint64), ("n_refs", C.c_uint64),
                ("n_global", C.c_uint32), ("enabled", C.c_uint32), ("max_cell_refs", C.c_uint32),
                ("ray_offset", C.c_float), ("cell_off", C.POINTER(C.c_uint32)), ("refs", C.POINTER(GridRef)),
                ("build_seconds", C.c_double), ("kind", C.c_uint32), ("axis_u", C.c_float * 3),
                ("axis_v", C.c_float * 3), ("axis_w", C.c_float * 3), ("u0", C.c_float), ("v0", C.c_float),
                ("cells_per_unit", C.c_float)]


class OracleStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "segments", "shadow_rays", "shaded_hits", "rng_draws",
                                          "max_draws_per_sample", "numeric_errors")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


_host = None
_gpu = None

# Every symbol include/pthost.h declares (tests check they are all exported).
HOST_SYMBOLS = ["pth_scene_load_isf", "pth_scene_free", "pth_scene_desc", "pth_scene_generate_ps5",
                "pth_scene_save_isf", "pth_convert_gltf", "pth_profile_load", "pth_profile_parse", "pth_png_read",
                "pth_png_decode", "pth_png_write_rgb8", "pth_free", "pth_prim_count", "pth_kd_build",
                "pth_kd_free", "pth_origin_grid_build", "pth_ortho_grid_build", "pth_origin_grid_auto_resolution", "pth_origin_grid_free",
                "pth_last_error"]
# Every symbol include/ptgpu.h declares.
GPU_SYMBOLS = ["pt_scene_create", "pt_scene_destroy", "pt_prep_create", "pt_prep_destroy", "pt_scene_create_from_prep",
               "pt_comm_unique_id", "pt_comm_create", "pt_comm_create_all", "pt_comm_destroy", "pt_gather_tiles", "pt_render_gathered", "pt_local_pixel_count", "pt_local_pixel_map",
               "pt_render", "pt_render_device", "pt_debug_render", "pt_assemble_tiles", "pt_get_timing", "pt_get_counters",
               "pt_scene_get_info", "pt_scene_set_cu_mask", "pt_stream_create_cu_mask", "pt_stream_destroy", "pt_get_cull_stats", "pt_scene_escape_copy", "pt_scene_grid_header", "pt_scene_grid_copy", "pt_trace_rays", "pt_trace_rays_wavefront",
               "pt_trace_rays_all", "pt_intersect_triangles",
               "pt_rng_words", "pt_eval_math", "pt_measure_copy_bandwidth", "pt_measure_gather_rate", "pt_last_error",
               "pt_version"]


def host_lib():
    """libpthost.so: loader, profile, PNG, generator, KD builder (no HIP)."""
    global _host
    if _host is None:
        path = PKG_DIR / "libpthost.so"
        if not path.exists():
            raise PtError(-2, f"{path} is missing: run `make host` (or __graft_entry__.build())")
        L = C.CDLL(str(path))
        L.pth_scene_load_isf.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.pth_scene_free.argtypes = [C.c_void_p]
        L.pth_scene_free.restype = None
        L.pth_scene_desc.argtypes = [C.c_void_p]
        L.pth_scene_desc.restype = C.POINTER(SceneDesc)
        L.pth_scene_generate_ps5.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]
        L.pth_scene_save_isf.argtypes = [C.c_void_p, C.c_char_p]
        L.pth_convert_gltf.argtypes = [C.c_char_p, C.c_char_p]
        L.pth_profile_load.argtypes = [C.c_char_p, C.POINTER(Profile)]
        L.pth_profile_parse.argtypes = [C.c_char_p, C.POINTER(Profile)]
        L.pth_png_read.argtypes = [C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.POINTER(C.c_uint8))]
        L.pth_png_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8))]
        L.pth_png_write_rgb8.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.pth_free.argtypes = [C.c_void_p]
        L.pth_free.restype = None
        L.pth_prim_count.argtypes = [C.POINTER(SceneDesc)]
        L.pth_prim_count.restype = C.c_uint64
        L.pth_kd_build.argtypes = [C.POINTER(SceneDesc), C.POINTER(KdTree)]
        L.pth_kd_free.argtypes = [C.POINTER(KdTree)]
        L.pth_kd_free.restype = None
        L.pth_origin_grid_build.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_float), C.c_uint32, C.c_float,
                                            C.c_float, C.POINTER(OriginGridC)]
        L.pth_ortho_grid_build.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_float), C.c_uint32, C.POINTER(OriginGridC)]
        L.pth_origin_grid_free.argtypes = [C.POINTER(OriginGridC)]
        L.pth_origin_grid_free.restype = None
        L.pth_last_error.restype = C.c_char_p
        _host = L
    return _host


def gpu_lib():
    """libptgpu.so: the HIP integrator behind the C ABI.  Fails loudly when missing."""
    global _gpu
    if _gpu is None:
        path = Path(os.environ.get("PT_GPU_LIB", PKG_DIR / "libptgpu.so"))  # override: A/B builds only
        if not path.exists():
            raise PtError(-4, f"{path} is missing: the HIP extension was not built "
                              "(run `make gpu` or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(str(path))
        vp = C.c_void_p
        L.pt_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(vp)]
        L.pt_scene_destroy.argtypes = [vp]
        L.pt_scene_destroy.restype = None
        L.pt_prep_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(vp)]
        L.pt_prep_destroy.argtypes = [vp]
        L.pt_prep_destroy.restype = None
        L.pt_scene_create_from_prep.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.pt_comm_unique_id.argtypes = [vp]
        L.pt_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.pt_comm_create_all.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
        L.pt_comm_destroy.argtypes = [vp]
        L.pt_comm_destroy.restype = None
        L.pt_gather_tiles.argtypes = [vp, C.POINTER(Profile), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, vp, vp, vp, vp]
        L.pt_render_gathered.argtypes = [vp, vp, C.POINTER(Profile), C.POINTER(Opts), C.c_uint64, vp]
        L.pt_local_pixel_count.argtypes = [C.POINTER(Profile), C.POINTER(Opts)]
        L.pt_local_pixel_count.restype = C.c_uint64
        L.pt_local_pixel_map.argtypes = [C.POINTER(Profile), C.POINTER(Opts), vp]
        L.pt_render.argtypes = [vp, C.POINTER(Profile), C.POINTER(Opts), vp, vp]
        L.pt_render_device.argtypes = [vp, C.POINTER(Profile), C.POINTER(Opts), vp, vp, vp]
        L.pt_assemble_tiles.argtypes = [C.POINTER(Profile), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                        C.c_uint32, vp, vp, vp]
        L.pt_debug_render.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_int)]
        L.pt_get_timing.argtypes = [vp, C.POINTER(Timing)]
        L.pt_get_counters.argtypes = [vp, C.POINTER(Counters)]
        L.pt_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
        L.pt_scene_set_cu_mask.argtypes = [vp, vp, C.c_uint32]
        L.pt_stream_create_cu_mask.argtypes = [C.c_int, vp, C.c_uint32, C.POINTER(C.c_void_p)]
        L.pt_stream_destroy.argtypes = [vp]
        L.pt_trace_rays.argtypes = [vp, vp, C.c_uint64, vp]
        L.pt_trace_rays_wavefront.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, vp]
        L.pt_scene_grid_header.argtypes = [vp, C.c_uint32, vp]
        L.pt_scene_escape_copy.argtypes = [vp, vp, C.c_uint64]
        L.pt_get_cull_stats.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.pt_scene_grid_copy.argtypes = [vp, C.c_uint32, vp, vp]
        L.pt_trace_rays_all.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp, vp]
        L.pt_intersect_triangles.argtypes = [C.c_int, vp, vp, C.c_uint64, vp]
        L.pt_rng_words.argtypes = [C.c_int, vp, C.c_uint64, C.c_uint32, vp]
        L.pt_eval_math.argtypes = [C.c_int, C.c_int, vp, C.c_uint64, vp]
        L.pt_measure_copy_bandwidth.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
        L.pt_measure_gather_rate.argtypes = [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.pt_last_error.restype = C.c_char_p
        L.pt_version.restype = C.c_char_p
        _gpu = L
    return _gpu


def check_host(rc):
    if rc != PT_OK:
        raise PtError(rc, host_lib().pth_last_error().decode(errors="replace"))


def check_gpu(rc):
    if rc != PT_OK:
        raise PtError(rc, gpu_lib().pt_last_error().decode(errors="replace"))


class HostScene:
    """Owned pth_scene handle (ISF file or generated)."""

    def __init__(self, handle):
        self.handle = handle
        self.desc = host_lib().pth_scene_desc(handle)

    @classmethod
    def load_isf(cls, path):
        h = C.c_void_p()
        check_host(host_lib().pth_scene_load_isf(os.fsencode(str(path)), C.byref(h)))
        return cls(h)

    @classmethod
    def generate_ps5(cls, target_tris, seed=0, flags=0):
        h = C.c_void_p()
        check_host(host_lib().pth_scene_generate_ps5(target_tris, seed, flags, C.byref(h)))
        return cls(h)

    def save_isf(self, directory):
        check_host(host_lib().pth_scene_save_isf(self.handle, os.fsencode(str(directory))))

    @property
    def n_triangles(self):
        return int(self.desc.contents.n_triangles)

    @property
    def n_prims(self):
        return int(host_lib().pth_prim_count(self.desc))

    @property
    def n_lights(self):
        return int(self.desc.contents.n_lights)

    def close(self):
        if self.handle:
            host_lib().pth_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OriginGrid:
    """pth_origin_grid (host/origin_grid.cpp): cube map of primitive lists around one point, with the cell lookup
    restated in numpy f32 exactly as the device computes it (csrc/pt_grid.h og_cell)."""

    def __init__(self, host_scene, origin=None, res=0, ray_offset=0.0, max_dir_len=1.001, direction=None):
        """origin: cube-map grid around a point; direction: orthographic grid for rays of that direction."""
        import numpy as np
        self.c = OriginGridC()
        if direction is not None:
            dvec = (C.c_float * 3)(*[float(v) for v in direction])
            check_host(host_lib().pth_ortho_grid_build(host_scene.desc, dvec, res, C.byref(self.c)))
        else:
            o = (C.c_float * 3)(*[float(v) for v in origin])
            check_host(host_lib().pth_origin_grid_build(host_scene.desc, o, res, ray_offset, max_dir_len, C.byref(self.c)))
        self.enabled = bool(self.c.enabled)
        self.res, self.n_global, self.n_refs = int(self.c.res), int(self.c.n_global), int(self.c.n_refs)
        if self.enabled:
            self.cell_off = np.ctypeslib.as_array(self.c.cell_off, (int(self.c.n_cells) + 1,))
            raw = np.ctypeslib.as_array(C.cast(self.c.refs, C.POINTER(C.c_uint32)), (max(1, self.n_refs) * 2,))
            self.ref_prim = raw[0::2]
            self.ref_mindist = raw[1::2].view(np.float32)

    @classmethod
    def from_device(cls, gpu_scene, which):
        """The grid the DEVICE built for a scene (csrc/pt_grid_build.h): which = 0 camera, 1 + i light i."""
        import numpy as np
        self = cls.__new__(cls)
        self.c = OriginGridC()
        self._owned = False
        check_gpu(gpu_scene.lib.pt_scene_grid_header(gpu_scene.handle, which, C.byref(self.c)))
        self.enabled = bool(self.c.enabled)
        self.res, self.n_global, self.n_refs = int(self.c.res), int(self.c.n_global), int(self.c.n_refs)
        if self.enabled:
            self.cell_off = np.zeros(int(self.c.n_cells) + 1, np.uint32)
            raw = np.zeros(max(1, self.n_refs) * 2, np.uint32)
            check_gpu(gpu_scene.lib.pt_scene_grid_copy(gpu_scene.handle, which, self.cell_off.ctypes.data, raw.ctypes.data))
            self.ref_prim = raw[0::2]
            self.ref_mindist = raw[1::2].view(np.float32)
        return self

    def cells(self, w):
        """Cell index of every direction w [n, 3] - orthographic grids: of every ray ORIGIN w - in the f32 arithmetic
        of the device (csrc/pt_grid.h og_cell / og_cell_ortho)."""
        import numpy as np
        w = np.ascontiguousarray(w, np.float32).reshape(-1, 3)
        if self.c.kind == 1:
            au, av = np.array(list(self.c.axis_u), np.float32), np.array(list(self.c.axis_v), np.float32)
            dot = lambda a: ((w[:, 0] * a[0] + w[:, 1] * a[1]).astype(np.float32) + w[:, 2] * a[2]).astype(np.float32)
            with np.errstate(all="ignore"):
                fu = ((dot(au) - np.float32(self.c.u0)) * np.float32(self.c.cells_per_unit)).astype(np.float32)
                fv = ((dot(av) - np.float32(self.c.v0)) * np.float32(self.c.cells_per_unit)).astype(np.float32)
            top = np.float32(self.res - 1)
            iu = np.nan_to_num(np.where(fu >= 0, np.minimum(np.floor(fu), top), 0)).astype(np.int64)
            iv = np.nan_to_num(np.where(fv >= 0, np.minimum(np.floor(fv), top), 0)).astype(np.int64)
            return iv * self.res + iu
        a = np.abs(w)
        axis = np.where((a[:, 0] >= a[:, 1]) & (a[:, 0] >= a[:, 2]), 0, np.where(a[:, 1] >= a[:, 2], 1, 2))
        idx = np.arange(len(w))
        wa, wb, wc = w[idx, axis], w[idx, (axis + 1) % 3], w[idx, (axis + 2) % 3]
        with np.errstate(all="ignore"):
            inv = np.float32(1.0) / np.abs(wa)
            half = np.float32(0.5 * self.res)
            fu = (wb * inv + np.float32(1.0)) * half
            fv = (wc * inv + np.float32(1.0)) * half
        top = np.float32(self.res - 1)
        iu = np.where(fu >= 0, np.minimum(np.floor(fu), top), 0)   # NaN -> 0, like the device's clamp
        iv = np.where(fv >= 0, np.minimum(np.floor(fv), top), 0)
        iu = np.nan_to_num(iu).astype(np.int64)
        iv = np.nan_to_num(iv).astype(np.int64)
        face = 2 * axis + (wa < 0)
        return (face * self.res + iv) * self.res + iu

    def depth(self, p):
        """Orthographic grids: depth of the points p along the rays' direction, as the device computes it."""
        import numpy as np
        p = np.ascontiguousarray(p, np.float32).reshape(-1, 3)
        a = np.array(list(self.c.axis_w), np.float32)
        return ((p[:, 0] * a[0] + p[:, 1] * a[1]).astype(np.float32) + p[:, 2] * a[2]).astype(np.float32)

    def candidates(self, cell):
        """(primitive words, mindist) of one cell, the global block in front."""
        import numpy as np
        b, e = int(self.cell_off[cell]), int(self.cell_off[cell + 1])
        sel = np.r_[0:self.n_global, b:e]
        return self.ref_prim[sel], self.ref_mindist[sel]

    def close(self):
        if getattr(self, "_owned", True):
            host_lib().pth_origin_grid_free(C.byref(self.c))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def convert_gltf(input_path, output_dir):
    """`path-tracer convert`: glTF 2.0 -> output_dir/scene.isf + textures (src/scene/gltf.rs:146-265)."""
    check_host(host_lib().pth_convert_gltf(os.fsencode(str(input_path)), os.fsencode(str(output_dir))))


def load_profile(path=None, text=None):
    p = Profile()
    if text is not None:
        check_host(host_lib().pth_profile_parse(text.encode(), C.byref(p)))
    else:
        check_host(host_lib().pth_profile_load(os.fsencode(str(path)) if path else None, C.byref(p)))
    return p


class Prep:
    """pt_prep: the host half of pt_scene_create (KD-tree, origin grids), built once for several devices."""

    def __init__(self, host_scene: HostScene):
        self.handle = C.c_void_p()
        self._host = host_scene
        check_gpu(gpu_lib().pt_prep_create(host_scene.desc, C.byref(self.handle)))

    def close(self):
        if self.handle:
            gpu_lib().pt_prep_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """pt_comm: RCCL communicator of the tile gather (one process per GPU: unique id from rank 0)."""

    def __init__(self, unique_id: bytes, rank, size, device):
        self.handle = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check_gpu(gpu_lib().pt_comm_create(buf, rank, size, device, C.byref(self.handle)))
        self.rank, self.size = rank, size

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        check_gpu(gpu_lib().pt_comm_unique_id(buf))
        return bytes(buf)

    def gather_tiles(self, profile, tile_w, tile_h, slice_pixels, elem_bytes, d_local, d_gathered, d_image, stream=0):
        check_gpu(gpu_lib().pt_gather_tiles(self.handle, C.byref(profile), tile_w, tile_h, slice_pixels, elem_bytes,
                                            d_local, d_gathered, d_image, stream))

    def close(self):
        if self.handle:
            gpu_lib().pt_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuScene:
    """Device-resident scene (pt_scene_create / pt_scene_destroy)."""

    def __init__(self, host_scene: HostScene, device=0, prep=None):
        self.lib = gpu_lib()
        self.handle = C.c_void_p()
        self._host = host_scene
        if prep is not None:   # host work (KD-tree, origin grids) done once by Prep, uploaded here
            check_gpu(self.lib.pt_scene_create_from_prep(prep.handle, device, C.byref(self.handle)))
        else:
            check_gpu(self.lib.pt_scene_create(host_scene.desc, device, C.byref(self.handle)))

    def info(self):
        i = SceneInfo()
        check_gpu(self.lib.pt_scene_get_info(self.handle, C.byref(i)))
        return i

    def render(self, profile, opts=None):
        """Host-buffer render: returns (rgb8 [n,3] uint8, accum [n,3] float32)."""
        import numpy as np
        opts = opts or Opts.make()
        n = int(self.lib.pt_local_pixel_count(C.byref(profile), C.byref(opts)))
        rgb = np.empty((n, 3), np.uint8)
        acc = np.empty((n, 3), np.float32)
        check_gpu(self.lib.pt_render(self.handle, C.byref(profile), C.byref(opts), rgb.ctypes.data, acc.ctypes.data))
        return rgb, acc

    def render_device(self, profile, opts, d_rgb8, d_accum, stream=0):
        check_gpu(self.lib.pt_render_device(self.handle, C.byref(profile), C.byref(opts), d_rgb8, d_accum, stream))

    def debug_render(self, width, height):
        """--debug-textures planes: dict name -> [H*W, 3] uint8, or {} when nothing was hit."""
        import numpy as np
        planes = np.zeros((7, width * height, 3), np.uint8)
        any_hit = C.c_int(0)
        check_gpu(self.lib.pt_debug_render(self.handle, width, height, planes.ctypes.data, C.byref(any_hit)))
        if not any_hit.value:
            return {}
        return dict(zip(DEBUG_PLANES, planes))

    def timing(self):
        t = Timing()
        check_gpu(self.lib.pt_get_timing(self.handle, C.byref(t)))
        return t

    def cull_stats(self):
        """(8x8 pixel blocks, blocks the camera-grid cull found empty) of the last frame; (0, 0): no cull ran."""
        n, e = C.c_uint32(), C.c_uint32()
        check_gpu(self.lib.pt_get_cull_stats(self.handle, C.byref(n), C.byref(e)))
        return n.value, e.value

    def counters(self):
        c = Counters()
        check_gpu(self.lib.pt_get_counters(self.handle, C.byref(c)))
        return c

    def trace(self, rays):
        import numpy as np
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        check_gpu(self.lib.pt_trace_rays(self.handle, rays.ctypes.data, len(rays), out.ctypes.data))
        return out

    def escape_masks(self):
        """(normals [n, 3] (0: no mask), v0 [n, 3], bits [n, 6, 64] bool: True = 'may hit') as the device built them."""
        import numpy as np
        n = int(self.info().n_prims)
        raw = np.zeros((n, 20), np.uint32)
        check_gpu(self.lib.pt_scene_escape_copy(self.handle, raw.ctypes.data, raw.nbytes))
        f = raw.view(np.float32)
        normals, v0 = f[:, 0:3].copy(), np.stack([f[:, 3], f[:, 4], f[:, 5]], axis=1)
        words = raw[:, 8:20].reshape(n, 6, 2)
        bits = ((words[:, :, :, None] >> np.arange(32, dtype=np.uint32)) & 1).astype(bool).reshape(n, 6, 64)
        return normals, v0, bits

    def trace_wavefront(self, rays, start_prims=None, mode=0):
        """Closest hits through k_wf_trace itself (mode bit 0: entry lists from start_prims, bit 1: k_wf_trace_wide)."""
        import numpy as np
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        sp = None if start_prims is None else np.ascontiguousarray(start_prims, np.uint32)
        check_gpu(self.lib.pt_trace_rays_wavefront(self.handle, rays.ctypes.data, None if sp is None else sp.ctypes.data,
                                                   len(rays), mode, out.ctypes.data))
        return out

    def trace_all(self, rays, max_hits=8):
        import numpy as np
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), max_hits), dtype=HIT_DTYPE)
        counts = np.zeros(len(rays), np.uint32)
        check_gpu(self.lib.pt_trace_rays_all(self.handle, rays.ctypes.data, len(rays), max_hits,
                                             out.ctypes.data, counts.ctypes.data))
        return out, counts

    def close(self):
        if self.handle:
            self.lib.pt_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


try:
    import numpy as _np
    HIT_DTYPE = _np.dtype([("prim", "<i4"), ("flags", "<i4"), ("dist", "<f4"), ("u", "<f4"), ("v", "<f4")])
except Exception:  # pragma: no cover
    HIT_DTYPE = None


def local_pixel_map(profile, opts):
    import numpy as np
    lib = gpu_lib()
    n = int(lib.pt_local_pixel_count(C.byref(profile), C.byref(opts)))
    out = np.empty(n, np.uint32)
    check_gpu(lib.pt_local_pixel_map(C.byref(profile), C.byref(opts), out.ctypes.data))
    return out


def measure_copy_bandwidth(device=0, nbytes=1 << 30, reps=5):
    """GB/s (read + write) of a plain device copy kernel: the achievable-HBM yardstick."""
    out = C.c_double(0.0)
    check_gpu(gpu_lib().pt_measure_copy_bandwidth(device, nbytes, reps, C.byref(out)))
    return out.value


def measure_gather_rate(device=0, table_bytes=32 << 20, bytes_per_load=8, loads_per_lane=512):
    """1e9 scattered lane-loads per second (the KD walk's access pattern): the ceiling the traversal kernels see."""
    out = C.c_double(0.0)
    check_gpu(gpu_lib().pt_measure_gather_rate(device, table_bytes, bytes_per_load, loads_per_lane, C.byref(out)))
    return out.value

/*
 * ptgpu.h — C ABI of the MI355X path-tracing integrator (libptgpu.so).
 *
 * This is the drop-in boundary for the per-pixel sampling hot path of
 * flomonster/path-tracer.  The reference has no FFI; the narrowest seam is
 *
 *     Renderer::new(&RenderConfig, Profile)            src/renderer/mod.rs:61-73
 *     Renderer::render(&self, &Scene) -> RgbImage      src/renderer/mod.rs:76-169
 *
 * called from run_render (src/main.rs:46-47) and the golden-image test helper
 * (src/main.rs:79).  A Rust host would bind exactly the functions declared
 * here (see INTEGRATION.md for the `extern "C"` block); the C++ host in
 * path-tracer_amd/host/ uses them the same way.
 *
 * Everything is plain C: POD structs, pointers and sizes.  The caller owns all
 * host buffers; the library copies the scene to the device in
 * pt_scene_create() and owns device memory until pt_scene_destroy().
 *
 * All functions return 0 on success and a negative pt_status on failure;
 * pt_last_error() returns a thread-local message (the CLI maps any failure to
 * exit code 2 like src/main.rs:14-22).
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* status codes                                                        */
/* ------------------------------------------------------------------ */
enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = -1,   /* bad argument / malformed scene            */
    PT_ERR_IO = -2,        /* file could not be read / written          */
    PT_ERR_PARSE = -3,     /* JSON / YAML / PNG syntax                  */
    PT_ERR_DEVICE = -4,    /* HIP runtime failure, no GPU               */
    PT_ERR_NUMERIC = -5,   /* NaN hit distance etc. (reference panics:  */
                           /* src/renderer/utils.rs:19)                 */
    PT_ERR_UNSUPPORTED = -6
};

/* ------------------------------------------------------------------ */
/* scene description: flat mirror of src/scene/internal/{mod,model,    */
/* material,light,camera,vertex,triangle}.rs                           */
/* ------------------------------------------------------------------ */

enum { PT_MODEL_MESH = 0, PT_MODEL_SPHERE = 1 };     /* internal/model.rs:10-21 */
enum { PT_LIGHT_POINT = 0, PT_LIGHT_DIRECTIONAL = 1 }; /* internal/light.rs:6-17 */
enum { PT_BRDF_COOK_TORRANCE = 0 };                  /* renderer/brdf/mod.rs:50-55 */
enum { PT_TONEMAP_REINHARD = 0, PT_TONEMAP_FILMIC = 1, PT_TONEMAP_ACES = 2 }; /* tonemap.rs:5-13 */

/* One decoded texture (image crate: into_rgb8() → 3 channels,
 * into_luma8() → 1 channel; internal/texture_bank.rs:21-51).  Row 0 is the
 * top row; texels are tightly packed u8. */
typedef struct pt_texture {
    uint64_t offset;    /* byte offset into pt_scene_desc.texels */
    uint32_t width;
    uint32_t height;
    uint32_t channels;  /* 1 or 3 */
    uint32_t _pad;
} pt_texture;

/* internal/material.rs:11-26.  tex_* = index into textures or -1. */
typedef struct pt_material {
    float albedo[3];
    float emissive[3];
    float opacity;
    float metalness;
    float roughness;
    float ior;
    int32_t tex_albedo;     /* rgb  */
    int32_t tex_emissive;   /* rgb  */
    int32_t tex_opacity;    /* luma */
    int32_t tex_metalness;  /* luma */
    int32_t tex_roughness;  /* luma */
    int32_t tex_normal;     /* rgb  */
} pt_material;

/* internal/model.rs:10-21.  Meshes own triangles [tri_first, tri_first+tri_count)
 * of pt_scene_desc.triangles; spheres use center/radius. */
typedef struct pt_model {
    int32_t kind;       /* PT_MODEL_* */
    int32_t material;   /* index into materials */
    uint32_t tri_first;
    uint32_t tri_count;
    float center[3];
    float radius;
} pt_model;

/* internal/light.rs:6-17.  vec = position (point) or direction (directional). */
typedef struct pt_light {
    int32_t kind;       /* PT_LIGHT_* */
    float vec[3];
    float color[3];
    float size;         /* unused by the integrator (hard shadows) */
} pt_light;

/* internal/camera.rs:7-48.  transform[4*k + r] = column k, row r (ISF
 * "transform"[k] is column k, cgmath Matrix4 layout). */
typedef struct pt_camera {
    float transform[16];
    float fov;          /* radians */
    float zfar;
    float znear;
} pt_camera;

/* internal/mod.rs:26-32.  triangles: 24 f32 per triangle in ISF order
 * (position3, normal3, tex_coords2) x 3 vertices; models reference
 * contiguous ranges in model order, so the global triangle index is the
 * reference's (model index, triangle index) order. */
typedef struct pt_scene_desc {
    uint32_t n_models;
    uint32_t n_materials;
    uint32_t n_textures;
    uint32_t n_lights;
    uint64_t n_triangles;
    uint64_t n_texel_bytes;
    const pt_model* models;
    const pt_material* materials;
    const pt_texture* textures;
    const pt_light* lights;
    const float* triangles;
    const uint8_t* texels;
    pt_camera camera;
    float background[3];
    uint32_t _pad;
} pt_scene_desc;

/* config/profile.rs:10-25 (defaults 1920x1080, bounces 4, samples 64,
 * COOK_TORRANCE, FILMIC). */
typedef struct pt_profile {
    uint32_t width;
    uint32_t height;
    uint32_t samples;
    uint32_t bounces;
    int32_t brdf;
    int32_t tonemap;
} pt_profile;

/* pt_opts.flags */
enum {
    PT_FLAG_TIMING = 1u << 0,   /* record HIP events around every kernel launch */
    PT_FLAG_COUNTERS = 1u << 1, /* run the instrumented kernel variant (ray /
                                   node / triangle counters; not for timing)  */
    PT_FLAG_NO_GRIDS = 1u << 2, /* cast camera and shadow rays through the KD-tree like every other ray
                                   (A/B measurements; the parity tests compare the two paths)          */
    PT_FLAG_MEGAKERNEL = 1u << 3, /* the one-lane-per-pixel integrator (k_render): a second, independent
                                   implementation of the path for cross-checks; ~10x slower            */
};

/* Which pixels this call renders.  The image is cut into tile_w x tile_h
 * tiles numbered row-major (k = ty * tiles_x + tx); this call renders the tiles
 * with (tx + ty * s) % shard_count == shard_rank - diagonal
 * stripes, s = the smallest odd number >= 3 coprime to shard_count (1 for two
 * ranks: a checkerboard), so that every rank takes tiles from every column and
 * row of the image (pt_local_pixel_map gives the exact map).  Every pixel keeps its GLOBAL
 * index i = x + y*W in the seed formula (renderer/mod.rs:110-112), so any
 * sharding is bit-identical to the unsharded render.  Output buffers are
 * PACKED in local order: tiles in ascending k, row-major inside a tile
 * (clipped at the image border).  shard_count <= 1 means the whole image in
 * plain row-major order. */
typedef struct pt_opts {
    uint32_t flags;
    int32_t device;        /* HIP device ordinal; -1 = current */
    uint32_t shard_rank;
    uint32_t shard_count;
    uint32_t tile_w;       /* 0 -> 32 */
    uint32_t tile_h;       /* 0 -> 32 */
    uint32_t sample_batch; /* samples per kernel launch; 0 = library default */
    uint32_t _pad;
    void (*progress)(uint32_t done_samples, uint32_t total_samples, void* user);
    void* progress_user;
    /* Progressive output (the reference's viewer feed, renderer/mod.rs:133-141: every pixel is sent as
     * post_processing(pixel / current_sample)).  When set, pt_render() calls it after every sample batch
     * with the packed RGB8 image of the samples done so far (same layout as the rgb8 result, valid only
     * during the call).  Host-buffer entry point only. */
    void (*preview)(const uint8_t* rgb8, uint64_t n_pixels, uint32_t done_samples, uint32_t total_samples, void* user);
    void* preview_user;
} pt_opts;

typedef struct pt_scene pt_scene;

/* ------------------------------------------------------------------ */
/* the hot path                                                        */
/* ------------------------------------------------------------------ */

/* Upload a scene: builds the KD-tree (replaces kdtree-ray),
 * the sRGB->linear table, and copies everything to the device. */
int pt_scene_create(const pt_scene_desc* desc, int device, pt_scene** out);
void pt_scene_destroy(pt_scene* scene);

/* The host half of pt_scene_create (validation, KD build, origin grids) as an object of its own: build it ONCE
 * and upload it to every device of a multi-GPU render instead of repeating seconds of CPU work per device. */
typedef struct pt_prep pt_prep;
int pt_prep_create(const pt_scene_desc* desc, pt_prep** out);
void pt_prep_destroy(pt_prep* prep);
int pt_scene_create_from_prep(const pt_prep* prep, int device, pt_scene** out);

/* Number of pixels pt_render writes for (profile, opts). */
uint64_t pt_local_pixel_count(const pt_profile* profile, const pt_opts* opts);
/* local packed index -> global pixel index i = x + y*W; out has
 * pt_local_pixel_count entries.  Host helper (no GPU needed). */
int pt_local_pixel_map(const pt_profile* profile, const pt_opts* opts, uint32_t* out);

/* Renderer::render (renderer/mod.rs:76-169) for this call's pixels.
 * rgb8:  n_local*3 bytes, tone-mapped + gamma + u8 (mod.rs:335-353)  [may be NULL]
 * accum: n_local*3 f32, SUM of sample radiance before the division by
 *        `samples` (the reference's `buffer`, mod.rs:81,130)          [may be NULL]
 * Both are HOST pointers; the call blocks until the image is complete. */
int pt_render(const pt_scene* scene, const pt_profile* profile, const pt_opts* opts,
              uint8_t* rgb8, float* accum);

/* Same, with DEVICE pointers (hipMalloc'd or torch CUDA tensors) and a
 * hipStream_t passed as void*.  Asynchronous with respect to the host
 * unless PT_FLAG_TIMING is set; the caller synchronises the stream. */
int pt_render_device(const pt_scene* scene, const pt_profile* profile, const pt_opts* opts,
                     void* d_rgb8, void* d_accum, void* hip_stream);

/* Scatter packed per-rank framebuffers (as produced by an all-gather of
 * equal-sized, zero-padded packed slices; slice r starts at r*slice_pixels)
 * into a row-major W*H image.  elem_bytes = 3 (rgb8) or 12 (f32 rgb).
 * Device pointers; hip_stream as above. */
int pt_assemble_tiles(const pt_profile* profile, uint32_t shard_count, uint32_t tile_w,
                      uint32_t tile_h, uint64_t slice_pixels, uint32_t elem_bytes,
                      const void* d_gathered, void* d_image, void* hip_stream);

/* The exchange step of a multi-GPU render inside the library: one RCCL ncclAllGather over xGMI of
 * the packed, zero-padded u8 (or f32) framebuffer slices + the scatter of pt_assemble_tiles, on `hip_stream`.
 * Communicators: one process per GPU - rank 0 calls pt_comm_unique_id, the host's own bootstrap (MPI, a TCP
 * store, torch.distributed) hands the 128 bytes to the other ranks, every rank calls pt_comm_create; one process
 * driving several GPUs - pt_comm_create_all (ncclCommInitAll; `out` receives n handles, devices must be distinct).
 * librccl.so is loaded on first use.  The reference has no counterpart (single process, rayon). */
enum { PT_COMM_ID_BYTES = 128 };
typedef struct pt_comm pt_comm;
int pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES]);
int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], int rank, int size, int device, pt_comm** out);
int pt_comm_create_all(const int* devices, int n, pt_comm** out);
void pt_comm_destroy(pt_comm* comm);
/* d_local: this rank's slice_pixels * elem_bytes packed slice; d_gathered: scratch of size * that; d_image: the
 * row-major W*H*elem_bytes frame, complete on every rank when the stream has drained.  Device pointers. */
int pt_gather_tiles(pt_comm* comm, const pt_profile* profile, uint32_t tile_w, uint32_t tile_h,
                    uint64_t slice_pixels, uint32_t elem_bytes, const void* d_local, void* d_gathered,
                    void* d_image, void* hip_stream);

/* Renderer::render on this rank's shard + the gather, host-buffer form: renders (profile, opts) on the device,
 * all-gathers the slices (slice_pixels = the largest pt_local_pixel_count of any rank) and, when rgb8_frame is not
 * NULL, copies the complete row-major RGB8 frame (W*H*3 bytes) to the host.  Every rank of the communicator must
 * call it; opts->shard_rank / shard_count must be the communicator's rank / size. */
int pt_render_gathered(const pt_scene* scene, pt_comm* comm, const pt_profile* profile, const pt_opts* opts,
                       uint64_t slice_pixels, uint8_t* rgb8_frame);

/* `--debug-textures` (src/renderer/debug_renderer.rs:11-105): one pixel-centre primary ray per
 * pixel, first entry of ray_cast() only, seven RGB8 planes of width*height*3 bytes each, in this
 * order: normal (n*0.5+0.5), albedo, opacity, metalness, roughness, emissive, ior/3 — each channel
 * `(v * 255.) as u8`, pixels without a hit stay 0.  *any_hit = 0 means no pixel hit anything (the
 * reference then writes no file at all).  planes is a HOST pointer. */
enum { PT_DEBUG_PLANES = 7 };
int pt_debug_render(const pt_scene* scene, uint32_t width, uint32_t height, uint8_t* planes, int* any_hit);

/* ------------------------------------------------------------------ */
/* measurement                                                         */
/* ------------------------------------------------------------------ */

typedef struct pt_timing {
    uint32_t launches;        /* launches of the dominant kernel in the last render
                                 (k_wf_trace for the wavefront integrator)         */
    float integrate_ms;       /* sum of their HIP-event durations                  */
    float postprocess_ms;     /* tone-map kernel                                   */
    float total_ms;           /* first launch -> last kernel done                  */
    float generate_ms;        /* per-stage sums (wavefront integrator)             */
    float trace_ms;
    float shade_ms;
    float shadow_ms;
    float accumulate_ms;
    uint32_t stage_launches;  /* all integrator launches                           */
    uint32_t bounce0_launches; /* launches of the fused bounce-0 kernel (k_wf_shade<GRID >= 2>: ChaCha block,
                                  camera cast, shading, shadow casts); 0 when the scene has no such kernel */
    float bounce0_ms;         /* sum of their HIP-event durations                  */
} pt_timing;

/* Exact work counters from the instrumented variant (PT_FLAG_COUNTERS);
 * they feed the algorithmic-bytes formula of SURVEY §8-d. */
typedef struct pt_counters {
    uint64_t samples;         /* path samples started                  */
    uint64_t segments;        /* closest-hit ray casts (R_seg)         */
    uint64_t shadow_rays;     /* get_light_info evaluations (R_sh)     */
    uint64_t nodes_visited;   /* KD nodes touched (V)                  */
    uint64_t tris_tested;     /* primitive tests (T)                   */
    uint64_t shaded_hits;     /* material fetches (H)                  */
    uint64_t rng_draws;
    uint64_t restarts;        /* alpha-walk continuation casts         */
    uint64_t max_nodes_per_cast;   /* longest single KD walk (wavefront integrator) */
    uint64_t casts_over_1k_nodes;
    uint64_t trace_nodes;          /* share of nodes_visited / tris_tested spent in  */
    uint64_t trace_tris;           /* closest-hit casts (the rest: shadow casts)      */
    uint64_t shadow_skipped;       /* of shadow_rays: not cast, because the light's BRDF term is exactly 0
                                    * and the visibility cannot change the colour                     */
    uint64_t bounce0_hits;         /* camera rays that hit a surface (shaded at bounce 0)             */
    uint64_t bounce0_shadow_rays;  /* shadow rays cast by the fused bounce-0 kernel (k_wf_shade<GRID >= 2>)  */
    uint64_t bounce0_tris;         /* primitive tests of that kernel (camera casts + shadow casts)    */
    uint64_t grid_tris;            /* of tris_tested: made through origin grids (camera / point lights) */
    uint64_t bounce0_cam_tris;     /* of bounce0_tris: the camera casts (also part of trace_tris)       */
    uint64_t deferred_casts;       /* closest-hit casts finished by k_wf_trace_wide (drain phase of k_wf_trace) */
    uint64_t exact_casts;          /* closest-hit casts k_wf_trace left to k_wf_trace_exact: rays whose direction has a component
                                    * below 8e-4, which the wavefront walker's slack does not cover (csrc/pt_integrator.h) */
    uint64_t masked_casts;         /* of segments: ray_cast calls of bounces >= 1 that were NOT cast because the escape mask of the
                                    * primitive the ray leaves proves them empty (csrc/pt_escape.h) */
    uint64_t bounce0_masked;       /* of masked_casts: found by the bounce-0 kernel (paths that never enter a queue) */
} pt_counters;

int pt_get_timing(const pt_scene* scene, pt_timing* out);
int pt_get_counters(const pt_scene* scene, pt_counters* out);
/* The camera-grid cull of the last frame rendered with this scene (synchronises the device): the number of 8x8 pixel
 * blocks of the rank's tiles and how many of them no camera ray can hit anything in - their samples are the background
 * (Renderer::render's miss branch, renderer/mod.rs:184-186) without an RNG block or a cast.  0 blocks: no cull ran
 * (no camera grid, PT_FLAG_NO_GRIDS / PT_FLAG_MEGAKERNEL / PT_FLAG_COUNTERS, PT_CAM_CULL=0). */
int pt_get_cull_stats(const pt_scene* scene, uint32_t* n_blocks, uint32_t* n_empty);

/* Scene statistics after the KD build. */
typedef struct pt_scene_info {
    uint64_t n_prims;         /* triangles + spheres              */
    uint64_t n_kd_nodes;
    uint64_t n_kd_leaves;
    uint64_t n_leaf_refs;     /* primitive references in leaves   */
    uint32_t kd_depth;
    uint32_t has_translucent; /* any opacity != 1 or opacity texture */
    float kd_build_seconds;
    float upload_seconds;
    uint64_t device_bytes;
    /* origin grids (cube maps of primitive lists around the camera / the point lights, csrc/pt_grid.h) */
    uint32_t cam_grid_res;    /* cells per face edge; 0 = camera rays use the KD-tree */
    uint32_t light_grids;     /* lights whose shadow rays use a grid (all or none)    */
    uint64_t grid_refs;       /* list entries of all grids                            */
    float grid_build_seconds;
    /* escape masks (csrc/pt_escape.h): proofs of misses for the rays that leave a primitive.  Built on the device when the
     * scene is about to render its third frame of the default pipeline (PT_ESCAPE_AFTER; a one-shot render is better off
     * without them) or when pt_scene_escape_copy asks for them: zero until then */
    float escape_build_seconds;
    uint32_t escape_prims;      /* primitives with a mask (the others: all directions "may hit") */
    float escape_clear_fraction; /* of those primitives' 384 direction cells: proven empty */
    /* the path queues of the LAST frame rendered from this scene (csrc/pt_gpu.hip, "frame plan"): the first frame of a
     * configuration runs in chunks of a fixed budget (PT_QUEUE_GIB, default 8; later frames: what their records need, at most PT_QUEUE_STEADY_GIB, 32) with every queue as long as the chunk and
     * counts what each bounce produces; later frames of the same configuration get queues of exactly those lengths */
    uint64_t queue_bytes;        /* device memory held by the queues, hit / shadow records, RNG planes and lists */
    uint32_t queue_chunk_items;  /* work items (pixel samples) per pass over the bounces                        */
    uint32_t frame_planned;      /* 1: sized from an earlier frame's counts; 0: first frame of its configuration */
} pt_scene_info;
int pt_scene_get_info(const pt_scene* scene, pt_scene_info* out);

/* Experiment (tools/cu_mask_pipelines.py): confine the scene's own streams to the compute units of
 * `mask` (hipExtStreamCreateWithCUMask; bit i of word i/32 = CU i) and size its grids for their number, so that two
 * scenes render disjoint tile sets of one frame side by side without taking each other's wave slots.  Before the first
 * render of the scene.  pt_stream_create_cu_mask makes the caller's stream for pt_render_device the same way.
 * Experiment only (measured, rejected: DESIGN.md section 8), not covered by the test suite: on ROCm 7.2 a process that has
 * created and destroyed CU-masked streams crashed inside the runtime when a later scene's kernels were first launched with
 * little device memory left (the out-of-memory fallback test run after a masked render; same frames bit for bit otherwise).
 * The reference has no counterpart. */
int pt_scene_set_cu_mask(pt_scene* scene, const uint32_t* mask, uint32_t n_words);
int pt_stream_create_cu_mask(int device, const uint32_t* mask, uint32_t n_words, void** out_stream);
int pt_stream_destroy(void* stream);

/* ------------------------------------------------------------------ */
/* test hooks (parity tests call the device code piecewise)            */
/* ------------------------------------------------------------------ */

/* One record per ray: the reference's first entry of ray_cast()
 * (renderer/utils.rs:11-21): prim = global primitive index in (model,
 * triangle) order (spheres count as one primitive each, interleaved in
 * model order), or -1 for a miss. */
typedef struct pt_hit {
    int32_t prim;
    int32_t flags;   /* bit0 backface (det<0), bit1 sphere, bit2 sphere exit hit */
    float dist;
    float u;
    float v;
} pt_hit;

/* rays: n x 6 f32 (origin3, direction3), HOST pointers. */
int pt_trace_rays(const pt_scene* scene, const float* rays, uint64_t n, pt_hit* out);

/* The first entry of ray_cast() through the wavefront integrator's own cast kernel (k_wf_trace; the rays of bounces
 * >= 1 of every frame go through it).  mode bit 0: start at the home node of the primitive the ray leaves
 * (start_prims[i], entry lists); bit 1: hand every cast to the cooperative kernel k_wf_trace_wide; bit 2 (study switch):
 * WITHOUT the hand-over of the rays the walker's slack does not cover (a direction component below 8e-4) to k_wf_trace_exact. */
int pt_trace_rays_wavefront(const pt_scene* scene, const float* rays, const uint32_t* start_prims, uint64_t n,
                            uint32_t mode, pt_hit* out);

/* The origin grids of a scene as the DEVICE built them (csrc/pt_grid_build.h): which = 0 the camera grid, 1 + i the grid
 * of light i.  pt_scene_grid_header fills the header fields of a pth_origin_grid (include/pthost.h; enabled = 0: the
 * scene has no such grid; cell_off / refs are left null), pt_scene_grid_copy copies the n_cells + 1 cell offsets and
 * the n_refs list entries into caller-provided host arrays - for the tests, which compare them with the host builder's
 * (pth_origin_grid_build: the same lists byte for byte) and run the conservativeness checks on them. */
/* The escape masks of a scene (csrc/pt_escape.h) as the device built them: 80 bytes per primitive - f32 normal (0 0 0: the
 * primitive has no mask), f32 v0, two spare words, then six faces x 64 direction bits (bit = row * 8 + column, set = "may
 * hit").  For the tests, which aim rays through clear cells and require brute force to find nothing. */
int pt_scene_escape_copy(const pt_scene* scene, void* out, uint64_t bytes);

struct pth_origin_grid;
struct pth_grid_ref;
int pt_scene_grid_header(const pt_scene* scene, uint32_t which, struct pth_origin_grid* out);
int pt_scene_grid_copy(const pt_scene* scene, uint32_t which, uint32_t* cell_off, struct pth_grid_ref* refs);

/* Up to max_hits hits per ray in the reference's sorted order (all hits,
 * stable by (dist, primitive order)); counts[i] = number written. */
int pt_trace_rays_all(const pt_scene* scene, const float* rays, uint64_t n, uint32_t max_hits,
                      pt_hit* out, uint32_t* counts);

/* Triangle::intersect (internal/triangle.rs:37-82) on the device for n
 * independent (ray, triangle) pairs: rays n x 6, tris n x 9 (v0,v1,v2). */
int pt_intersect_triangles(int device, const float* rays, const float* tris, uint64_t n,
                           pt_hit* out);

/* First n_words of StdRng::seed_from_u64(seed) (rand 0.8.5 = ChaCha12,
 * PCG32 key expansion) generated ON THE DEVICE for each seed. */
int pt_rng_words(int device, const uint64_t* seeds, uint64_t n_seeds, uint32_t n_words,
                 uint32_t* out);

/* Device libm restatements evaluated on the GPU (bit-exactness checks
 * against glibc): fn 0 = powf(x, 1/2.2f), 1 = acosf, 2 = sinf, 3 = cosf. */
int pt_eval_math(int device, int fn, const float* x, uint64_t n, float* out);

/* Measurement aid for the roofline ("achievable-copy figure with a
 * stream kernel on the box"): copies `bytes` of device memory with a plain
 * 16-byte-per-lane grid-stride kernel `reps` times and returns the best
 * read+write rate in GB/s (2 * bytes / time).  No reference counterpart. */
int pt_measure_copy_bandwidth(int device, uint64_t bytes, uint32_t reps, double* gb_per_s);

/* Second yardstick: scattered loads.  Every lane of every wavefront reads `bytes_per_load` (8 or 16) bytes at
 * pseudo-random 16-byte-aligned offsets of a `table_bytes` table, `loads_per_lane` times with eight independent
 * loads in flight; returns lane-loads per second in units of 1e9.  This is the access pattern of the KD walk, and
 * its ceiling (not HBM bandwidth) is what k_wf_trace / k_wf_shadow run against.  No reference counterpart. */
int pt_measure_gather_rate(int device, uint64_t table_bytes, uint32_t bytes_per_load, uint32_t loads_per_lane,
                           double* giga_loads_per_s);

const char* pt_last_error(void);
const char* pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */

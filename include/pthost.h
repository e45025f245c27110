/*
 * pthost.h — host-side surface that "stays" either side of the hot path
 * (libpthost.so; no HIP dependency): the ISF scene loader, profile.yml
 * parser, PNG codec, the glTF converter, the deterministic PS5 stand-in
 * scene generator, the KD-tree builder and the origin-grid builder.  The `path-tracer` CLI, the tests and bench.py use it to
 * produce the flat pt_scene_desc that pt_scene_create() consumes.
 *
 * Reference counterparts:
 *   pth_scene_load_isf   src/scene/mod.rs:16-22, src/scene/isf.rs:5-142,
 *                        src/scene/internal/{mod,model,material,texture_bank}.rs
 *   pth_profile_load     src/config/profile.rs:10-40, resolution.rs:3-16
 *   pth_png_*            image crate: open().into_rgb8()/into_luma8(), RgbImage::save
 *   pth_kd_build         kdtree-ray KDTree::build (internal/mod.rs:42, model.rs:96)
 *   pth_convert_gltf     src/scene/gltf.rs:146-265 (convert_gltf_to_isf)
 */
#ifndef PTHOST_H
#define PTHOST_H

#include "ptgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* An owned scene: `desc` points into storage owned by the handle. */
typedef struct pth_scene pth_scene;

int pth_scene_load_isf(const char* path, pth_scene** out);
void pth_scene_free(pth_scene* s);
const pt_scene_desc* pth_scene_desc(const pth_scene* s);

/* Deterministic synthetic stand-in for the (unpublished) PS5 scene
 * (SURVEY §8-d): ground quad + two tessellated curved shells + an emissive
 * strip + one point light, black background, fov 0.6911112.
 *   target_tris  approximate triangle count (500000 for BASELINE cfg 3/4)
 *   seed         PCG32 seed for the vertex jitter (0 for the BASELINE configs)
 *   flags        bit0: shells get opacity.factor 0.5 + a 1024^2 checker
 *                opacity texture (BASELINE cfg 5)
 *                bit1: procedural textures of every other kind: the shells get a
 *                normal map, metalness and roughness textures (one of them an
 *                albedo texture too), the core an emissive texture
 *                bit2: four walls and a ceiling close the scene into a room
 *                (no path escapes: the bounce loop runs to its end)
 *                bit3: the framing of the reference's PS5 render (readme/ps5_b5_s128.png:
 *                38.8 % of its 8x8 pixel blocks are sky): a 12.94 x 12.94 ground seen
 *                corner-on from above, the object a quarter of the frame, the light
 *                behind it to the left (host/scene_gen.cpp has the fitted recipe) */
int pth_scene_generate_ps5(uint64_t target_tris, uint64_t seed, uint32_t flags, pth_scene** out);

/* Write a scene as ISF JSON (+ textures as PNG next to it). */
int pth_scene_save_isf(const pth_scene* s, const char* dir);

/* `convert <INPUT> <OUTPUT>` (src/scene/gltf.rs:146-265): glTF 2.0 (.gltf / .glb) -> OUTPUT/scene.isf + PNG
 * textures; scene 0, first camera, KHR_lights_punctual lights (spot -> point, size 0.1). */
int pth_convert_gltf(const char* input, const char* output_dir);

/* profile.yml: every key optional; unknown keys ignored. path == NULL gives
 * Profile::default(). */
int pth_profile_load(const char* path, pt_profile* out);
int pth_profile_parse(const char* yaml_text, pt_profile* out);

/* PNG. Decoded images are 8-bit; want_channels 1 (into_luma8), 3
 * (into_rgb8) or 4 (into_rgba8: the glTF converter splits base-colour
 * textures into rgb + alpha). *pixels is malloc'd; free with pth_free. */
int pth_png_read(const char* path, uint32_t want_channels, uint32_t* w, uint32_t* h,
                 uint8_t** pixels);
int pth_png_decode(const uint8_t* data, size_t len, uint32_t want_channels, uint32_t* w,
                   uint32_t* h, uint8_t** pixels);
int pth_png_write_rgb8(const char* path, uint32_t w, uint32_t h, const uint8_t* rgb);
void pth_free(void* p);

/* ------------------------------------------------------------------ */
/* KD-tree (single tree over every primitive of every model)           */
/* ------------------------------------------------------------------ */

/* 8-byte node (layout after pbrt's KdAccelNode):
 *   interior: w0 = split position (f32 bits), w1 = (above_child << 2) | axis,
 *             the below child is the next node
 *   leaf:     w0 = first leaf-reference index,  w1 = (n_refs << 2) | 3 */
typedef struct pth_kd_node {
    uint32_t w0;
    uint32_t w1;
} pth_kd_node;

typedef struct pth_kdtree {
    uint64_t n_nodes;
    uint64_t n_refs;
    uint64_t n_leaves;
    uint32_t depth;
    uint32_t _pad;
    float bounds_min[3];
    float bounds_max[3];
    pth_kd_node* nodes;     /* malloc'd */
    uint32_t* refs;         /* malloc'd: primitive ids, leaf after leaf */
    double build_seconds;
    /* surface-area expectation for a random line through the root box:
     * interior nodes visited and primitive tests per ray */
    double expected_nodes;
    double expected_tests;
} pth_kdtree;

/* Primitive ids: one id per triangle and per sphere, in model order (a
 * sphere takes one id at its model's position) — the reference's tie order
 * for equal distances. */
uint64_t pth_prim_count(const pt_scene_desc* desc);
int pth_kd_build(const pt_scene_desc* desc, pth_kdtree* out);
void pth_kd_free(pth_kdtree* kd);

/* ------------------------------------------------------------------ */
/* Origin grid: candidate filter for rays through one point            */
/* ------------------------------------------------------------------ */

/* One list entry: primitive id (bit 31 = sphere) and a LOWER bound of the distance from the grid's
 * origin to any point at which a ray can hit the primitive. */
typedef struct pth_grid_ref {
    uint32_t prim;
    float mindist;
} pth_grid_ref;

/* Cube map of primitive lists around `origin` (host/origin_grid.cpp): face f = 2 * axis + (negative side),
 * cell (iu, iv) of face f at cell_off[(f * res + iv) * res + iu], iu = floor((w[b] / |w[a]| + 1) * res / 2)
 * with b = (a + 1) % 3, c = (a + 2) % 3 for a direction w whose largest component is w[a].  refs[0 .. n_global)
 * are tested by every ray; a cell's list is refs[cell_off[c] .. cell_off[c + 1]), ascending (mindist, prim).
 * Stands in for kdtree-ray (src/renderer/utils.rs:13, src/scene/internal/model.rs:67-68) for camera rays
 * (src/renderer/mod.rs:114-124) and point-light shadow rays (src/renderer/mod.rs:301-331). */
typedef struct pth_origin_grid {
    float origin[3];
    uint32_t res;            /* cells per face edge */
    uint64_t n_cells;        /* 6 * res * res */
    uint64_t n_refs;         /* entries of refs, the global block included */
    uint32_t n_global;
    uint32_t enabled;        /* 0: not built (too many global primitives, non-finite origin): use the KD-tree */
    uint32_t max_cell_refs;
    float ray_offset;        /* distance by which a ray may miss the origin (0 for a camera) */
    uint32_t* cell_off;      /* malloc'd, n_cells + 1 */
    pth_grid_ref* refs;      /* malloc'd */
    double build_seconds;
    /* kind 1: orthographic grid for rays of ONE direction (pth_ortho_grid_build): res x res cells over the plane
     * (axis_u, axis_v), cell (iu, iv) = floor(((p . axis_u) - u0) * cells_per_unit), ... of the ray's origin p;
     * axis_w = the rays' direction (unit); a list entry's `mindist` is MINUS an upper bound of the primitive's depth
     * along axis_w: a ray starting at depth z tests the entries with key <= -z (nothing else lies ahead of it). */
    uint32_t kind;
    float axis_u[3], axis_v[3], axis_w[3];
    float u0, v0, cells_per_unit;
} pth_origin_grid;

/* res = 0 chooses the resolution from the primitive count (PT_OG_RES overrides).  ray_offset: 0 for rays that
 * start exactly AT origin; for shadow rays the largest distance between the ray's line and the origin.
 * max_dir_len: upper bound of the length of the ray directions that will be looked up (Triangle::intersect
 * takes the direction as it is: a camera matrix with a scale gives directions longer than 1). */
int pth_origin_grid_build(const pt_scene_desc* desc, const float origin[3], uint32_t res, float ray_offset,
                          float max_dir_len, pth_origin_grid* out);
/* The same for rays that all have `direction` (not necessarily unit: the shadow rays of a directional light,
 * src/renderer/mod.rs:283-299, which run along -light.direction with no distance limit). */
int pth_ortho_grid_build(const pt_scene_desc* desc, const float direction[3], uint32_t res, pth_origin_grid* out);
/* The resolution the builders choose for res == 0 (4 sqrt(n) cells per face edge, a power of two in 32 ... 8192; PT_OG_RES
 * overrides): the caller budgets the memory of all grids of a scene with it (pt_scene_create, PT_OG_BUDGET_GIB). */
uint32_t pth_origin_grid_auto_resolution(uint64_t n_prims);
void pth_origin_grid_free(pth_origin_grid* g);

const char* pth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* PTHOST_H */

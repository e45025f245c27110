// Device-side scene layout (HBM) shared by the kernels and the host API.
//
// All arrays are read-only during a render and sized for one MI355X
// (288 GB HBM3E): the whole scene, KD-tree and textures are replicated on
// every GPU.
//
//   kd_nodes    8 B / node   interior: (split, pair << 2 | axis), children = nodes pair and pair + 1;
//               leaf: (first leaf record, n << 2 | 3).  Treelet order (4-level subtrees share a
//               128-B line); the host builder's DFS array (include/pthost.h) is permuted on upload
//   leaf_prims  48 B / leaf reference, stored leaf after leaf so a leaf's
//               primitives are one contiguous, 16-B aligned run:
//                 q0 = (v0.x, v0.y, v0.z, bits(prim_id | kind<<31))
//                 q1 = (e1.x, e1.y, e1.z, e2.x)      e1 = v1 - v0 (f32, as
//                 q2 = (e2.y, e2.z, 0, 0)            triangle.rs:43-44 computes it)
//               spheres: q0 = (c.x, c.y, c.z, bits(prim | 1<<31)), q1.x = radius
//   prim_attr   64 B / primitive, indexed by global primitive id (shading only):
//                 a0 = (n0.xyz, uv0.x) a1 = (n1.xyz, uv0.y) a2 = (n2.xyz, uv1.x)
//                 a3 = (uv1.y, uv2.x, uv2.y, bits(model))
//               spheres: a0 = (c.xyz, radius), a3.w = bits(model)
//   prim_pos    48 B / primitive (v0, e1, e2 as in leaf_prims) for the tangent
//               frame of normal-mapped materials and the test hooks
//   materials   64 B / model (pt_material as is)
//   textures    pt_texture table + one u8 texel blob
//   srgb_lut    256 f32: powf(c/255, 2.2) evaluated by the host libm
//               (Material::get_albedo, internal/material.rs:132-146)
#pragma once
#include <stdint.h>

#include "ptgpu.h"

#define PT_PRIM_SPHERE 0x80000000u
#define PT_PRIM_INDEX(pid) ((pid) & 0x3fffffffu)
#define PT_KD_STACK 64

struct DevLight {
    int32_t kind;
    float vec[3];
    float color[3];
    uint32_t tame;   // 1: every |colour component| < 1e30 (so colour / (4 pi d^2) stays finite for d > 1e-3)
};

// Origin grid (host/origin_grid.cpp, csrc/pt_grid.h): cube map of primitive lists around one point.
//   cell_off  4 B / cell (+1): first reference of the cell; 6 faces x res x res cells
//   refs      8 B / reference: (primitive id | sphere bit, bits(lower bound of the distance from the origin));
//             the first n_global references are tested by every ray
// res == 0: no grid (the casts use the KD-tree).
// kind 1 (orthographic, rays of one direction: the shadow rays of a directional light): res x res cells over the
// plane (axis_u, axis_v); a reference's second word is MINUS an upper bound of the primitive's depth along axis_w.
struct DevGrid {
    const uint32_t* cell_off;
    const uint2* refs;
    uint32_t res;
    uint32_t n_global;
    float half_res;
    uint32_t kind;
    float axis_u[3], axis_v[3], axis_w[3];
    float u0, v0, cells_per_unit;
};

struct DevScene {
    const uint2* kd_nodes;
    const float4* leaf_prims;
    const float4* prim_attr;
    const float4* prim_pos;
    const uint2* entry_lists;       // ancestors of the primitives' home nodes (trav_enter, pt_wavefront.h)
    const uint32_t* prim_entry;     // per primitive: list offset << 6 | entries (0: start at the root)
    const pt_material* materials;   // indexed by MODEL (material resolved on upload)
    const pt_texture* textures;
    const uint8_t* texels;
    const float* srgb_lut;
    const DevLight* lights;
    uint32_t n_lights;
    uint32_t n_prims;
    uint32_t n_nodes;
    uint32_t n_node_slots;   // entries of kd_nodes (device layout, >= n_nodes)
    uint32_t has_translucent;
    float bounds_min[3];
    float bounds_max[3];
    float slab_min[3];       // the EXACT bounding box of the scene (union of Model::bound(), model.rs:76-86): the box of
    float slab_max[3];       // kdtree-ray's slab test (scene_slab, pt_integrator.h)
    // camera (internal/camera.rs:36-48): columns of the transform, tan(fov/2) from host tanf
    float cam_c0[3], cam_c1[3], cam_c2[3], cam_c3[3];
    float tan_half_fov;
    float background[3];
    // origin grids: camera rays; shadow rays of point lights (light_grids[light], all or none: all_lights_gridded)
    DevGrid cam_grid;
    const DevGrid* light_grids;
    uint32_t all_lights_gridded;
    float light_grid_max_normal2;   // |surface normal|^2 up to which a shadow ray stays within the grids' margin
    // escape masks (pt_escape.h): 80 bytes per primitive - plane (normal, v0) and 6 x 64 direction bits; null: none
    const float4* escape;
};

struct RenderParams {
    uint32_t width, height;
    uint32_t samples;       // profile.samples (seed stride, final division)
    uint32_t bounces;
    uint32_t sample_begin;  // this launch renders samples (sample_begin, sample_end], 1-based
    uint32_t sample_end;
    int32_t tonemap;
    // pixel sharding (pt_opts)
    uint32_t shard_rank, shard_count, tile_w, tile_h;
    uint32_t tiles_x, tiles_y;
    uint32_t n_local;       // pixels rendered by this call
    uint32_t tile_k_base;      // != 0: tile_table[tile_k_base + lt] = global number of local tile lt (sharded renders)
    uint32_t tile_order_base;  // != 0: tile_table[tile_order_base + j] = the j-th local tile the wavefront integrator visits
    // exact division of the work-item decoding by multiply-high (pt_fastdiv, dividends < 2^27):
    // the sample batch, the 8x8 blocks of a tile, the 8-pixel columns of a tile, the tile columns of the image
    uint32_t div_batch[2], div_tile_blocks[2], div_tile_cols[2], div_tiles_x[2];   // {magic, shift}
};

// q = n / d for n < 2^27 with {magic, shift} from pt_fastdiv_make(d): magic = ceil(2^(32 + shift) / d),
// shift = max(0, ceil(log2 d) - 5); the rounding error magic * d - 2^(32 + shift) < d times n < 2^27 stays below
// 2^(32 + shift), so the quotient is exact.  d = 1 is encoded as magic 0.
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t pt_fastdiv(uint32_t n, const uint32_t (&ms)[2]) {
    return ms[0] ? __umulhi(n, ms[0]) >> ms[1] : n;
}
#endif
static inline void pt_fastdiv_make(uint32_t d, uint32_t (&ms)[2]) {
    if (d <= 1u) {
        ms[0] = 0u;
        ms[1] = 0u;
        return;
    }
    uint32_t s = 0;
    while ((1u << s) < d) ++s;                       // ceil(log2 d)
    const uint32_t shift = s > 5u ? s - 5u : 0u;
    const unsigned long long pow = 1ull << (32u + shift);
    ms[0] = (uint32_t)((pow + d - 1ull) / d);        // < 2^32: d > 2^(s-1) and shift >= s - 5 (d >= 2: pow / d <= 2^31 ...)
    ms[1] = shift;
}

// Work counters (PT_FLAG_COUNTERS variant only).
struct DevCounters {
    unsigned long long samples, segments, shadow_rays, nodes_visited, tris_tested, shaded_hits, rng_draws,
        restarts, max_nodes_per_cast, casts_over_1k_nodes, trace_nodes, trace_tris, shadow_skipped,
        bounce0_hits, bounce0_shadow_rays, bounce0_tris, grid_tris, bounce0_cam_tris, deferred_casts, exact_casts, masked_casts, bounce0_masked;
    unsigned long long stamps[8];  // diagnostics: phase cycles of a -DWF_STAMPS build; else [0,1,3,4,5] rounds / steps / time of k_wf_trace_wide
    unsigned long long cast_hist[16];  // k_wf_trace casts by nodes visited, 64 per bin (PT_DEBUG_HIST prints it)
#ifdef WF_EXIT_TIMES
    // diagnostic build only: s_memrealtime (100 MHz) at the start of every k_wf_trace launch (minimum over
    // workgroups) and at the exit of every wavefront, per launch (bounce): the shape of the drain phase
    unsigned long long launch_start[8];
    unsigned long long wave_exit[8][8192];
    unsigned long long wave_queue_done[8][8192];   // when the wavefront found the queue exhausted
#endif
};

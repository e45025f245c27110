// Kernels around the origin grids (pt_grid.h) that are stages of their own.  With a camera grid AND light grids
// k_wf_shade<GRID = 2> does all casts of bounce 0 inline; k_og_primary serves scenes with a camera grid only
// (directional lights).
//
//   k_og_primary          bounce 0 of k_wf_trace: closest hit (+ alpha walk) of every camera ray -> hits[]
//   k_og_shadow           k_wf_shadow through the light grids (bounces >= 1)
//   k_og_shadow_offgrid   surfaces whose normal is too long for the grids' margin: get_light_info on the KD-tree
#pragma once
#include "pt_wavefront.h"

// ---------------------------------------------------------------------------
// bounce 0: closest hit + alpha walk (mod.rs:182-205) of every camera ray of the chunk
// ---------------------------------------------------------------------------
template <bool ALPHA, bool COUNT>
__global__ __launch_bounds__(256) void k_og_primary(DevScene S, WfParams W, const uint32_t* __restrict__ tile_offsets,
                                                    uint4* __restrict__ hits, const uint4* __restrict__ rng_planes,
                                                    uint32_t* __restrict__ draws, DevCounters* __restrict__ gctr) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= W.n_items) return;
    const uint2 sc = *(const uint2*)(rng_planes + i);   // jittered screen position (k_wf_rng)
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    RawHit best;
    bool hit = false;
    uint32_t draw = 2;   // the pixel jitter
    if (sc.x != WF_ITEM_INVALID) {
        f3 o, d;
        primary_from_screen(S, __uint_as_float(sc.x), __uint_as_float(sc.y), o, d);
        const uint32_t cell = og_cell(S.cam_grid, d);
        const float dlen = mag3(d);
        const float kmax = (dlen > 1.0f ? dlen : 1.0f) * 1.00002f;
        if (COUNT) lc.segments++;
        hit = og_next_hit<COUNT>(S, S.cam_grid, cell, o, d, kmax, INFINITY, -INFINITY, 0u, best, lc);
        if (ALPHA) {
            RawHit kept = best;
            bool have_kept = false;
            while (hit) {
                const float opacity = hit_opacity(S, o, d, best);
                if (COUNT) lc.shaded++;
                bool stop = opacity >= 1.f;
                if (!stop && opacity > 0.001f) {
                    WfRng fb;
                    fb.block = 0xffffffffu;
                    const float r = wf_rng_draw(fb, W, tile_offsets, rng_planes, i, draw++);
                    stop = r < opacity;
                    if (COUNT) lc.shadow_rays++;   // (the alpha-draw counter, as in k_wf_trace)
                }
                if (stop) break;
                kept = best;   // skipped: remember it, look for the next entry of the list
                have_kept = true;
                if (COUNT) lc.restarts++;
                hit = og_next_hit<COUNT>(S, S.cam_grid, cell, o, d, kmax, INFINITY, kept.key, kept.ord, best, lc);
            }
            if (!hit && have_kept) {   // every hit skipped: the last one is shaded
                best = kept;
                hit = true;
            }
        }
    }
    wf_store_hit(hits, W.hcap, i, best, hit);
    if (ALPHA) draws[i] = draw;
    if (COUNT) {
        atomicAdd(&gctr->segments, (unsigned long long)lc.segments);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->trace_tris, (unsigned long long)lc.tris);
        atomicAdd(&gctr->grid_tris, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
        if (ALPHA) atomicAdd(&gctr->shaded_hits, (unsigned long long)lc.shaded);
        if (ALPHA) atomicAdd(&gctr->rng_draws, (unsigned long long)lc.shadow_rays);
    }
}

// A finished shadow job: the sample is complete (staging) or the colour of the path's next record is patched.
PT_D void og_retire(f3 color, uint32_t next_idx, uint32_t out_slot, float4* __restrict__ queue_next, uint32_t cap,
                    float* __restrict__ staging) {
    if (next_idx == 0xffffffffu) {
        float* out = staging + (size_t)out_slot * 3;
        out[0] = color.x;
        out[1] = color.y;
        out[2] = color.z;
    } else {
        wf_path_rec(queue_next, cap, next_idx)[1] = make_float4(color.x, color.y, color.z, 0.f);
    }
}

// ---------------------------------------------------------------------------
// shadow: get_light_info (mod.rs:281-333) for every light of every record of the shadow queue, all lights point
// lights with a grid.  Same records in, same arithmetic per light, same order of the additions as k_wf_shadow.
// A surface whose normal is too long for the grids' margin (|n| > 1.5: the shadow ray starts n * 1e-5 off the
// line through the light) is appended to `offgrid` and left to k_og_shadow_offgrid.
// Used for bounces >= 1: there a lean kernel of its own (60 registers, 8 waves per SIMD) next to the 128-register
// shade kernel beats casting inline (MI355X, config 3, bounce 1: 3.66 + 0.74 ms against 5.98 ms).
// ---------------------------------------------------------------------------
// (opaque scenes: held to the 64 registers of 8 waves per SIMD; the rarely taken branch of kdtree-ray's box test would
// otherwise raise the allocation to 70 and cost a wave)
template <bool ALPHA, bool COUNT, bool DIRL>
__global__ __launch_bounds__(256, (!ALPHA && !COUNT && !DIRL) ? 8 : 1) void k_og_shadow(DevScene S, WfParams W, const float4* __restrict__ shadow_q,
                                                   const float4* __restrict__ contrib, float4* __restrict__ queue_next,
                                                   float* __restrict__ staging, uint32_t* __restrict__ offgrid,
                                                   WfCounters* __restrict__ ctr, DevCounters* __restrict__ gctr) {
    const uint32_t n = min(ctr[W.bounce].shadow_count, W.scap);
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    uint32_t n_skipped = 0;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {
        const float4* sq = shadow_q + (size_t)idx * 4;
        const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3];
        const f3 pos = mk3(s0.x, s0.y, s0.z), gn = mk3(s0.w, s1.x, s1.y);
        const f2 uv = {s1.z, s1.w};
        f3 color = mk3(s2.x, s2.y, s2.z);
        const bool sphere = (__float_as_uint(s3.y) & WF_FLAG_SPHERE) != 0;
        if (!(dot3(gn, gn) <= S.light_grid_max_normal2)) {   // (rare: one atomic per surface is fine)
            offgrid[atomicAdd(&ctr[W.bounce].offgrid_count, 1u)] = idx;
            continue;
        }
        for (uint32_t li = 0; li < S.n_lights; ++li) {
            const float4 c = contrib[(size_t)li * W.scap + idx];
            const f3 term = mk3(c.x, c.y, c.z);
            if (S.n_lights > 1 && wf_light_is_moot(S.lights[li], term, pos)) {   // (a single light was already filtered by k_wf_shade)
                if (COUNT) {
                    lc.shadow_rays++;
                    n_skipped++;
                }
                continue;
            }
            const f3 rad = og_light_radiance<ALPHA, COUNT, DIRL>(S, li, pos, gn, uv, sphere, lc);
            if (!(rad.x == 0.f && rad.y == 0.f && rad.z == 0.f)) color = color + mul_ew(term, rad);
        }
        og_retire(color, __float_as_uint(s2.w), __float_as_uint(s3.x), queue_next, W.qcap_out, staging);
    }
    if (COUNT) {
        atomicAdd(&gctr->shadow_rays, (unsigned long long)lc.shadow_rays);
        atomicAdd(&gctr->shadow_skipped, (unsigned long long)n_skipped);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->grid_tris, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
    }
}

// ---------------------------------------------------------------------------
// Surfaces whose normal is too long for the light grids' margin: get_light_info on the KD-tree (light_radiance,
// pt_integrator.h) for all their lights, then the path is retired like k_wf_shadow does.  LIST: the records
// k_og_shadow set aside (offgrid[0 .. offgrid_count)); otherwise the whole shadow queue - what k_wf_shade<GRID>
// still writes - and the jobs k_wf_shadow (KD-tree pipeline) set aside because a shadow ray of theirs needs more slack than the
// wavefront walker gives (slack_is_capped): those carry the light to go on with (s3.z) and the colour so far.  Normally
// there is nothing to do and the launch returns at once.
// ---------------------------------------------------------------------------
template <bool COUNT, bool LIST>
__global__ __launch_bounds__(256) void k_og_shadow_offgrid(DevScene S, WfParams W, const float4* __restrict__ shadow_q,
                                                           const float4* __restrict__ contrib,
                                                           float4* __restrict__ queue_next, float* __restrict__ staging,
                                                           const uint32_t* __restrict__ offgrid,
                                                           WfCounters* __restrict__ ctr, DevCounters* __restrict__ gctr) {
    const uint32_t n = min(LIST ? ctr[W.bounce].offgrid_count : ctr[W.bounce].shadow_count, W.scap);
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    uint32_t n_skipped = 0;
    for (uint32_t j = blockIdx.x * 256u + threadIdx.x; j < n; j += gridDim.x * 256u) {
        const uint32_t idx = LIST ? offgrid[j] : j;
        const float4* sq = shadow_q + (size_t)idx * 4;
        const float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3];
        Surface sf;
        sf.pos = mk3(s0.x, s0.y, s0.z);
        sf.normal = mk3(s0.w, s1.x, s1.y);
        sf.uv = {s1.z, s1.w};
        sf.sphere = (__float_as_uint(s3.y) & WF_FLAG_SPHERE) != 0;
        sf.backface = false;
        sf.model = 0;
        sf.tangent = mk3(0.f, 0.f, 0.f);
        f3 color = mk3(s2.x, s2.y, s2.z);
        for (uint32_t li = LIST ? __float_as_uint(s3.z) : 0u; li < S.n_lights; ++li) {   // (k_wf_shade writes 0 there)
            const DevLight& L = S.lights[li];
            const float4 c = contrib[(size_t)li * W.scap + idx];
            const f3 term = mk3(c.x, c.y, c.z);
            if (S.n_lights > 1 && wf_light_is_moot(L, term, sf.pos)) {   // (a single light was already filtered by k_wf_shade)
                if (COUNT) {
                    lc.shadow_rays++;
                    n_skipped++;
                }
                continue;
            }
            f3 rad, ldir;
            light_radiance<COUNT, true>(S, L, sf, rad, ldir, lc);
            if (!(rad.x == 0.f && rad.y == 0.f && rad.z == 0.f)) color = color + mul_ew(term, rad);
        }
        og_retire(color, __float_as_uint(s2.w), __float_as_uint(s3.x), queue_next, W.qcap_out, staging);
    }
    if (COUNT) {
        atomicAdd(&gctr->shadow_rays, (unsigned long long)lc.shadow_rays);
        atomicAdd(&gctr->shadow_skipped, (unsigned long long)n_skipped);
        atomicAdd(&gctr->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
    }
}

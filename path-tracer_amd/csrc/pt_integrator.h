// Device integrator: KD-tree ray casting, Möller–Trumbore, material fetch,
// Cook–Torrance, bounce loop.  One independent transcription of the
// reference's hot path for gfx950 (the CPU oracle under oracle/ is a second,
// separate one used only by the tests).
//
// Reference map (all under /root/reference/src):
//   kd_traverse / next_hit      renderer/utils.rs:11-21 (ray_cast: all hits sorted by
//                               distance; here: iterate "next hit after (dist, order)")
//   isect_triangle              scene/internal/triangle.rs:37-82
//   isect_sphere                scene/internal/model.rs:26-64
//   make_surface                renderer/hit.rs:100-137, 55-82
//   material_sample             renderer/material_sample.rs:20-40, scene/internal/material.rs:115-214
//   CookTorrance ct_*           renderer/brdf/cook_torrance.rs:10-183, brdf/mod.rs:35-48, utils.rs:34-36
//   light_radiance              renderer/mod.rs:281-333
//   render_path                 renderer/mod.rs:172-278, utils.rs:23-31
//   primary_ray                 renderer/mod.rs:107-124, scene/internal/camera.rs:36-48
//   post_process                renderer/mod.rs:335-353, renderer/tonemap.rs:15-54
#pragma once
#include "pt_device.h"
#include "pt_math.h"
#include "pt_rng.h"

struct LocalCtr {
    uint32_t segments, shadow_rays, nodes, tris, shaded, restarts;
};

// One entry of the reference's sorted hit list.
struct RawHit {
    float key;       // Hit::get_dist(): the sort key of ray_cast (utils.rs:19)
    uint32_t ord;    // tie order: primitive id * 2 (+1 for a sphere's exit hit)
    uint32_t pid;    // primitive id | PT_PRIM_SPHERE
    float u, v;      // barycentrics; spheres: u = ray parameter t of the hit
    uint32_t flags;  // bit0 backface, bit1 sphere, bit2 sphere exit
};

// Slack of the walk along the ray (kd_build.cpp, robustness rules): both children are visited when the plane parameter lies
// within it of the node's interval, and the walk goes on while the next segment starts within it of the best hit.  What it has
// to cover - THE SLOP MODEL every walker of this file is exact against: Triangle::intersect in f32 accepts rays that pass a
// triangle's edge on the OUTSIDE by a DISTANCE s of at most a few 10^-6 of the ray's length (largest seen in all tests and
// stress runs: 3e-6 x length = 50 x 2^-24).  A triangle lives only in the cells its own extent overlaps, so such a ray may run
// past the triangle's cell: it leaves the enclosing box through a face of axis a' - the END of the node's interval - and would
// have entered the triangle's cell s / |d_a| later along the ray, a the axis of the plane in between: a far child may be
// skipped only if the plane lies beyond the interval's end by more than s / |d_a'| + s / |d_a|.
//   * kd_traverse (megakernel, test hooks, k_og_shadow_offgrid) keeps exactly that sum, per interval end and plane, with
//     s = PT_SLACK_K x t, uncapped;
//   * the wavefront walker (pt_wavefront.h trav_step) uses ONE relative slack per ray for every test,
//         PT_SLACK_K * max |1 / d_axis|  (floor PT_SLACK_MIN)                                                  (exit_rel)
//     which covers the sum for s <= PT_SLACK_K / 2 x t = 4e-6 x t.  A ray with a direction component below
//     PT_SLACK_K / PT_SLACK_MAX = 8e-4 (0.24 % of random directions) would need a slack above PT_SLACK_MAX = 0.01 - with such a
//     slack a lane visits both children of almost every node and becomes the longest cast of its launch (uncapped: one shard of
//     eight -4 %, config 5 -4.6 %, the KD-tree-only pipeline -18 %; profiles/r03_experiments.txt item 13).  Round 4: those rays
//     are NOT walked by the wavefront walker at all (slack_is_capped): k_wf_trace / k_wf_shadow hand them over when they fetch
//     them - k_wf_trace_exact, k_og_shadow_offgrid - so PT_SLACK_MAX is a switch-over point between two exact walkers, not a
//     cap on anybody's slack;
//   * kd_traverse_box (k_wf_trace_exact, k_og_shadow_offgrid) fattens the RAY: a box of half-width c + PT_SLACK_K x t around
//     the ray's point at t; whatever the ray's direction, a hit within that of a leaf's box has the leaf visited.
// History: a constant 1e-5 (rounds 1-2) left rays with a component below 0.03 exposed - the grid-vs-KD check on config 5 found
// one in 2e10 casts (d_x = 0.018, needs 1.6e-5 at a z plane); a constant 1e-4 (round 3, first form) still left components of
// 3e-3 ... 0.03: tools/stress_paths.py found a camera ray with d_y = 0.011 through the shared edge of two translucent triangles
// whose second hit the walk lost (needs 2.7e-4; profiles/r03_experiments.txt items 3, 13); round 3 shipped the per-ray slack
// CAPPED at 0.01, a band 100 x narrower but not gone.  Fattening the primitives in the builder instead (PT_KD_PAD=1) costs 4-9 %.
#ifndef PT_SLACK_MIN
#define PT_SLACK_MIN 1e-5f
#endif
#ifndef PT_SLACK_K
#define PT_SLACK_K 8e-6f
#endif
#ifndef PT_SLACK_MAX
#define PT_SLACK_MAX 0.01f
#endif
#define PT_EXIT_ABS 1e-6f
PT_D float exit_rel(float ix, float iy, float iz) {
    return fminf(PT_SLACK_MAX, fmaxf(PT_SLACK_MIN, PT_SLACK_K * fmaxf(fmaxf(fabsf(ix), fabsf(iy)), fabsf(iz))));
}
// The ray would need more than PT_SLACK_MAX (ix, iy, iz = 1 / d as the walker computes them; a NaN or zero component: yes).
// Such a cast never enters the wavefront walker: see the slop model above.
PT_D bool slack_is_capped(float ix, float iy, float iz) {
    return !(PT_SLACK_K * fmaxf(fmaxf(fabsf(ix), fabsf(iy)), fabsf(iz)) <= PT_SLACK_MAX);
}
// ... uncapped: the exact walker's (kd_traverse)
PT_D float restart_param_exact(float t_prev, float ix, float iy, float iz) {
    const float r = PT_SLACK_MIN + PT_SLACK_K * fmaxf(fmaxf(fabsf(ix), fabsf(iy)), fabsf(iz));
    const float t = t_prev - (t_prev * r + PT_EXIT_ABS);   // (r = inf: -inf)
    return t > 0.f ? t : 0.f;
}
// where a cast that continues behind t_prev (alpha walk, next_hit) may start: the exit slack before it
PT_D float restart_param(float t_prev, float ix, float iy, float iz) {
    const float r = exit_rel(ix, iy, iz);
    const float t = t_prev - (t_prev * r + PT_EXIT_ABS);
    return t > 0.f ? t : 0.f;
}

// ---------------------------------------------------------------------------
// KD-tree traversal (front to back).  `leaf(first_ref, n_refs)` tests the
// primitives of one leaf and returns true to stop the traversal; `limit` is
// the largest hit key still of interest (the closest hit so far).  A node
// whose entry parameter, scaled to key units, lies beyond `limit` ends the
// walk.  The stack holds (far child, its exit parameter); the entry parameter
// of a popped node is the exit parameter of the segment just finished.
// ---------------------------------------------------------------------------
// This walker (the megakernel's, the test hooks', the shadow walk of the KD-tree-only paths) keeps the slack EXACTLY where the
// argument above puts it, uncapped: every interval end remembers the relative slack of the axis whose plane (or scene-box
// face) made it - K / |d_axis| - and a plane test adds the tested plane's own; the early exit uses the ray's largest.  It costs
// a few instructions per step and a third stack column, which this path can afford; the wavefront walker (pt_wavefront.h
// trav_step) uses the ray's largest slack, capped (exit_rel), for everything.  tools/stress_paths.py plays the two against each
// other.
template <bool COUNT, class LeafFn>
PT_D void kd_traverse(const DevScene& S, f3 o, f3 d, float t_start, float key_scale, float& limit,
                      LocalCtr& lc, LeafFn&& leaf) {
    const float oa[3] = {o.x, o.y, o.z};
    const float inv[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    const float ra[3] = {PT_SLACK_K * fabsf(inv[0]), PT_SLACK_K * fabsf(inv[1]), PT_SLACK_K * fabsf(inv[2])};   // (inf for d_axis = 0)
    const float rmax = fmaxf(fmaxf(ra[0], ra[1]), ra[2]);
    // clip against the (padded) scene bounds
    float tmin = t_start, tmax = INFINITY;
    float rel_lo = rmax, rel_hi = rmax;   // (a restart's t_start: no axis of its own)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float tn = (S.bounds_min[a] - oa[a]) * inv[a];
        float tf = (S.bounds_max[a] - oa[a]) * inv[a];
        if (tn > tf) {
            float tmp = tn;
            tn = tf;
            tf = tmp;
        }
        if (tn > tmin) {   // NaN (0 * inf) never narrows the interval
            tmin = tn;
            rel_lo = ra[a];
        }
        if (tf < tmax) {
            tmax = tf;
            rel_hi = ra[a];
        }
    }
    if (tmin > tmax) return;

    uint32_t st_node[PT_KD_STACK];
    float st_tmax[PT_KD_STACK], st_rel[PT_KD_STACK], st_tmin[PT_KD_STACK], st_rlo[PT_KD_STACK];
    int sp = 0;
    uint32_t node = 0;
    while (true) {
        uint2 nd = S.kd_nodes[node];
        if (COUNT) lc.nodes++;
        uint32_t axis = nd.y & 3u;
        if (axis != 3u) {
            float split = __uint_as_float(nd.x);
            // select by value, not by operand (see pt_wavefront.h trav_step)
            float tp0 = (split - o.x) * inv[0], tp1 = (split - o.y) * inv[1], tp2 = (split - o.z) * inv[2];
            bool bf0 = (o.x < split) || (o.x == split && d.x <= 0.f);
            bool bf1 = (o.y < split) || (o.y == split && d.y <= 0.f);
            bool bf2 = (o.z < split) || (o.z == split && d.z <= 0.f);
            float tplane = axis == 0 ? tp0 : (axis == 1 ? tp1 : tp2);
            bool below_first = axis == 0 ? bf0 : (axis == 1 ? bf1 : bf2);
            uint32_t below = nd.y >> 2, above = below + 1;  // sibling pair (treelet layout, pt_gpu.hip)
            uint32_t first = below_first ? below : above;
            uint32_t second = below_first ? above : below;
            // A primitive touching the split plane lives on one side only (kd_build.cpp), so a
            // hit AT the plane must see both children: the one-child shortcuts keep the slack's
            // distance from the interval ends (the end's own axis + the tested plane's).
            const float rel_a = axis == 0 ? ra[0] : (axis == 1 ? ra[1] : ra[2]);
            const float s_hi = rel_hi + rel_a + PT_SLACK_MIN, s_lo = rel_lo + rel_a + PT_SLACK_MIN;
            // (a plane behind the origin is out of reach unless the ray runs along it: rel_a > 1 means |d_axis| < PT_SLACK_K)
            if (tplane > tmax + (tmax * s_hi + PT_EXIT_ABS) || (tplane <= 0.f && !(rel_a > 1.0f))) {
                node = first;
            } else if (tplane > 0.f && tplane < tmin - (tmin * s_lo + PT_EXIT_ABS)) {
                node = second;
            } else {  // also taken when tplane or a slack is NaN / inf: visit both (conservative)
                st_node[sp] = second;
                st_tmax[sp] = tmax;
                st_rel[sp] = rel_hi;
                node = first;
                if (!(tplane > 0.f)) {
                    // the ray runs ALONG a plane behind (or through) its origin: it stays on the near side, within the slop
                    // of the far one, for the whole interval - both children get all of it
                    st_tmin[sp] = tmin;
                    st_rlo[sp] = rel_lo;
                } else {
                    // The far child starts where the plane is reached - never beyond the node's own interval (a plane within the
                    // slack PAST tmax must not inflate anybody's interval: nested, that compounds, the start reported for a
                    // later segment overtakes segments still on the stack, and the early exit below drops them:
                    // profiles/r03_experiments.txt item 3) nor before its start - and the slack of that start is the PLANE's
                    // (or the node's own start's, if larger), whichever of the three made the number: a ray that runs along the
                    // plane (rel_a huge) is within reach of the far child long before the point where the parameters say it
                    // crosses.  Round 4: until then the start inherited the slack of the node's END when the plane lay beyond
                    // it, and a point interval lost the slack of its end at the next plane - rays with a component below
                    // ~1e-6 through the edges of axis-aligned quads lost hits (tests/test_gpu_parity.py
                    // test_near_axis_rays_go_through_the_exact_walker found them on alpha_transparency).
                    float far_start = tplane < tmax ? tplane : tmax;
                    if (!(far_start > tmin)) far_start = tmin;
                    st_tmin[sp] = far_start;
                    st_rlo[sp] = fmaxf(rel_a, rel_lo);
                    if (tplane < tmax) {   // the near child ends at the plane (a plane before the interval's start: at the point tmin)
                        tmax = far_start;
                        rel_hi = rel_a;
                    }
                }
                ++sp;
            }
            continue;
        }
        uint32_t n_refs = nd.y >> 2;
        if (n_refs && leaf(nd.x, n_refs)) return;
        // next segment
        while (true) {
            if (sp == 0) return;
            --sp;
            tmin = st_tmin[sp];
            rel_lo = st_rlo[sp];
            node = st_node[sp];
            tmax = st_tmax[sp];
            rel_hi = st_rel[sp];
            // everything from here on starts at tmin or later - each segment by the slack of its own start, at most the ray's
            // largest, earlier for a primitive the ray passes on the outside: stop when even that is beyond the best hit
            // (a ray that runs along a plane - rmax > 1 - may have stacked segments out of order: no early exit for it)
            if (!(rmax > 1.0f) && tmin * key_scale > limit + (limit * (rmax + PT_SLACK_MIN) + PT_EXIT_ABS)) return;
            break;
        }
    }
}

// ---------------------------------------------------------------------------
// The same candidate set by a different argument, for the rays the wavefront walker does not take (slack_is_capped): the RAY
// is fattened instead of the intervals - at parameter t it is a box of half-width g(t) = c + k t around the point o + t d,
// k = PT_SLACK_K per unit of ray length (twice what the wavefront walker's slack covers, slop model above), c = 2^-20 of
// the largest scene coordinate (the rounding of the differences o - v0 the intersection test starts from: ~1e-5 in the
// generated scenes, the size of the normal bias of mod.rs:58).  A node is visited iff the fat ray touches its box: going
// down the tree the interval is clipped by every plane against the cone's OUTER edge - the below child keeps
// {x_a(t) - g(t) <= split}, the above child {x_a(t) + g(t) >= split} - both linear in t, one reciprocal per axis and side,
// made once per ray.  A ray that runs along a plane (|d_a| < k: the two edges of the cone move apart) takes both sides for
// as long as it IS within g of both, not because a relative slack exploded; a hit within g(t_hit) of a leaf's box has that
// leaf visited whatever the direction.  No axis bookkeeping, no special case beyond the sign of the edge's slope.  Depth
// first, nearer child first; a stacked segment that starts beyond the best hit is dropped when it is popped (hits are
// accepted wherever they are found, the result is the minimum of a total order: the visiting order is free).
// (First form, measured: a constant growth g(t_exit) - simpler, but 50 x the normal bias: every ray that leaves a surface at
// a grazing angle walked that surface's leaves for as long as it stayed within g of it.)
// ---------------------------------------------------------------------------
// x, y or z by a lane's axis as two v_cndmask: a `?:` chain (or an array indexed by the axis) becomes a scratch array - three
// dependent scratch loads per use on the hottest path (pt_wavefront.h wf_select has the history)
PT_D float pt_by_axis(uint32_t axis, float x, float y, float z) {
    const unsigned long long ax0 = __builtin_amdgcn_uicmp(axis, 0u, 32), ax1 = __builtin_amdgcn_uicmp(axis, 1u, 32);   // EQ
    float yz, r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(yz) : "v"(z), "v"(y), "s"(ax1));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(yz), "v"(x), "s"(ax0));
    return r;
}

struct FatRay {
    float o[3], c;
    float sl[3], sh[3], il[3], ih[3];   // slopes of the cone's lower / upper edge per axis, and their reciprocals
    // false: a NaN, infinite or all-zero direction (hits nothing), or the fat ray misses the (padded) scene box; else
    // [t0, t1] = its stretch inside it, from t_start on
    PT_D bool init(const DevScene& S, f3 org, f3 d, float t_start, float& t0, float& t1) {
        const float da[3] = {d.x, d.y, d.z};
        o[0] = org.x;
        o[1] = org.y;
        o[2] = org.z;
        const float dlen = mag3(d);
        if (!(dlen > 0.f) || !(dlen < INFINITY)) return false;
        const float k = PT_SLACK_K * dlen;
        float cmax = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) cmax = fmaxf(cmax, fmaxf(fabsf(S.bounds_min[a]), fabsf(S.bounds_max[a])));
        c = cmax * 9.5367431640625e-07f + PT_EXIT_ABS;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            sl[a] = da[a] - k;
            sh[a] = da[a] + k;
            il[a] = 1.0f / sl[a];
            ih[a] = 1.0f / sh[a];
        }
        t0 = t_start > 0.f ? t_start : 0.f;
        t1 = INFINITY;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (!clip_to(o[a], sl[a], il[a], S.bounds_max[a], true, t0, t1) || !clip_to(o[a], sh[a], ih[a], S.bounds_min[a], false, t0, t1))
                return false;
        return true;
    }
    // clip [t0, t1] to {x_a(t) - g(t) <= bound} (UPPER: the cone's lower edge, slope s, below the bound) or to
    // {x_a(t) + g(t) >= bound} (its upper edge above the bound); false if nothing is left
    PT_D bool clip_to(float o_a, float s, float inv, float bound, bool upper, float& t0, float& t1) const {
        const float num = upper ? (bound - o_a) + c : (bound - o_a) - c;   // edge(t) = o_a -+ c + s t  vs  bound
        if (s == 0.f) return upper ? !(num < 0.f) : !(num > 0.f);
        const float tp = num * inv;   // (a NaN - 0 x inf for a subnormal slope - is ignored by fminf / fmaxf: the interval stays)
        if ((s > 0.f) == upper) t1 = fminf(t1, tp);   // the edge moves away from the allowed side: an end
        else t0 = fmaxf(t0, tp);                      // ... towards it: a start
        return t0 <= t1;
    }
    // the two children of a node with plane (axis, split): [b0, b1] for the below child, [a0, a1] for the above child
    PT_D void children(uint32_t axis, float split, float t0, float t1, bool& vb, float& b0, float& b1, bool& va, float& a0, float& a1) const {
        const float o_a = pt_by_axis(axis, o[0], o[1], o[2]);
        const float s_l = pt_by_axis(axis, sl[0], sl[1], sl[2]), i_l = pt_by_axis(axis, il[0], il[1], il[2]);
        const float s_h = pt_by_axis(axis, sh[0], sh[1], sh[2]), i_h = pt_by_axis(axis, ih[0], ih[1], ih[2]);
        b0 = a0 = t0;
        b1 = a1 = t1;
        vb = clip_to(o_a, s_l, i_l, split, true, b0, b1);
        va = clip_to(o_a, s_h, i_h, split, false, a0, a1);
    }
};

template <bool COUNT, class LeafFn>
PT_D void kd_traverse_box(const DevScene& S, f3 o, f3 d, float t_start, float key_scale, float& limit, LocalCtr& lc, LeafFn&& leaf) {
    FatRay F;
    float t0, t1;
    if (!F.init(S, o, d, t_start, t0, t1)) return;
    uint32_t st_node[PT_KD_STACK];
    float st_t0[PT_KD_STACK], st_t1[PT_KD_STACK];
    int sp = 0;
    uint32_t node = 0;
    while (true) {
        const uint2 nd = S.kd_nodes[node];
        if (COUNT) lc.nodes++;
        const uint32_t axis = nd.y & 3u;
        bool descend = false;
        if (axis != 3u) {
            const uint32_t below = nd.y >> 2, above = below + 1;
            float b0, b1, a0, a1;
            bool vb, va;
            F.children(axis, __uint_as_float(nd.x), t0, t1, vb, b0, b1, va, a0, a1);
            if (vb && va) {
                const bool below_first = b0 <= a0;
                st_node[sp] = below_first ? above : below;
                st_t0[sp] = below_first ? a0 : b0;
                st_t1[sp] = below_first ? a1 : b1;
                ++sp;
                node = below_first ? below : above;
                t0 = below_first ? b0 : a0;
                t1 = below_first ? b1 : a1;
                descend = true;
            } else if (vb || va) {
                node = vb ? below : above;
                t0 = vb ? b0 : a0;
                t1 = vb ? b1 : a1;
                descend = true;
            }
        } else {
            const uint32_t n_refs = nd.y >> 2;
            if (n_refs && leaf(nd.x, n_refs)) return;
        }
        if (descend) continue;
        while (true) {
            if (sp == 0) return;
            --sp;
            node = st_node[sp];
            t0 = st_t0[sp];
            t1 = st_t1[sp];
            // (the segment's hits have keys >= t0 * min(1, |d|): beyond the best one it holds nothing of interest)
            if (!(t0 * key_scale > limit + PT_EXIT_ABS)) break;
        }
    }
}

// The three 16-byte words of a primitive record, as one batch of loads.  Without this the compiler sinks the loads of
// e1.yz / e2 into the triangle branch (a sphere only needs the first five words), so that the common case - a triangle -
// pays TWO dependent trips to memory per test; the traversal kernels are bound by exactly those trips.
PT_D void load_prim_record(const float4* __restrict__ rec, float4& q0, float4& q1, float4& q2) {
    q0 = rec[0];
    q1 = rec[1];
    q2 = rec[2];
    asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w),
                      "+v"(q2.x), "+v"(q2.y));
}

// Triangle::intersect on a leaf record (v0, e1, e2).  Returns true and fills
// (dist, u, v, backface) on a hit.
PT_D bool isect_triangle(f3 o, f3 d, f3 v0, f3 e1, f3 e2, float& dist, float& u, float& v, bool& backface) {
    f3 pvec = cross3(d, e2);
    float det = dot3(e1, pvec);
    if (fabsf(det) < 0.000001f) return false;
    float invdet = 1.0f / det;
    f3 tvec = o - v0;
    u = dot3(tvec, pvec) * invdet;
    if (!(u >= 0.0f && u <= 1.0f)) return false;
    f3 qvec = cross3(tvec, e1);
    v = dot3(d, qvec) * invdet;
    if (v < 0.0f || u + v > 1.0f) return false;
    dist = dot3(e2, qvec) * invdet;
    // `dist < 1e-6` rejects; a NaN distance panics the reference's sort (utils.rs:19):
    // it is rejected here.
    if (!(dist >= 0.000001f)) return false;
    backface = det < 0.0f;
    return true;
}

// Model::intersect for spheres: returns the number of hits (0..2); t[] are the
// ray parameters, key[] = |hit_point - origin| in the reference's order (entry first).
PT_D int isect_sphere(f3 o, f3 d, f3 center, float radius, float t[2], float key[2], bool exit_hit[2]) {
    f3 rc = o - center;
    float a = dot3(d, d);
    float b = 2.0f * dot3(rc, d);
    float c = dot3(rc, rc) - radius * radius;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return 0;
    float t1 = (-b - sqrtf(disc)) / (2.0f * a);
    float t2 = (-b + sqrtf(disc)) / (2.0f * a);
    if (!(t1 <= t2)) return 0;  // assert!(t1 <= t2) in the reference
    if (t2 < 0.0f) return 0;
    f3 p2 = o + d * t2;
    float k2 = mag3(p2 - o);
    if (t1 < 0.0f) {
        t[0] = t2;
        key[0] = k2;
        exit_hit[0] = true;
        return 1;
    }
    f3 p1 = o + d * t1;
    t[0] = t1;
    key[0] = mag3(p1 - o);
    exit_hit[0] = false;
    t[1] = t2;
    key[1] = k2;
    exit_hit[1] = true;
    return 2;
}

// kdtree-ray's ray / box test against the scene's bounding box (the reference filters every ray_cast through the
// crate, utils.rs:13; its slab method in f32 - inv = 1 / d, t = (bound - o) * inv, tmin = max of the per-axis minima,
// tmax = min of the maxima, f32::min / max dropping NaN, hit <=> tmax >= max(tmin, 0) - rejects a ray that clips an
// EDGE of the box within rounding although a triangle lying in one of the two faces is hit: the reference's golden
// white_furnace_direct, main.rs:149-165, pins two such camera rays; oracle: kdtree_ray_slab).  Every space of the
// crate's trees is a sub-box of this box, so what the box rejects no space accepts: the cast has no hits at all.
PT_D bool scene_slab(const DevScene& S, f3 o, f3 d) {
    const float oa[3] = {o.x, o.y, o.z}, da[3] = {d.x, d.y, d.z};
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = 1.0f / da[a];
        const float t1 = (S.slab_min[a] - oa[a]) * inv, t2 = (S.slab_max[a] - oa[a]) * inv;
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
    }
    return tmax >= fmaxf(tmin, 0.f);
}
// Does a cast from o along d survive that test?  Evaluated once per cast, at its end, when it has a result.
// A ray whose origin lies STRICTLY inside the box always passes, in f32 exactly as in real numbers: on every axis
// (min - o) < 0 < (max - o), so whatever the sign of 1 / d - infinite for a zero component included, and no 0 x inf = NaN
// because neither difference is 0 - one of the two products is <= 0 and the other >= 0: tmin <= 0 <= tmax.  That is every ray
// that starts on a surface inside the scene (origin = hit + normal x 1e-5, mod.rs:266-268) and every camera inside it: six
// comparisons.  Only the others - a camera outside the scene's box, rays that leave a surface lying IN a face of the box
// outwards (white_furnace_direct's cubes) - run the test itself.  (Rounds 2-3 marked the primitives near two faces of the box
// instead and tested the casts that ended on one: round 4's near-axis test rays showed two more ways to fail - a ray that
// runs in, or a few ulps outside, a face plane of the box is rejected wherever it hits - that no marking of primitives
// covers; tests/test_oracle_golden.py counts on the CPU the casts the box rejects although the origin is strictly inside: 0.)
#ifdef PT_NO_SCENE_SLAB   // A/B builds only (tools/build_variant.sh): what the test costs
PT_D bool hit_passes_slab(const DevScene&, f3, f3) { return true; }
#else
PT_D bool hit_passes_slab(const DevScene& S, f3 o, f3 d) {
    const bool inside = o.x > S.slab_min[0] && o.x < S.slab_max[0] && o.y > S.slab_min[1] && o.y < S.slab_max[1] &&
                        o.z > S.slab_min[2] && o.z < S.slab_max[2];
    return inside || scene_slab(S, o, d);
}
#endif

PT_D bool key_less(float ka, uint32_t oa, float kb, uint32_t ob) { return ka < kb || (ka == kb && oa < ob); }

// The entry of ray_cast()'s sorted list that follows (t_prev, ord_prev); the
// first entry for t_prev = -inf.  Returns false when the list is exhausted.
template <bool COUNT>
PT_D bool next_hit(const DevScene& S, f3 o, f3 d, float t_prev, uint32_t ord_prev, RawHit& best, LocalCtr& lc) {
    best.key = INFINITY;
    best.ord = 0xffffffffu;
    best.pid = 0xffffffffu;
    float dlen = mag3(d);
    float key_scale = dlen < 1.0f ? dlen : 1.0f;              // key >= t * min(1, |d|)
    float t_start = t_prev > 0.f ? restart_param_exact(t_prev * (dlen > 1.0f ? 1.0f / dlen : 1.0f), 1.0f / d.x, 1.0f / d.y, 1.0f / d.z) : 0.f;
    if (!(t_start > 0.f)) t_start = 0.f;
    float limit = INFINITY;
    kd_traverse<COUNT>(S, o, d, t_start, key_scale, limit, lc, [&](uint32_t first, uint32_t n) {
        const float4* lp = S.leaf_prims + (size_t)first * 3;
        for (uint32_t i = 0; i < n; ++i) {
            float4 q0 = lp[3 * i], q1 = lp[3 * i + 1], q2 = lp[3 * i + 2];
            uint32_t pid = __float_as_uint(q0.w);
            if (COUNT) lc.tris++;
            if (!(pid & PT_PRIM_SPHERE)) {
                float dist, u, v;
                bool bf;
                if (!isect_triangle(o, d, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q1.w, q2.x, q2.y), dist,
                                    u, v, bf))
                    continue;
                uint32_t ord = PT_PRIM_INDEX(pid) * 2u;
                if (key_less(t_prev, ord_prev, dist, ord) && key_less(dist, ord, best.key, best.ord)) {
                    best.key = dist;
                    best.ord = ord;
                    best.pid = pid;
                    best.u = u;
                    best.v = v;
                    best.flags = bf ? 1u : 0u;
                    limit = dist;
                }
            } else {
                float t[2], key[2];
                bool ex[2];
                int nh = isect_sphere(o, d, mk3(q0.x, q0.y, q0.z), q1.x, t, key, ex);
                for (int k = 0; k < nh; ++k) {
                    uint32_t ord = PT_PRIM_INDEX(pid) * 2u + (ex[k] ? 1u : 0u);
                    if (key[k] == key[k] && key_less(t_prev, ord_prev, key[k], ord) &&
                        key_less(key[k], ord, best.key, best.ord)) {
                        best.key = key[k];
                        best.ord = ord;
                        best.pid = pid;
                        best.u = t[k];
                        best.v = 0.f;
                        best.flags = 2u | (ex[k] ? 4u : 0u);
                        limit = key[k];
                    }
                }
            }
        }
        return false;
    });
    if (best.pid != 0xffffffffu && !hit_passes_slab(S, o, d)) best.pid = 0xffffffffu;   // no hits at all
    return best.pid != 0xffffffffu;
}

// next_hit() on the grown-box walker (kd_traverse_box): the casts k_wf_trace hands to k_wf_trace_exact.
template <bool COUNT>
PT_D bool next_hit_box(const DevScene& S, f3 o, f3 d, float t_prev, uint32_t ord_prev, RawHit& best, LocalCtr& lc) {
    best.key = INFINITY;
    best.ord = 0xffffffffu;
    best.pid = 0xffffffffu;
    const float dlen = mag3(d);
    const float key_scale = dlen < 1.0f ? dlen : 1.0f;   // key >= t * min(1, |d|)
    float limit = INFINITY;
    // (every restart of an alpha walk starts at the origin again: these are the rare rays, and "behind t_prev" needs no slack
    // argument this way - the acceptance rule below is the only filter)
    kd_traverse_box<COUNT>(S, o, d, 0.f, key_scale, limit, lc, [&](uint32_t first, uint32_t n) {
        const float4* lp = S.leaf_prims + (size_t)first * 3;
        for (uint32_t i = 0; i < n; ++i) {
            const float4 q0 = lp[3 * i], q1 = lp[3 * i + 1], q2 = lp[3 * i + 2];
            const uint32_t pid = __float_as_uint(q0.w);
            if (COUNT) lc.tris++;
            if (!(pid & PT_PRIM_SPHERE)) {
                float dist, u, v;
                bool bf;
                if (!isect_triangle(o, d, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q1.w, q2.x, q2.y), dist, u, v, bf)) continue;
                const uint32_t ord = PT_PRIM_INDEX(pid) * 2u;
                if (key_less(t_prev, ord_prev, dist, ord) && key_less(dist, ord, best.key, best.ord)) {
                    best.key = dist;
                    best.ord = ord;
                    best.pid = pid;
                    best.u = u;
                    best.v = v;
                    best.flags = bf ? 1u : 0u;
                    limit = dist;
                }
            } else {
                float t[2], key[2];
                bool ex[2];
                const int nh = isect_sphere(o, d, mk3(q0.x, q0.y, q0.z), q1.x, t, key, ex);
                for (int k = 0; k < nh; ++k) {
                    const uint32_t ord = PT_PRIM_INDEX(pid) * 2u + (ex[k] ? 1u : 0u);
                    if (key[k] == key[k] && key_less(t_prev, ord_prev, key[k], ord) && key_less(key[k], ord, best.key, best.ord)) {
                        best.key = key[k];
                        best.ord = ord;
                        best.pid = pid;
                        best.u = t[k];
                        best.v = 0.f;
                        best.flags = 2u | (ex[k] ? 4u : 0u);
                        limit = key[k];
                    }
                }
            }
        }
        return false;
    });
    if (best.pid != 0xffffffffu && !hit_passes_slab(S, o, d)) best.pid = 0xffffffffu;   // kdtree-ray's box test: no hits at all
    return best.pid != 0xffffffffu;
}

// ---------------------------------------------------------------------------
// Surface record = the reference's Hit (renderer/hit.rs:5-37) + model index.
// ---------------------------------------------------------------------------
struct Surface {
    f3 pos;
    f3 normal;     // Hit::get_geometric_normal(): interpolated, NOT normalised (triangles)
    f3 tangent;    // only valid when the material has a normal map
    f2 uv;
    uint32_t model;
    bool sphere;
    bool backface;
};

PT_D void make_surface(const DevScene& S, f3 o, f3 d, const RawHit& h, Surface& s) {
    uint32_t prim = PT_PRIM_INDEX(h.pid);
    const float4* at = S.prim_attr + (size_t)prim * 4;
    float4 a3 = at[3];
    s.model = __float_as_uint(a3.w);
    if (h.flags & 2u) {  // Model::intersect sphere arm (model.rs:40-63)
        float4 a0 = at[0];
        f3 center = mk3(a0.x, a0.y, a0.z);
        f3 hp = o + d * h.u;
        f3 n = normalize3(hp - center);
        s.pos = hp;
        s.normal = (h.flags & 4u) ? -n : n;
        s.sphere = true;
        s.backface = false;
        s.uv = {0.f, 0.f};
        s.tangent = mk3(0.f, 0.f, 0.f);
        return;
    }
    float4 a0 = at[0], a1 = at[1], a2 = at[2];
    float u = h.u, v = h.v;
    // Hit::new_triangle (hit.rs:100-137)
    f3 n0 = mk3(a0.x, a0.y, a0.z), n1 = mk3(a1.x, a1.y, a1.z), n2 = mk3(a2.x, a2.y, a2.z);
    f2 uv0 = {a0.w, a1.w}, uv1 = {a2.w, a3.x}, uv2 = {a3.y, a3.z};
    s.normal = (1.0f - u - v) * n0 + u * n1 + v * n2;
    s.uv = uv0 + u * (uv1 - uv0) + v * (uv2 - uv0);
    s.pos = o + d * h.key;  // ray.origin + ray.direction * dist (triangle.rs:77)
    s.sphere = false;
    s.backface = (h.flags & 1u) != 0;
    s.tangent = mk3(0.f, 0.f, 0.f);
    if (S.materials[s.model].tex_normal >= 0) {
        const float4* pp = S.prim_pos + (size_t)prim * 3;
        float4 q1 = pp[1], q2 = pp[2];
        f3 edge1 = mk3(q1.x, q1.y, q1.z), edge2 = mk3(q1.w, q2.x, q2.y);
        f2 duv1 = uv1 - uv0, duv2 = uv2 - uv0;
        float f = 1.0f / (duv1.x * duv2.y - duv2.x * duv1.y);
        s.tangent = normalize3(mk3(f * (duv2.y * edge1.x - duv1.y * edge2.x), f * (duv2.y * edge1.y - duv1.y * edge2.y),
                                   f * (duv2.y * edge1.z - duv1.y * edge2.z)));
    }
}

// Material::get_pixel (material.rs:115-130): nearest texel, wrap, row 0 = top.
PT_D const uint8_t* texel(const DevScene& S, int32_t tex, f2 uv) {
    pt_texture t = S.textures[tex];
    uint32_t px = wrap_texel(uv.x * (float)t.width, t.width);
    uint32_t py = wrap_texel(uv.y * (float)t.height, t.height);
    return S.texels + t.offset + ((size_t)py * t.width + px) * t.channels;
}

PT_D float luma_channel(const DevScene& S, int32_t tex, float factor, f2 uv) {
    if (tex >= 0) return (float)texel(S, tex, uv)[0] / 255.f * factor;
    return factor;
}

struct MatSample {  // renderer/material_sample.rs:6-18 (ior is carried but unused by the BRDF)
    float metalness, roughness, opacity;
    f3 albedo, emissive;
};

// Hit::get_material_sample(model): `kind_sphere`/`uv` come from the hit, the
// material from the model (mod.rs:324 mixes the two on purpose).
PT_D float material_opacity(const DevScene& S, uint32_t model, bool kind_sphere, f2 uv) {
    const pt_material& m = S.materials[model];
    if (kind_sphere) return m.opacity;
    return luma_channel(S, m.tex_opacity, m.opacity, uv);
}

// Opacity of the surface a raw hit belongs to (render_pixel's alpha walk, mod.rs:190-203; directional shadow
// rays, mod.rs:289-297).  Only a material with an opacity TEXTURE needs the hit's uv - i.e. the surface record
// with its three attribute fetches and the interpolation; everywhere else the opacity is the material's factor.
PT_D float hit_opacity(const DevScene& S, f3 o, f3 d, const RawHit& h) {
    const uint32_t model = __float_as_uint(S.prim_attr[(size_t)PT_PRIM_INDEX(h.pid) * 4 + 3].w);
    const pt_material& m = S.materials[model];
    if ((h.flags & 2u) || m.tex_opacity < 0) return m.opacity;   // spheres ignore textures (MaterialSample::simple)
    Surface sf;
    make_surface(S, o, d, h, sf);
    return luma_channel(S, m.tex_opacity, m.opacity, sf.uv);
}

PT_D void material_sample(const DevScene& S, uint32_t model, bool kind_sphere, f2 uv, MatSample& r) {
    const pt_material& m = S.materials[model];
    f3 albedo = ld3(m.albedo), emissive = ld3(m.emissive);
    if (kind_sphere) {  // MaterialSample::simple
        r.metalness = m.metalness;
        r.roughness = max_rs(m.roughness, 0.0001f);
        r.albedo = albedo;
        r.opacity = m.opacity;
        r.emissive = emissive;
        return;
    }
    r.metalness = luma_channel(S, m.tex_metalness, m.metalness, uv);
    r.roughness = max_rs(luma_channel(S, m.tex_roughness, m.roughness, uv), 0.0001f);
    if (m.tex_albedo >= 0) {
        const uint8_t* p = texel(S, m.tex_albedo, uv);
        // (c/255).powf(2.2): 256 possible values, tabulated by the host libm
        r.albedo = mul_ew(mk3(S.srgb_lut[p[0]], S.srgb_lut[p[1]], S.srgb_lut[p[2]]), albedo);
    } else {
        r.albedo = albedo;
    }
    r.opacity = luma_channel(S, m.tex_opacity, m.opacity, uv);
    if (m.tex_emissive >= 0) {
        const uint8_t* p = texel(S, m.tex_emissive, uv);
        r.emissive = mul_ew(mk3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f), emissive);
    } else {
        r.emissive = emissive;
    }
}

// Hit::get_normal (hit.rs:55-82)
PT_D f3 shading_normal(const DevScene& S, const Surface& s) {
    if (s.sphere) return s.normal;
    const pt_material& m = S.materials[s.model];
    f3 n = s.normal;
    if (m.tex_normal >= 0) {
        const uint8_t* p = texel(S, m.tex_normal, s.uv);
        f3 nm = mk3((float)p[0] / 127.5f - 1.f, (float)p[1] / 127.5f - 1.f, (float)p[2] / 127.5f - 1.f);
        f3 bitangent = cross3(s.normal, s.tangent);
        n = normalize3(s.tangent * nm.x + bitangent * nm.y + s.normal * nm.z);
    }
    return s.backface ? -n : n;
}

// ---------------------------------------------------------------------------
// Cook–Torrance (cook_torrance.rs)
// ---------------------------------------------------------------------------
struct Brdf {
    float metalness, roughness;
    f3 albedo, emissive, f0, wm;
};

PT_D void ct_init(Brdf& b, const MatSample& m) {
    b.metalness = m.metalness;
    b.roughness = m.roughness;
    b.albedo = m.albedo;
    b.emissive = m.emissive;
    b.f0 = mk3(0.04f, 0.04f, 0.04f) * (1.f - m.metalness) + m.albedo * m.metalness;
    b.wm = mk3(0.f, 0.f, 0.f);
}
PT_D f3 ct_fresnel(const Brdf& b, float cos_theta) {
    return b.f0 + mk3(1.f - b.f0.x, 1.f - b.f0.y, 1.f - b.f0.z) * powi5(1.f - cos_theta);
}
PT_D float ct_g1(float ndv, float k) { return ndv / (ndv * (1.f - k) + k); }
PT_D float ct_geometry(const Brdf& b, f3 n, f3 v, f3 l) {
    float ndv = max_rs(dot3(n, v), 0.f);
    float ndl = max_rs(dot3(n, l), 0.f);
    float k = powi2(b.roughness + 1.f) / 8.f;
    return ct_g1(ndv, k) * ct_g1(ndl, k);
}
PT_D float ct_distribution(const Brdf& b, f3 n, f3 h) {
    float a = b.roughness * b.roughness;
    float a2 = a * a;
    float ndh = max_rs(dot3(n, h), 0.f);
    float ndh2 = ndh * ndh;
    float denom = ndh2 * (a2 - 1.f) + 1.f;
    denom = PT_PI * denom * denom;
    return a2 / denom;
}
PT_D f3 ct_diffuse(const Brdf& b, f3 ks, f3 n, f3 l) {
    f3 kd = mk3(1.f - ks.x, 1.f - ks.y, 1.f - ks.z) * (1.f - b.metalness);
    f3 diffuse = mul_ew(kd, b.albedo) / PT_PI;
    return diffuse * max_rs(dot3(n, l), 0.f);
}
PT_D f3 ct_eval_direct(const Brdf& b, f3 n, f3 view, f3 light) {
    f3 h = normalize3(view + light);
    float dterm = ct_distribution(b, n, h);
    f3 f = ct_fresnel(b, max_rs(dot3(h, view), 0.f));
    float g = ct_geometry(b, n, view, light);
    f3 spec = (dterm * f * g) / max_rs(4.f * max_rs(dot3(n, view), 0.f) * max_rs(dot3(n, light), 0.f), 0.0001f);
    spec = spec * max_rs(dot3(n, light), 0.f);
    f3 diff = ct_diffuse(b, f, n, light);
    return diff + spec + b.emissive;
}
PT_D f3 ct_eval_indirect(const Brdf& b, f3 n, f3 view, f3 light) {
    f3 h = normalize3(view + light);
    f3 f = ct_fresnel(b, max_rs(dot3(h, view), 0.f));
    float g = ct_geometry(b, n, view, light);
    f3 spec = mk3(0.f, 0.f, 0.f);
    if (dot3(n, light) > 0.f) {
        float wn = fabsf(dot3(view, b.wm));
        float wd = fabsf(dot3(view, n)) * fabsf(dot3(b.wm, n));
        spec = f * g * (wn / wd);
    }
    return ct_diffuse(b, f, n, light) + spec;
}
PT_D f3 transform_to_world(f3 vec, f3 n) {
    f3 nt;
    if (fabsf(n.x) > fabsf(n.y)) nt = mk3(n.z, 0.f, -n.x) / sqrtf(n.x * n.x + n.z * n.z);
    else nt = mk3(0.f, -n.z, n.y) / sqrtf(n.y * n.y + n.z * n.z);
    f3 nb = cross3(n, nt);
    return mk3(vec.x * nb.x + vec.y * n.x + vec.z * nt.x, vec.x * nb.y + vec.y * n.y + vec.z * nt.y,
               vec.x * nb.z + vec.y * n.z + vec.z * nt.z);
}
PT_D f3 ct_sample(Brdf& b, f3 n, f3 v, float r1, float r2) {
    float a = b.roughness * b.roughness;
    float a2 = a * a;
    float theta = pt_acosf(sqrtf((1.f - r1) / (r1 * (a2 - 1.f) + 1.f)));
    float phi = 2.f * PT_PI * r2;
    float st = pt_sinf(theta);
    f3 m = normalize3(mk3(st * pt_cosf(phi), pt_cosf(theta), st * pt_sinf(phi)));
    b.wm = normalize3(transform_to_world(m, n));
    f3 dir = (2.f * max_rs(dot3(v, b.wm), 0.f)) * b.wm - v;  // reflection(): utils.rs:34-36
    return normalize3(dir);
}

// ---------------------------------------------------------------------------
// Direct light (get_light_info, mod.rs:281-333)
// ---------------------------------------------------------------------------
// BOX: on the grown-box walker (kd_traverse_box / next_hit_box) instead of kd_traverse - k_og_shadow_offgrid, which is handed
// the jobs whose rays the wavefront walker does not take (near-axis rays: kd_traverse's per-axis slack makes THEM long walks)
template <bool COUNT, bool BOX = false>
PT_D void light_radiance(const DevScene& S, const DevLight& L, const Surface& hit, f3& radiance, f3& direction,
                         LocalCtr& lc) {
    const bool point = L.kind == PT_LIGHT_POINT;
    f3 color = ld3(L.color);
    float dist = 0.f;
    if (point) {
        direction = hit.pos - ld3(L.vec);
        dist = mag3(direction);
        direction = normalize3(direction);
        float dissipation = 4.f * PT_PI * dist * dist;
        color = color / dissipation;
    } else {
        direction = ld3(L.vec);
    }
    f3 so = hit.pos + hit.normal * 0.00001f;  // NORMAL_BIAS (mod.rs:58)
    f3 sd = -1.f * direction;
    if (COUNT) lc.shadow_rays++;

    if (!S.has_translucent) {
        // Every opacity is exactly 1: the first list entry that passes the range
        // test zeroes the colour, so "any hit in range" decides.
        bool blocked = false;
        float dlen = mag3(sd);
        float key_scale = dlen < 1.0f ? dlen : 1.0f;
        // hits farther than the light cannot pass the range test (|so + sd*t - pos| > dist)
        float limit = point ? (dist + 1e-4f) * 1.0001f : INFINITY;
        auto any_hit = [&](uint32_t first, uint32_t n) {
            const float4* lp = S.leaf_prims + (size_t)first * 3;
            for (uint32_t i = 0; i < n; ++i) {
                float4 q0 = lp[3 * i], q1 = lp[3 * i + 1], q2 = lp[3 * i + 2];
                uint32_t pid = __float_as_uint(q0.w);
                if (COUNT) lc.tris++;
                if (!(pid & PT_PRIM_SPHERE)) {
                    float t, u, v;
                    bool bf;
                    if (!isect_triangle(so, sd, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q1.w, q2.x, q2.y), t,
                                        u, v, bf))
                        continue;
                    if (point && mag3((so + sd * t) - hit.pos) > dist) continue;
                    blocked = true;
                    return true;
                } else {
                    float t[2], key[2];
                    bool ex[2];
                    int nh = isect_sphere(so, sd, mk3(q0.x, q0.y, q0.z), q1.x, t, key, ex);
                    for (int k = 0; k < nh; ++k) {
                        if (!(key[k] == key[k])) continue;
                        if (point && mag3((so + sd * t[k]) - hit.pos) > dist) continue;
                        blocked = true;
                        return true;
                    }
                }
            }
            return false;
        };
        if (BOX) kd_traverse_box<COUNT>(S, so, sd, 0.f, key_scale, limit, lc, any_hit);
        else kd_traverse<COUNT>(S, so, sd, 0.f, key_scale, limit, lc, any_hit);
        if (blocked && !hit_passes_slab(S, so, sd)) blocked = false;   // (a ray the box rejects has no hits at all)
        radiance = blocked ? color * 0.0f : color;
        return;
    }

    // General case: walk the sorted hit list, attenuating by (1 - opacity).
    float t_prev = -INFINITY;
    uint32_t ord_prev = 0;
    RawHit h;
    bool first = true;
    while (BOX ? next_hit_box<COUNT>(S, so, sd, t_prev, ord_prev, h, lc) : next_hit<COUNT>(S, so, sd, t_prev, ord_prev, h, lc)) {
        if (COUNT && !first) lc.restarts++;
        first = false;
        float opacity;
        if (point) {
            f3 sp = so + sd * ((h.flags & 2u) ? h.u : h.key);
            if (mag3(sp - hit.pos) > dist) break;
            // the SHADED hit's kind / uv with the occluder's material (mod.rs:324)
            uint32_t smodel = __float_as_uint(S.prim_attr[(size_t)PT_PRIM_INDEX(h.pid) * 4 + 3].w);
            opacity = material_opacity(S, smodel, hit.sphere, hit.uv);
        } else {
            Surface sh;
            make_surface(S, so, sd, h, sh);
            opacity = material_opacity(S, sh.model, sh.sphere, sh.uv);
        }
        color = color * (1.f - opacity);
        if (sum3(color) == 0.f) break;
        t_prev = h.key;
        ord_prev = h.ord;
    }
    radiance = color;
}

// ---------------------------------------------------------------------------
// Path (render_pixel + compute_radiance), one bounce-loop iteration at a time.
//
// The reference's loop body (mod.rs:180-226) is exposed as path_step() so the
// persistent kernel can interleave iterations of DIFFERENT samples in the lanes
// of one wavefront: a lane whose path ends fetches its next work item while its
// neighbours keep bouncing.  `draw()` returns the next rng.gen::<f32>().
// ---------------------------------------------------------------------------
struct PathState {
    f3 o, d;         // current ray
    f3 color, thr;   // RadianceInfo (mod.rs:41-48)
    uint32_t bounce;
};

PT_D void path_begin(PathState& ps, f3 o, f3 d) {
    ps.o = o;
    ps.d = d;
    ps.color = mk3(0.f, 0.f, 0.f);
    ps.thr = mk3(1.f, 1.f, 1.f);
    ps.bounce = 0;
}

// Runs iteration `ps.bounce` of the bounce loop.  Returns true when the path is
// finished (ps.color is then the value render_pixel returns).
template <bool COUNT, class DrawFn>
PT_D bool path_step(const DevScene& S, uint32_t bounces, PathState& ps, LocalCtr& lc, DrawFn&& draw) {
    const f3 o = ps.o, d = ps.d;
    if (COUNT) lc.segments++;
    // alpha walk over the sorted hit list (mod.rs:188-205)
    Surface surf;
    MatSample ms;
    f3 normal = mk3(0.f, 0.f, 0.f);
    bool have = false;
    float t_prev = -INFINITY;
    uint32_t ord_prev = 0;
    RawHit h;
    while (next_hit<COUNT>(S, o, d, t_prev, ord_prev, h, lc)) {
        if (COUNT && have) lc.restarts++;
        make_surface(S, o, d, h, surf);
        material_sample(S, surf.model, surf.sphere, surf.uv, ms);
        normal = shading_normal(S, surf);
        have = true;
        if (COUNT) lc.shaded++;
        float opacity = ms.opacity;
        if (opacity >= 1.f || (opacity > 0.001f && draw() < opacity)) break;
        t_prev = h.key;
        ord_prev = h.ord;
    }
    if (!have) {
        ps.color = ps.color + mul_ew(ps.thr, ld3(S.background));
        return true;
    }
    f3 view = -1.f * d;
    Brdf brdf;
    ct_init(brdf, ms);
    f3 color = ps.color + mul_ew(ps.thr, ms.emissive);
    for (uint32_t li = 0; li < S.n_lights; ++li) {
        f3 lrad, ldir;
        light_radiance<COUNT>(S, S.lights[li], surf, lrad, ldir, lc);
        if (lrad.x == 0.f && lrad.y == 0.f && lrad.z == 0.f) continue;
        f3 rl = -1.f * ldir;
        color = color + mul_ew(mul_ew(ps.thr, ct_eval_direct(brdf, normal, view, rl)), lrad);
    }
    ps.color = color;
    f3 thr = ps.thr;
    if (ps.bounce < bounces) {
        ps.o = surf.pos + surf.normal * 0.00001f;
        float r1 = draw();
        float r2 = draw();
        ps.d = ct_sample(brdf, normal, view, r1, r2);
        f3 w = ct_eval_indirect(brdf, normal, view, ps.d) / 1.0f;  // / brdf.pdf()
        thr = mul_ew(thr, w);
    }
    if (dot3(thr, thr) < 0.00001f) return true;
    if (ps.bounce > 3) {  // russian_roulette (utils.rs:23-31)
        float p = max_rs(max_rs(thr.x, thr.y), thr.z);
        thr = thr * (1.f / p);
        if (draw() > p) return true;
    }
    ps.thr = thr;
    ps.bounce++;
    return ps.bounce > bounces;
}

// render_pixel for one sample (megakernel form).
template <bool COUNT>
PT_D f3 render_path(const DevScene& S, uint32_t bounces, f3 o, f3 d, PtRng& rng, uint32_t* slab, uint32_t tid,
                    LocalCtr& lc) {
    PathState ps;
    path_begin(ps, o, d);
    while (!path_step<COUNT>(S, bounces, ps, lc, [&]() { return pt_rng_f32(rng, slab, tid); })) {
    }
    return ps.color;
}

// Camera ray (mod.rs:107-124), in two halves: the jittered screen position of a pixel sample and the
// ray through a screen position (the wavefront integrator computes the first half once per item).
PT_D void primary_screen(const DevScene& S, uint32_t x, uint32_t y, uint32_t width, uint32_t height, float r1, float r2,
                         float& sx, float& sy) {
    float wf = (float)width, hf = (float)height;
    float ratio = wf / hf;
    sx = (float)x + r1;
    sx = sx / wf * 2.f - 1.f;
    sx *= S.tan_half_fov * ratio;
    sy = (float)y + r2;
    sy = 1.f - sy / hf * 2.f;
    sy *= S.tan_half_fov;
}
PT_D void primary_from_screen(const DevScene& S, float sx, float sy, f3& o, f3& d) {
    f3 dir = normalize3(mk3(sx, sy, -1.f));
    f3 c0 = ld3(S.cam_c0), c1 = ld3(S.cam_c1), c2 = ld3(S.cam_c2), c3 = ld3(S.cam_c3);
    d = c0 * dir.x + c1 * dir.y + c2 * dir.z + c3 * 0.0f;
    o = c3;
}
PT_D void primary_ray(const DevScene& S, uint32_t x, uint32_t y, uint32_t width, uint32_t height, float r1, float r2,
                      f3& o, f3& d) {
    float sx, sy;
    primary_screen(S, x, y, width, height, r1, r2, sx, sy);
    primary_from_screen(S, sx, sy, o, d);
}

// tonemap + gamma + u8 (tonemap.rs:15-54, mod.rs:335-353)
PT_D f3 tonemap(int type, f3 c) {
    if (type == PT_TONEMAP_REINHARD) return div_ew(c, c + mk3(1.f, 1.f, 1.f));
    if (type == PT_TONEMAP_ACES) {
        f3 num = mul_ew(c, 2.51f * c + mk3(0.03f, 0.03f, 0.03f));
        f3 den = mul_ew(c, 2.43f * c + mk3(0.59f, 0.59f, 0.59f)) + mk3(0.14f, 0.14f, 0.14f);
        f3 r = div_ew(num, den);
        auto cl = [](float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); };
        return mk3(cl(r.x), cl(r.y), cl(r.z));
    }
    f3 col = c - mk3(0.004f, 0.004f, 0.004f);
    col = mk3(max_rs(col.x, 0.f), max_rs(col.y, 0.f), max_rs(col.z, 0.f));
    f3 num = mul_ew(col, 6.2f * col + mk3(0.5f, 0.5f, 0.5f));
    f3 den = mul_ew(col, 6.2f * col + mk3(1.7f, 1.7f, 1.7f)) + mk3(0.06f, 0.06f, 0.06f);
    return div_ew(num, den);
}

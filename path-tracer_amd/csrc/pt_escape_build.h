// Escape masks: the builder (see pt_escape.h for what a mask asserts and how it is constructed).
#pragma once
#include "pt_escape.h"
#include "pt_wavefront.h"

// ---------------------------------------------------------------------------
// The builder.
// ---------------------------------------------------------------------------
struct EscBuildParams {
    float delta_in;      // how far from P (in its plane) an accepted hit's point may lie: PT_SLACK_K x the scene's reach + rounding
    float slop_far;      // absolute fattening of far geometry: the same
    float alpha_stop;    // a node is opened while its cone is wider than this (radians)
    float r_near_scale;  // primitives at or below P's plane within this many origin-set radii count for beta
};

#define ESC_STACK 96
__global__ __launch_bounds__(256) void k_escape_build(DevScene S, EscBuildParams E, float4* __restrict__ out, uint32_t n_prims,
                                                      uint32_t* __restrict__ stats) {
    __shared__ uint32_t st_node[4][ESC_STACK];
    __shared__ float st_box[4][ESC_STACK][6];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t cu = lane & 7u, cv = lane >> 3;
    // this lane's six cells: unit centre direction and the angular radius of the cell (+ the rounding of esc_cell)
    f3 cdir[6];
    float cos_rho, sin_rho;
    {
        const float u0 = (float)cu * 0.25f - 1.0f, v0 = (float)cv * 0.25f - 1.0f;
        const f3 c = normalize3(mk3(1.0f, u0 + 0.125f, v0 + 0.125f));
        float worst = 1.0f;
        for (int k = 0; k < 4; ++k) {
            const f3 q = normalize3(mk3(1.0f, u0 + ((k & 1) ? 0.25f : 0.f), v0 + ((k & 2) ? 0.25f : 0.f)));
            worst = fminf(worst, dot3(c, q));
        }
        const float rho = acosf(fminf(1.0f, worst)) + 2e-5f;
        cos_rho = cosf(rho);
        sin_rho = sinf(rho);
        // face f = 2 * axis + negative: (a, b, c) = (x, y, z), (y, z, x), (z, x, y)   [esc_cell: wb = next axis, wc = the one after]
        for (int f = 0; f < 6; ++f) {
            const float s = (f & 1) ? -1.0f : 1.0f;
            const int a = f >> 1;
            const float va = s * c.x, vb = c.y, vc = c.z;
            cdir[f] = a == 0 ? mk3(va, vb, vc) : (a == 1 ? mk3(vc, va, vb) : mk3(vb, vc, va));
        }
    }
    // cells of this lane touched by the cone (axis, cos / sin of its half-angle): bit f
    auto touched = [&](f3 axis, float cos_a, float sin_a) -> uint32_t {
        // angle(centre, axis) <= alpha + rho  <=>  dot >= cos(alpha + rho)   (alpha + rho >= pi: everything)
        const float lim = cos_a * cos_rho - sin_a * sin_rho;
        const bool all = (sin_a * cos_rho + cos_a * sin_rho) < 0.f;   // sin(alpha + rho) < 0: alpha + rho > pi
        uint32_t m = 0;
#pragma unroll
        for (int f = 0; f < 6; ++f) m |= (all || dot3(cdir[f], axis) >= lim) ? (1u << f) : 0u;
        return m;
    };
    for (uint32_t prim = blockIdx.x * 4u + wave; prim < n_prims; prim += gridDim.x * 4u) {
        const float4* pp = S.prim_pos + (size_t)prim * 3;
        const float4 q0 = pp[0], q1 = pp[1], q2 = pp[2];
        const uint32_t pid = __float_as_uint(q0.w);
        float4* rec = out + (size_t)prim * 5;
        bool safe = !(pid & PT_PRIM_SPHERE);
        const f3 v0 = mk3(q0.x, q0.y, q0.z), e1 = mk3(q1.x, q1.y, q1.z), e2 = mk3(q1.w, q2.x, q2.y);
        f3 nrm = cross3(e1, e2);
        const float nlen = mag3(nrm);
        if (!(nlen > 1e-30f) || !(nlen < 1e30f)) safe = false;
        if (safe) {
            nrm = nrm * (1.0f / nlen);
            // the side the vertex normals point to (the normal bias of mod.rs:266-268 moves the origin there)
            const float4* at = S.prim_attr + (size_t)prim * 4;
            const float4 a0 = at[0], a1 = at[1], a2 = at[2];
            const f3 nsum = mk3(a0.x + a1.x + a2.x, a0.y + a1.y + a2.y, a0.z + a1.z + a2.z);
            if (dot3(nrm, nsum) < 0.f) nrm = -1.f * nrm;
        }
        uint32_t mask = 0u;   // bit f: this lane's cell of face f is blocked
        // primitives at or below the plane NEAR P, the largest |o - v0| an origin on P can have with one of them: coplanar ones
        // (reach_a) and the others (reach_b) - see the end of the walk
        float reach_a = 0.f, reach_b = 0.f;
        if (safe) {
            const f3 p1 = v0 + e1, p2 = v0 + e2;
            const f3 cen = (v0 + p1 + p2) * (1.0f / 3.0f);
            const float r_p = sqrtf(fmaxf(fmaxf(dot3(v0 - cen, v0 - cen), dot3(p1 - cen, p1 - cen)), dot3(p2 - cen, p2 - cen)));
            const f3 c_o = cen + nrm * (0.5f * PT_ESC_H_HI);
            const float r_o = r_p + E.delta_in + PT_ESC_H_HI;
            const float r_near = E.r_near_scale * r_o;
            reach_a = 2.0f * r_o;                     // (P itself)
            const float cut = 0.25f * PT_ESC_H_LO;   // "at or below the plane"
            const float cut_c = PT_ESC_H_LO * 0.0625f;   // "in the plane"
            int sp = 0;
            uint32_t node = 0u;
            float lo[3] = {S.bounds_min[0], S.bounds_min[1], S.bounds_min[2]}, hi[3] = {S.bounds_max[0], S.bounds_max[1], S.bounds_max[2]};
            while (safe) {
                const uint2 nd = S.kd_nodes[node];
                const uint32_t axis = nd.y & 3u;
                const f3 cb = mk3(0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]));
                const f3 hb = mk3(0.5f * (hi[0] - lo[0]), 0.5f * (hi[1] - lo[1]), 0.5f * (hi[2] - lo[2]));
                const float rb = mag3(hb) + E.slop_far;
                // highest point of the box above P's plane
                const float top = dot3(nrm, cb - v0) + (fabsf(nrm.x) * hb.x + fabsf(nrm.y) * hb.y + fabsf(nrm.z) * hb.z);
                const f3 to = cb - c_o;
                const float dist = mag3(to);
                bool open = false;   // descend / look at the leaf's primitives
                if (top <= cut) {
                    // everything in here lies at or below the plane: only what is NEAR P matters (for beta)
                    open = dist - rb <= r_near;
                } else if (dist > 2.0f * (r_o + rb)) {
                    const float sin_a = (r_o + rb) / dist, cos_a = sqrtf(fmaxf(0.f, 1.0f - sin_a * sin_a));
                    const uint32_t t = touched(to * (1.0f / dist), cos_a, sin_a);
                    const bool narrow = sin_a <= E.alpha_stop || axis == 3u;
                    if (narrow) mask |= t;
                    else open = wf_any((t & ~mask) != 0u);   // (every cell it touches blocked already: nothing to learn)
                } else {
                    open = true;
                }
                bool descend = false;
                if (open && axis != 3u) {
                    const float split = __uint_as_float(nd.x);
                    const uint32_t below = nd.y >> 2;
                    // push the above child, go on with the below child
                    if (sp >= ESC_STACK) {
                        safe = false;   // (cannot happen: the tree's depth is below 64)
                        break;
                    }
                    if (lane == 0u) {
                        st_node[wave][sp] = below + 1u;
                        for (int k = 0; k < 3; ++k) {
                            st_box[wave][sp][k] = (uint32_t)k == axis ? split : lo[k];
                            st_box[wave][sp][3 + k] = hi[k];
                        }
                    }
                    ++sp;
                    if (axis == 0u) hi[0] = split;
                    else if (axis == 1u) hi[1] = split;
                    else hi[2] = split;
                    node = below;
                    descend = true;
                } else if (open) {
                    // a leaf close to P (or at / below its plane, near): its primitives one by one
                    const uint32_t n_refs = nd.y >> 2;
                    const float4* lp = S.leaf_prims + (size_t)nd.x * 3;
                    for (uint32_t i = 0; i < n_refs && safe; ++i) {
                        const float4 t0 = lp[3 * i], t1 = lp[3 * i + 1], t2 = lp[3 * i + 2];
                        const uint32_t qid = __float_as_uint(t0.w);
                        if (PT_PRIM_INDEX(qid) == prim) continue;   // P itself: in its own plane
                        f3 cq;
                        float rq, top_q, bot_q;
                        if (qid & PT_PRIM_SPHERE) {
                            cq = mk3(t0.x, t0.y, t0.z);
                            rq = fabsf(t1.x);
                            top_q = dot3(nrm, cq - v0) + rq;
                            bot_q = top_q - 2.0f * rq;
                        } else {
                            const f3 a = mk3(t0.x, t0.y, t0.z), b = a + mk3(t1.x, t1.y, t1.z), c = a + mk3(t1.w, t2.x, t2.y);
                            cq = (a + b + c) * (1.0f / 3.0f);
                            rq = sqrtf(fmaxf(fmaxf(dot3(a - cq, a - cq), dot3(b - cq, b - cq)), dot3(c - cq, c - cq)));
                            const float sa = dot3(nrm, a - v0), sb = dot3(nrm, b - v0), sc = dot3(nrm, c - v0);
                            top_q = fmaxf(fmaxf(sa, sb), sc);
                            bot_q = fminf(fminf(sa, sb), sc);
                        }
                        const f3 tq = cq - c_o;
                        const float dq = mag3(tq);
                        if (top_q <= cut) {
                            if (dq - rq <= r_near) {
                                const bool coplanar = !(qid & PT_PRIM_SPHERE) && top_q <= cut_c && bot_q >= -cut_c;
                                if (coplanar) reach_a = fmaxf(reach_a, dq + rq + r_o);
                                else reach_b = fmaxf(reach_b, dq + rq + r_o);
                            }
                            continue;
                        }
                        const float rqf = rq + E.slop_far;
                        if (dq > 1.05f * (r_o + rqf)) {
                            const float sin_a = fminf(1.0f, (r_o + rqf) / dq), cos_a = sqrtf(fmaxf(0.f, 1.0f - sin_a * sin_a));
                            mask |= touched(tq * (1.0f / dq), cos_a, sin_a);
                        } else {
                            safe = false;   // rises above the plane right next to P: not examined
                        }
                    }
                }
                if (descend) continue;
                if (sp == 0) break;
                --sp;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                node = st_node[wave][sp];
                for (int k = 0; k < 3; ++k) {
                    lo[k] = st_box[wave][sp][k];
                    hi[k] = st_box[wave][sp][3 + k];
                }
            }
            // Geometry at or below the plane, seen from an origin H_LO or more above it, by a ray that rises:
            //   * a primitive IN P's plane (P itself, the rest of a flat floor): the true distance is negative, and the f32 test
            //     gets its sign from e2 . ((o - v0) x e1) = -(height) |e1 x e2|, computed with an absolute error of ~16 eps
            //     |o - v0| |e1| |e2|: reliable - the hit rejected, dist < 1e-6 - while height > 2e-6 |o - v0| (eps = 2^-24, the
            //     triangle's angle allowed for).  Far ones (|o - v0| large) fail that, but the point the test then makes up lies
            //     at the ray's origin, whose foot is r_near away from them: rejected by u, v;
            //   * another primitive below the plane near P (a convex neighbour that falls away): the ray crosses ITS plane
            //     outside it, by the ray's height above P's plane or more; the test places that point with an error of ~32 eps
            //     |o - v0| / sin(angle to its plane), and that angle is at least the ray's rise: safe for sin(beta) >= 32 eps
            //     reach / H_LO.
            if (safe && PT_ESC_H_LO < 2e-6f * reach_a) safe = false;
            if (safe) {
                const float sin_b = fmaxf(2e-3f, 32.0f * 5.9604645e-8f * reach_b / PT_ESC_H_LO);
#pragma unroll
                for (int f = 0; f < 6; ++f) {
                    const float x = fminf(1.0f, fmaxf(-1.0f, dot3(cdir[f], nrm)));
                    // the smallest d . N over the cell: cos(angle + rho)
                    const float s = sqrtf(fmaxf(0.f, 1.0f - x * x));
                    const float low = (x * cos_rho - s * sin_rho);
                    const bool wraps = (s * cos_rho + x * sin_rho) < 0.f;   // angle + rho > pi
                    if (wraps || low < sin_b || !(sin_b < 1.0f)) mask |= 1u << f;
                }
            }
        }
        if (!safe) mask = 0x3fu;
        // the six 64-bit face words: bit (row * 8 + column) = this lane
        uint32_t words[12];
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const unsigned long long b = __ballot((mask >> f) & 1u);
            words[2 * f] = (uint32_t)b;
            words[2 * f + 1] = (uint32_t)(b >> 32);
        }
        if (lane == 0u) {
            rec[0] = safe ? make_float4(nrm.x, nrm.y, nrm.z, v0.x) : make_float4(0.f, 0.f, 0.f, 0.f);
            rec[1] = make_float4(v0.y, v0.z, 0.f, 0.f);
            rec[2] = make_float4(__uint_as_float(words[0]), __uint_as_float(words[1]), __uint_as_float(words[2]), __uint_as_float(words[3]));
            rec[3] = make_float4(__uint_as_float(words[4]), __uint_as_float(words[5]), __uint_as_float(words[6]), __uint_as_float(words[7]));
            rec[4] = make_float4(__uint_as_float(words[8]), __uint_as_float(words[9]), __uint_as_float(words[10]), __uint_as_float(words[11]));
            if (stats) {
                atomicAdd(&stats[0], safe ? 1u : 0u);
                uint32_t clear = 0;
                for (int k = 0; k < 12; ++k) clear += 32u - (uint32_t)__popc(words[k]);
                atomicAdd(&stats[1], safe ? clear : 0u);
            }
        }
    }
}

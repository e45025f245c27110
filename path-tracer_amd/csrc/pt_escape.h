// Escape masks: proofs of MISSES for the rays of bounces >= 1.
//
// The reference casts every bounce ray (renderer/mod.rs:180-186: ray_cast, empty -> colour + throughput x background); three
// of four such casts of an open scene return nothing.  A ray that leaves primitive P is a miss whatever the sample's random
// numbers are if nothing can be hit from anywhere on P in its direction - a property of the SCENE, computed once per scene:
// per primitive a cube map of directions, 8 x 8 cells per face (384 bits), a SET bit meaning "something may be hit" (or: not
// examined).  k_wf_shade consults it when it has sampled the next direction: a clear bit ends the path with the background
// term - no queue record, no cast, no hit record.  Like the origin grids (pt_grid.h) the masks are a filter in front of the
// reference's arithmetic, conservative by construction, and the parity tests compare the masked pipeline with two that know
// no masks (KD-tree pipeline, megakernel) bit for bit.
//
// What a clear bit of P's mask asserts: for every origin x with H_LO <= sigma(x) <= H_HI (sigma = height above P's plane, on
// the side of its normal) whose foot lies within delta of P, and every direction d of the cell, ray_cast(x, d) is empty.
// The shade kernel checks the height of the ACTUAL origin (hit point + normal x 1e-5, mod.rs:266-268: an f32 value, its
// height above the plane through v0 is computed from x - v0, a small difference, to ~1e-8); the foot condition is the slop
// model's (pt_integrator.h: a hit accepted by f32 Moeller-Trumbore lies within PT_SLACK_K x t of its triangle).
//
// Construction (k_escape_build, one wavefront per primitive, lane = cell (column, row), six faces per lane):
//   * everything at or below P's plane (sigma <= H_LO / 4: P itself, coplanar and convex neighbours, the ground under an
//     object) can only be hit by a ray that does not rise: such geometry blocks the directions with d . N < sin(beta), beta
//     large enough that the f32 intersection test is well conditioned for every such primitive near P (its error in position,
//     ~32 eps |o - v0| / sin(theta), stays below H_LO - the rising ray is then farther from the primitive's plane than the
//     test's slop at every t);
//   * geometry that rises above the plane: the KD-tree is walked with P's origin set as a sphere; a node (or a leaf's
//     primitive) whose bounding sphere is clear of it blocks the cone of directions between the two spheres, fattened by the
//     slop; a node is opened until its cone is narrower than a fraction of a cell (or every cell it touches is blocked
//     already);
//   * a primitive above the plane whose sphere is NOT clear of the origin set (a concave neighbour, something resting on P):
//     P is left unexamined - all bits set (record normal = 0).  Spheres, degenerate triangles: likewise.
// So the masks answer for flat and convex neighbourhoods - the ground, table tops, the outside of smooth convex bodies -
// which is where the escaping rays of an open scene start (config 3 in the reference's framing: 85 % of them on the ground).
#pragma once
#include "pt_device.h"
#include "pt_integrator.h"

#define PT_ESC_RES 8u
#define PT_ESC_WORDS 20u            // 80-byte record: (N.xyz, v0.x) (v0.yz, -, -) then six faces x 64 bits
#define PT_ESC_H_LO 5e-6f
#define PT_ESC_H_HI 1e-3f

// Direction -> (face, column, row) of the 8 x 8 cube map (the arithmetic of og_cell_coords, pt_grid.h, at res 8).
PT_D void esc_cell(f3 w, uint32_t& face, uint32_t& cu, uint32_t& cv) {
    const float ax = fabsf(w.x), ay = fabsf(w.y), az = fabsf(w.z);
    const bool fx = ax >= ay && ax >= az, fy = !fx && ay >= az;
    const float wa = fx ? w.x : (fy ? w.y : w.z);
    const float wb = fx ? w.y : (fy ? w.z : w.x);
    const float wc = fx ? w.z : (fy ? w.x : w.y);
    const float inv = 1.0f / fabsf(wa);
    const float fu = (wb * inv + 1.0f) * (0.5f * PT_ESC_RES), fv = (wc * inv + 1.0f) * (0.5f * PT_ESC_RES);
    const float top = (float)(PT_ESC_RES - 1u);
    const float u = fu >= 0.f ? fminf(floorf(fu), top) : 0.f, v = fv >= 0.f ? fminf(floorf(fv), top) : 0.f;
    face = (fx ? 0u : (fy ? 2u : 4u)) + (wa < 0.f ? 1u : 0u);
    cu = (uint32_t)u;
    cv = (uint32_t)v;
}

// Is the ray (o, d) that leaves primitive `prim` proven to hit nothing?
PT_D bool escape_proves_miss(const DevScene& S, uint32_t prim, f3 o, f3 d) {
    const float4* r = S.escape + (size_t)prim * 5;
    const float4 r0 = r[0], r1 = r[1];
    const f3 n = mk3(r0.x, r0.y, r0.z);
    if (n.x == 0.f && n.y == 0.f && n.z == 0.f) return false;   // not examined
    const f3 rel = o - mk3(r0.w, r1.x, r1.y);
    const float h = dot3(n, rel);
    if (!(h >= PT_ESC_H_LO && h <= PT_ESC_H_HI)) return false;   // (NaN: no proof)
    if (!(d.x == d.x && d.y == d.y && d.z == d.z)) return false;
    uint32_t face, cu, cv;
    esc_cell(d, face, cu, cv);
    const uint2 bits = ((const uint2*)(r + 2))[face];
    const uint32_t bit = cv * PT_ESC_RES + cu;
    const uint32_t word = bit < 32u ? bits.x : bits.y;
    return ((word >> (bit & 31u)) & 1u) == 0u;
}


// Origin grids: the geometry shared by the host builder (host/origin_grid.cpp) and the device builder
// (csrc/pt_grid_build.h) - footprints of the primitives (what a primitive contributes to a grid: a polygon, a margin in
// cells, a distance bound) and their conservative rasterisation into the cells of a cube map / an orthographic grid.
// All of it is f64 arithmetic with IEEE operations only (+ - * / sqrt, floor, comparisons), compiled without
// contraction on both sides, so host and device produce the SAME lists (tests/test_origin_grid.py compares them byte
// for byte).  See host/origin_grid.cpp for the derivation of the margins.
#pragma once
#include <cmath>
#include <cstdint>

#include "ptgpu.h"
#include "pthost.h"

#if defined(__HIPCC__)
#define OG_HD __host__ __device__
#else
#define OG_HD
#endif

namespace pth {
namespace og {

OG_HD inline double og_min(double a, double b) { return a < b ? a : b; }
OG_HD inline double og_max(double a, double b) { return a > b ? a : b; }
OG_HD inline bool og_finite(double x) { return x - x == 0.0; }
// cos / sin of k * 45 degrees and cos(22.5 degrees) as constants: one libm less that could differ between host and device
#define OG_R2 0.70710678118654757
constexpr double kCosPi8 = 0.92387953251128674;

constexpr double kEps32 = 5.9604644775390625e-8;  // 2^-24
constexpr int kMaxPoly = 24;                      // octagon clipped by four planes: <= 12 vertices

struct Vec {
    double x, y, z;
};
OG_HD inline Vec operator-(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
OG_HD inline Vec operator+(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
OG_HD inline Vec operator*(Vec a, double s) { return {a.x * s, a.y * s, a.z * s}; }
OG_HD inline double dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
OG_HD inline Vec cross(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
OG_HD inline double len(Vec a) { return sqrt(dot(a, a)); }
OG_HD inline double comp(Vec a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
OG_HD inline double oct_cos(int k) {
    const double t[8] = {1.0, OG_R2, 0.0, -OG_R2, -1.0, -OG_R2, 0.0, OG_R2};
    return t[k];
}
OG_HD inline double oct_sin(int k) {
    const double t[8] = {0.0, OG_R2, 1.0, OG_R2, 0.0, -OG_R2, -1.0, -OG_R2};
    return t[k];
}

// Distance from the origin to triangle (a, b, c) (Ericson, Real-Time Collision Detection 5.1.5).
OG_HD inline double origin_triangle_distance(Vec a, Vec b, Vec c) {
    const Vec p{0, 0, 0};
    Vec ab = b - a, ac = c - a, ap = p - a;
    double d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0 && d2 <= 0) return len(a);
    Vec bp = p - b;
    double d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0 && d4 <= d3) return len(b);
    double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) return len(a + ab * (d1 / (d1 - d3)));
    Vec cp = p - c;
    double d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0 && d5 <= d6) return len(c);
    double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) return len(a + ac * (d2 / (d2 - d6)));
    double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) return len(b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6))));
    double denom = 1.0 / (va + vb + vc);
    return len(a + ab * (vb * denom) + ac * (vc * denom));
}

struct Footprint {       // what one primitive contributes to the grid
    bool skip = false;   // never intersected by anything (kept out)
    bool global = false; // tested by every ray
    int n = 0;           // polygon (relative to O) whose projection covers the primitive
    Vec poly[8];
    double margin = 0;   // cells
    float mindist = 0;   // lower bound of |hit - O|
};

struct GridParams {
    Vec origin;
    uint32_t res;
    double base_margin;   // cells
    double max_margin;    // cells
    double near_radius;   // world units
    double ray_offset;    // world units by which a ray may miss O (0: camera)
    double max_dir_len;   // longest ray direction (Triangle::intersect does not normalise it)
    double abs_slack;     // world units subtracted from every stored distance
    // orthographic grids (rays with ONE direction: the shadow rays of a directional light)
    bool ortho = false;
    Vec axis_u{1, 0, 0}, axis_v{0, 1, 0}, axis_w{0, 0, 1};   // axis_w = the rays' direction (unit); u, v span the grid plane
    double u0 = 0, v0 = 0, cells_per_unit = 1;
    double ray_len = 1;    // |direction| of the rays (Triangle::intersect takes it as it is)
    double reach_max = 0;  // largest |ray origin - vertex| (the scene's diagonal: the rays start anywhere in the scene)
};

OG_HD inline float round_down(double v) {
    if (!(v > 0)) return 0.f;
    float f = (float)v;
    if ((double)f > v) f = nextafterf(f, 0.f);
    return f;
}

OG_HD inline Footprint triangle_footprint(const GridParams& P, const float* v) {
    Footprint fp;
    Vec a{v[0], v[1], v[2]}, b{v[8], v[9], v[10]}, c{v[16], v[17], v[18]};
    a = a - P.origin;
    b = b - P.origin;
    c = c - P.origin;
    Vec e1 = b - a, e2 = c - a, n = cross(e1, e2);
    const double l1 = len(e1), l2 = len(e2), ln = len(n), lmax = og_max(l1, l2);
    const double dmax = og_max(len(a), og_max(len(b), len(c)));
    if (!og_finite(dmax) || !og_finite(ln)) {
        fp.global = true;   // non-finite vertex: whatever the f32 test makes of it, every ray sees it
        return fp;
    }
    // ---- error model of the f32 Möller–Trumbore (triangle.rs:37-82) for a ray of length <= dl from (about) O:
    //   det = e1 . (d x e2)            computed with absolute error <= e_det
    //   u, v = (t . p, d . q) / det    numerators computed with absolute error <= e_num  (|t| <= dmax)
    // A hit is only accepted with |det| >= 1e-6, i.e. with a true determinant of at least thr.
    const double dl = P.max_dir_len;
    // (first-order rounding analysis: cross product 2 sqrt(3) eps |a||b|, dot product 3 eps |a||b|, the f32 difference
    // o - v0 eps |t|: 6.5 eps |d||e1||e2| for det, 7.5 eps |d||t||e| for the numerators; 10 leaves a margin)
    const double e_det = 10.0 * kEps32 * dl * l1 * l2;
    const double e_num = 10.0 * kEps32 * dl * dmax * lmax;
    const double thr = 1e-6 * (1.0 - 1e-5) - e_det;
    const double dmin = origin_triangle_distance(a, b, c);
    const double h = ln > 0 ? fabs(dot(n, a)) / ln : 0.0;   // distance from O to the triangle's plane
    // world-space distance (in the triangle's plane) by which an accepted hit may lie outside the exact triangle:
    // two bounds, either is valid.  (1) the true determinant of an accepted ray is >= thr;  (2) the ray meets the
    // plane at distance D_P <= dmax + slop from O, where its true determinant is dl |n| h / D_P.
    double slop = INFINITY;
    if (thr > 0) {
        slop = (e_num + e_det) / thr * (l1 + l2);
        // accepted rays meet the plane no farther than h dl |n| / thr from O: if even that (plus the slop) does not
        // reach the triangle, no ray through O is ever accepted - the silhouette triangles of a fine mesh
        if ((h + P.ray_offset) * dl * ln / thr + slop + P.ray_offset < dmin * (1.0 - 1e-6)) {
            fp.skip = true;
            return fp;
        }
    }
    if (ln > 0 && h > P.ray_offset) {
        const double k = (e_num + e_det) * (l1 + l2) / (dl * ln * (h - P.ray_offset));   // slop = k (dmax + slop)
        if (k < 0.5) slop = og_min(slop, k * dmax / (1.0 - k));
    }
    const double reach = dmin - slop - P.ray_offset;   // nearest point of O at which a ray can be accepted
    if (!(reach >= P.near_radius) || !(reach > 0.25 * dmin)) {
        fp.global = true;
        return fp;
    }
    const double cell_angle = 2.0 / P.res;   // at the face centre, where a cell subtends the largest angle
    // a tangent-plane coordinate moves by up to 3x the angle (du/dtheta = 1 + u^2 <= 2 on an axis, corners beyond)
    const double margin = P.base_margin + 3.0 * ((slop + P.ray_offset) / reach) / cell_angle;
    if (!(margin <= P.max_margin)) {
        fp.global = true;
        return fp;
    }
    fp.n = 3;
    fp.poly[0] = a;
    fp.poly[1] = b;
    fp.poly[2] = c;
    fp.margin = margin;
    fp.mindist = round_down((reach - P.abs_slack) * (1.0 - 1e-5));
    return fp;
}

// Orthographic grid: every ray has the direction ray_len * axis_w and starts anywhere in the scene.  The stored key
// is MINUS an upper bound of the primitive's depth along axis_w (so that lists ascend like the origin grids' and
// "key > -depth(ray origin)" ends the scan: such a primitive lies entirely behind the ray's start).
OG_HD inline Footprint triangle_footprint_ortho(const GridParams& P, const float* v) {
    Footprint fp;
    Vec a{v[0], v[1], v[2]}, b{v[8], v[9], v[10]}, c{v[16], v[17], v[18]};
    Vec e1 = b - a, e2 = c - a, n = cross(e1, e2);
    const double l1 = len(e1), l2 = len(e2), ln = len(n), lmax = og_max(l1, l2);
    if (!og_finite(len(a)) || !og_finite(len(b)) || !og_finite(len(c)) || !og_finite(ln)) {
        fp.global = true;
        return fp;
    }
    // the same error model as triangle_footprint(); the determinant is the same for every ray: dl (n . axis_w) - and
    // since the direction is known, so are the magnitudes of the products that are rounded on the way to it
    // (p = d x e2: two products and a difference per component; det = e1 . p: three products, two sums).  An axis-
    // aligned face seen exactly edge-on by an axis-aligned light gets the bound 0: its f32 determinant IS 0.
    const double dl = P.ray_len;
    const Vec dv = P.axis_w * dl;
    const double pa[3] = {fabs(dv.y * e2.z) + fabs(dv.z * e2.y), fabs(dv.z * e2.x) + fabs(dv.x * e2.z),
                          fabs(dv.x * e2.y) + fabs(dv.y * e2.x)};
    const Vec pv = cross(dv, e2);
    const double e_det = 2.0 * kEps32 * (2.0 * (fabs(e1.x) * pa[0] + fabs(e1.y) * pa[1] + fabs(e1.z) * pa[2]) +
                                         3.0 * (fabs(e1.x * pv.x) + fabs(e1.y * pv.y) + fabs(e1.z * pv.z)));
    const double e_num = 10.0 * kEps32 * dl * P.reach_max * lmax;
    const double thr = 1e-6 * (1.0 - 1e-5) - e_det;
    const double det = dl * fabs(dot(n, P.axis_w));
    if (thr > 0 && det < thr) {   // (edge-on to the light: the f32 test rejects it for every ray)
        fp.skip = true;
        return fp;
    }
    const double det_lo = og_max(det - e_det, thr > 0 ? thr : 0.0);
    const double slop = det_lo > 0 ? (e_num + e_det) / det_lo * (l1 + l2) : INFINITY;
    const double margin = P.base_margin + slop * P.cells_per_unit;
    if (!(margin <= P.max_margin)) {
        fp.global = true;
        return fp;
    }
    fp.n = 3;
    fp.poly[0] = a;
    fp.poly[1] = b;
    fp.poly[2] = c;
    fp.margin = margin;
    const double depth = og_max(dot(a, P.axis_w), og_max(dot(b, P.axis_w), dot(c, P.axis_w))) + slop + P.abs_slack;
    float key = (float)-depth;
    if ((double)key > -depth) key = nextafterf(key, -INFINITY);   // round towards "deeper"
    fp.mindist = key;
    return fp;
}

OG_HD inline Footprint sphere_footprint_ortho(const GridParams& P, const pt_model& mo) {
    Footprint fp;
    Vec c{mo.center[0], mo.center[1], mo.center[2]};
    const double r = fabs((double)mo.radius);
    if (!og_finite(len(c)) || !og_finite(r)) {
        fp.global = true;
        return fp;
    }
    // silhouette: a disc of radius r around the centre; the f32 discriminant of a grazing ray moves it by ~ eps * reach
    const double slop = 64.0 * kEps32 * (P.reach_max + r);
    const double rad = (r * 1.002 + slop) / kCosPi8;
    fp.n = 8;
    for (int k = 0; k < 8; ++k) {
        fp.poly[k] = c + P.axis_u * (rad * oct_cos(k)) + P.axis_v * (rad * oct_sin(k));
    }
    fp.margin = P.base_margin;
    const double depth = dot(c, P.axis_w) + r + slop + P.abs_slack;
    float key = (float)-depth;
    if ((double)key > -depth) key = nextafterf(key, -INFINITY);
    fp.mindist = key;
    return fp;
}

OG_HD inline Footprint sphere_footprint(const GridParams& P, const pt_model& mo) {
    Footprint fp;
    Vec c{mo.center[0], mo.center[1], mo.center[2]};
    c = c - P.origin;
    double r = fabs((double)mo.radius), D = len(c);
    if (!og_finite(D) || !og_finite(r)) {
        fp.global = true;
        return fp;
    }
    double dmin = D - r;
    // the silhouette cone has sin(alpha) = r / D; the f32 discriminant moves a grazing silhouette by ~2 eps / alpha,
    // and the ray offset by ray_offset / dmin: both go into the margin; a cone wider than ~80 degrees is not worth it
    if (!(dmin >= P.near_radius) || !(dmin > 0.02 * D)) {
        fp.global = true;
        return fp;
    }
    double sin_a = r / D, tan_a = sin_a / sqrt(1.0 - sin_a * sin_a);
    double cell_angle = 2.0 / P.res;
    double slop_angle = sin_a > 0 ? 64.0 * kEps32 / sin_a : 0.0;
    double margin = P.base_margin + 3.0 * (slop_angle + P.ray_offset / dmin) / cell_angle;
    if (!(margin <= P.max_margin)) {
        fp.global = true;
        return fp;
    }
    // octagon around the cone's cross-section in the plane through the centre, perpendicular to the axis
    Vec axis = c * (1.0 / D);
    Vec t = fabs(axis.x) < 0.6 ? Vec{1, 0, 0} : Vec{0, 1, 0};
    Vec u = cross(axis, t);
    u = u * (1.0 / len(u));
    Vec w = cross(axis, u);
    double rad = D * tan_a * 1.002 / kCosPi8;
    fp.n = 8;
    for (int k = 0; k < 8; ++k) {
        fp.poly[k] = c + u * (rad * oct_cos(k)) + w * (rad * oct_sin(k));
    }
    fp.margin = margin;
    fp.mindist = round_down((dmin - P.ray_offset - P.abs_slack - 64.0 * kEps32 * D) * (1.0 - 1e-5));
    return fp;
}

// Clip polygon (in, n) against the half-space k . p >= 0 (a plane through the origin).
OG_HD inline int clip_plane(const Vec* in, int n, Vec k, Vec* out) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        Vec p = in[i], q = in[(i + 1) % n];
        double dp = dot(k, p), dq = dot(k, q);
        if (dp >= 0) out[m++] = p;
        if ((dp >= 0) != (dq >= 0)) {
            double t = dp / (dp - dq);
            out[m++] = p + (q - p) * t;
        }
    }
    return m;
}

// Calls emit(base + iy * R + ix) for every cell whose margin-grown square the convex polygon (px, py) - in cell units -
// may touch.  Edge functions of the polygon: a cell is dropped when its grown square lies entirely outside one edge;
// slivers (no reliable orientation) and tiny boxes keep their whole bounding box.
template <class Emit>
OG_HD inline void cover_cells(const double* px, const double* py, int n, bool all, double m, uint32_t R, size_t base, Emit&& emit,
                            uint32_t row0 = 0, uint32_t row_step = 1) {
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int i = 0; i < n && !all; ++i) {
        x0 = og_min(x0, px[i]);
        x1 = og_max(x1, px[i]);
        y0 = og_min(y0, py[i]);
        y1 = og_max(y1, py[i]);
    }
    if (all) {
        x0 = y0 = 0;
        x1 = y1 = R;
    }
    auto cell_lo = [&](double v) { return (uint32_t)og_min(R - 1, og_max(0.0, floor(v - m))); };
    auto cell_hi = [&](double v) { return (uint32_t)og_min(R - 1, og_max(0.0, floor(v + m))); };
    uint32_t ix0 = cell_lo(x0), ix1 = cell_hi(x1), iy0 = cell_lo(y0), iy1 = cell_hi(y1);
    double area2 = 0;
    if (!all)
        for (int i = 0; i < n; ++i) {
            int j = (i + 1) % n;
            area2 += px[i] * py[j] - px[j] * py[i];
        }
    const bool trim = !all && (ix1 - ix0 >= 2 || iy1 - iy0 >= 2) && fabs(area2) > 1e-6;
    const double orient = area2 > 0 ? 1.0 : -1.0;
    // (row0 / row_step: the rows of the bounding box are dealt out to the lanes of a wavefront on the device)
    for (uint32_t iy = iy0 + row0; iy <= iy1; iy += row_step)
        for (uint32_t ix = ix0; ix <= ix1; ++ix) {
            if (trim) {
                const double cx0 = ix - m, cx1 = ix + 1.0 + m, cy0 = iy - m, cy1 = iy + 1.0 + m;
                bool outside = false;
                for (int i = 0; i < n && !outside; ++i) {
                    int j = (i + 1) % n;
                    // inside(p) = orient * cross(edge, p - v_i) >= 0; take the corner that maximises it
                    double ex = px[j] - px[i], ey = py[j] - py[i];
                    double nx = -ey * orient, ny = ex * orient;   // inward normal
                    double cx = nx >= 0 ? cx1 : cx0, cy = ny >= 0 ? cy1 : cy0;
                    double val = nx * (cx - px[i]) + ny * (cy - py[i]);
                    // (an absolute epsilon in cell^2 units keeps touching cells)
                    if (val < -1e-9 * (fabs(nx) + fabs(ny)) * R) outside = true;
                }
                if (outside) continue;
            }
            emit(base + (size_t)iy * R + ix);
        }
}

// Calls emit(cell) for every cell of every face whose (margin-grown) square the projection of the footprint may touch.
template <class Emit>
OG_HD inline void rasterize(const GridParams& P, const Footprint& fp, Emit&& emit, uint32_t row0 = 0, uint32_t row_step = 1) {
    const uint32_t R = P.res;
    if (P.ortho) {   // parallel projection onto the plane (axis_u, axis_v): one "face"
        double px[kMaxPoly], py[kMaxPoly];
        for (int i = 0; i < fp.n; ++i) {
            px[i] = (dot(fp.poly[i], P.axis_u) - P.u0) * P.cells_per_unit;
            py[i] = (dot(fp.poly[i], P.axis_v) - P.v0) * P.cells_per_unit;
        }
        cover_cells(px, py, fp.n, false, fp.margin, R, 0, emit, row0, row_step);
        return;
    }
    const double half = 0.5 * R, m = fp.margin, mu = m * 2.0 / R;
    for (int face = 0; face < 6; ++face) {
        const int a = face >> 1, b = (a + 1) % 3, c = (a + 2) % 3;
        const double s = (face & 1) ? -1.0 : 1.0;
        // pyramid of the face, grown by the margin: |p_b| <= (1 + mu) s p_a, |p_c| <= (1 + mu) s p_a
        Vec buf[2][kMaxPoly];
        int n = fp.n, cur = 0;
        for (int i = 0; i < n; ++i) buf[0][i] = fp.poly[i];
        for (int side = 0; side < 4 && n > 0; ++side) {
            double k[3] = {0, 0, 0};
            k[a] = (1.0 + mu) * s;
            k[side < 2 ? b : c] = (side & 1) ? 1.0 : -1.0;
            n = clip_plane(buf[cur], n, Vec{k[0], k[1], k[2]}, buf[cur ^ 1]);
            cur ^= 1;
        }
        if (n == 0) continue;
        double px[kMaxPoly], py[kMaxPoly];
        bool all = false;
        for (int i = 0; i < n; ++i) {
            double wa = s * comp(buf[cur][i], a);
            if (!(wa > 1e-300)) {   // the footprint reaches O itself (near_radius normally prevents this)
                all = true;
                break;
            }
            px[i] = (comp(buf[cur][i], b) / wa + 1.0) * half;
            py[i] = (comp(buf[cur][i], c) / wa + 1.0) * half;
        }
        cover_cells(px, py, n, all, m, R, (size_t)face * R * R, emit, row0, row_step);
    }
}


}  // namespace og

// Parameters of a grid (host): what pth_origin_grid_build / pth_ortho_grid_build derive from the scene before any
// primitive is looked at.  `g` receives the header fields (kind, res, n_cells, axes ...); false = no grid (enabled = 0).
bool og_params_point(const pt_scene_desc& d, const float origin[3], uint32_t res, float ray_offset, float max_dir_len,
                     og::GridParams& P, ::pth_origin_grid& g);
bool og_params_ortho(const pt_scene_desc& d, const float direction[3], uint32_t res, og::GridParams& P, ::pth_origin_grid& g);
}  // namespace pth

// Origin grids: a second candidate filter in front of Triangle::intersect / Model::intersect, for
// the ray populations of the path that all pass through ONE point.
//
// In the reference every ray_cast() (src/renderer/utils.rs:11-21) goes through the kdtree-ray
// filter, whatever the ray.  Two populations are special, though:
//   * camera rays (src/renderer/mod.rs:114-124): the origin is camera column 3, bit for bit the
//     same for every sample of every pixel - 71 % of all closest-hit casts of BASELINE config 3;
//   * shadow rays towards a point light (src/renderer/mod.rs:301-331): origin = hit + normal * 1e-5,
//     direction = -normalize(hit - light): they all pass within 1e-5 * |normal| of the light.
// For rays through a common point O visibility is a 2-D problem: the directions around O are cut
// into the cells of a cube map (6 faces x res x res, uniform in the tangent plane), and every cell
// keeps the list of primitives whose projection from O overlaps it, sorted by their distance from
// O.  A cast is then ONE table lookup plus Möller–Trumbore over a handful of primitives - instead
// of ~20 dependent KD-node fetches taken by 64 diverging lanes - and neighbouring rays read the
// same lists.  The reference's result depends only on the SET of primitives whose intersect()
// succeeds, so the filter must be conservative, never exact:
//   * the projection is computed in f64 and grown by `margin` cells: a base margin for the f32
//     rounding of the device's cell lookup, plus, per triangle, the slop of the f32 Möller–Trumbore
//     itself (a grazing triangle accepts hits whose exact intersection point lies outside it by
//     ~16 eps D^2 |e1||e2| / (|n| h), h = distance from O to the triangle's plane), plus, for a
//     light, the angle by which a shadow ray can miss O;
//   * a triangle that no ray through O can hit with |det| >= 1e-6 (triangle.rs:48: the silhouette triangles of
//     a fine mesh seen edge-on) is left out; a primitive whose margin would exceed `PT_OG_MAX_MARGIN` cells, that
//     comes closer to O than `near_radius`, or that is degenerate goes to the GLOBAL list every ray tests;
//   * the stored distance of a primitive is a lower bound (f64 minimum distance, minus the slop,
//     rounded down), so "stop at the first primitive farther than the best hit / the light" is safe.
// A grid with too many global primitives is not built (enabled = 0) and the casts stay on the KD-tree.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

#include <sys/mman.h>

#include "host_common.hpp"

namespace pth {
namespace {

// The grids' two big arrays (400 MB - 1.6 GB of cell offsets, as much again in list entries): anonymous mappings with
// transparent huge pages asked for.  Faulted in 4 KB at a time by ~100 threads at once they cost 0.5 s of page-fault
// contention per grid (and slowed the KD build running beside them); 2 MB pages are 512x fewer faults.  Zero-filled.
size_t big_size(size_t bytes) { return (std::max<size_t>(bytes, 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); }
void* big_alloc(size_t bytes) {
    const size_t n = big_size(bytes) + ((size_t)2 << 20);   // room to align the start to 2 MB
    void* raw = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw == MAP_FAILED) throw std::bad_alloc();
    uintptr_t a = ((uintptr_t)raw + (((uintptr_t)2 << 20) - 1)) & ~(((uintptr_t)2 << 20) - 1);
    // (trim the unaligned head and the unused tail so that munmap(ptr, big_size) releases exactly the block)
    if (a > (uintptr_t)raw) munmap(raw, a - (uintptr_t)raw);
    const uintptr_t end = a + big_size(bytes), raw_end = (uintptr_t)raw + n;
    if (raw_end > end) munmap((void*)end, raw_end - end);
    madvise((void*)a, big_size(bytes), MADV_HUGEPAGE);
    return (void*)a;
}
void big_free(void* p, size_t bytes) {
    if (p) munmap(p, big_size(bytes));
}

constexpr double kEps32 = 5.9604644775390625e-8;  // 2^-24
constexpr int kMaxPoly = 24;                      // octagon clipped by four planes: <= 12 vertices

struct Vec {
    double x, y, z;
};
inline Vec operator-(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec operator+(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec operator*(Vec a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec cross(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double len(Vec a) { return std::sqrt(dot(a, a)); }
inline double comp(Vec a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// Distance from the origin to triangle (a, b, c) (Ericson, Real-Time Collision Detection 5.1.5).
double origin_triangle_distance(Vec a, Vec b, Vec c) {
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

float round_down(double v) {
    if (!(v > 0)) return 0.f;
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, 0.f);
    return f;
}

Footprint triangle_footprint(const GridParams& P, const float* v) {
    Footprint fp;
    Vec a{v[0], v[1], v[2]}, b{v[8], v[9], v[10]}, c{v[16], v[17], v[18]};
    a = a - P.origin;
    b = b - P.origin;
    c = c - P.origin;
    Vec e1 = b - a, e2 = c - a, n = cross(e1, e2);
    const double l1 = len(e1), l2 = len(e2), ln = len(n), lmax = std::max(l1, l2);
    const double dmax = std::max(len(a), std::max(len(b), len(c)));
    if (!std::isfinite(dmax) || !std::isfinite(ln)) {
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
    const double h = ln > 0 ? std::fabs(dot(n, a)) / ln : 0.0;   // distance from O to the triangle's plane
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
        if (k < 0.5) slop = std::min(slop, k * dmax / (1.0 - k));
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
Footprint triangle_footprint_ortho(const GridParams& P, const float* v) {
    Footprint fp;
    Vec a{v[0], v[1], v[2]}, b{v[8], v[9], v[10]}, c{v[16], v[17], v[18]};
    Vec e1 = b - a, e2 = c - a, n = cross(e1, e2);
    const double l1 = len(e1), l2 = len(e2), ln = len(n), lmax = std::max(l1, l2);
    if (!std::isfinite(len(a)) || !std::isfinite(len(b)) || !std::isfinite(len(c)) || !std::isfinite(ln)) {
        fp.global = true;
        return fp;
    }
    // the same error model as triangle_footprint(); the determinant is the same for every ray: dl (n . axis_w) - and
    // since the direction is known, so are the magnitudes of the products that are rounded on the way to it
    // (p = d x e2: two products and a difference per component; det = e1 . p: three products, two sums).  An axis-
    // aligned face seen exactly edge-on by an axis-aligned light gets the bound 0: its f32 determinant IS 0.
    const double dl = P.ray_len;
    const Vec dv = P.axis_w * dl;
    const double pa[3] = {std::fabs(dv.y * e2.z) + std::fabs(dv.z * e2.y), std::fabs(dv.z * e2.x) + std::fabs(dv.x * e2.z),
                          std::fabs(dv.x * e2.y) + std::fabs(dv.y * e2.x)};
    const Vec pv = cross(dv, e2);
    const double e_det = 2.0 * kEps32 * (2.0 * (std::fabs(e1.x) * pa[0] + std::fabs(e1.y) * pa[1] + std::fabs(e1.z) * pa[2]) +
                                         3.0 * (std::fabs(e1.x * pv.x) + std::fabs(e1.y * pv.y) + std::fabs(e1.z * pv.z)));
    const double e_num = 10.0 * kEps32 * dl * P.reach_max * lmax;
    const double thr = 1e-6 * (1.0 - 1e-5) - e_det;
    const double det = dl * std::fabs(dot(n, P.axis_w));
    if (thr > 0 && det < thr) {   // (edge-on to the light: the f32 test rejects it for every ray)
        fp.skip = true;
        return fp;
    }
    const double det_lo = std::max(det - e_det, thr > 0 ? thr : 0.0);
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
    const double depth = std::max(dot(a, P.axis_w), std::max(dot(b, P.axis_w), dot(c, P.axis_w))) + slop + P.abs_slack;
    float key = (float)-depth;
    if ((double)key > -depth) key = std::nextafterf(key, -INFINITY);   // round towards "deeper"
    fp.mindist = key;
    return fp;
}

Footprint sphere_footprint_ortho(const GridParams& P, const pt_model& mo) {
    Footprint fp;
    Vec c{mo.center[0], mo.center[1], mo.center[2]};
    const double r = std::fabs((double)mo.radius);
    if (!std::isfinite(len(c)) || !std::isfinite(r)) {
        fp.global = true;
        return fp;
    }
    // silhouette: a disc of radius r around the centre; the f32 discriminant of a grazing ray moves it by ~ eps * reach
    const double slop = 64.0 * kEps32 * (P.reach_max + r);
    const double rad = (r * 1.002 + slop) / std::cos(M_PI / 8);
    fp.n = 8;
    for (int k = 0; k < 8; ++k) {
        const double ang = 2 * M_PI * k / 8;
        fp.poly[k] = c + P.axis_u * (rad * std::cos(ang)) + P.axis_v * (rad * std::sin(ang));
    }
    fp.margin = P.base_margin;
    const double depth = dot(c, P.axis_w) + r + slop + P.abs_slack;
    float key = (float)-depth;
    if ((double)key > -depth) key = std::nextafterf(key, -INFINITY);
    fp.mindist = key;
    return fp;
}

Footprint sphere_footprint(const GridParams& P, const pt_model& mo) {
    Footprint fp;
    Vec c{mo.center[0], mo.center[1], mo.center[2]};
    c = c - P.origin;
    double r = std::fabs((double)mo.radius), D = len(c);
    if (!std::isfinite(D) || !std::isfinite(r)) {
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
    double sin_a = r / D, tan_a = sin_a / std::sqrt(1.0 - sin_a * sin_a);
    double cell_angle = 2.0 / P.res;
    double slop_angle = sin_a > 0 ? 64.0 * kEps32 / sin_a : 0.0;
    double margin = P.base_margin + 3.0 * (slop_angle + P.ray_offset / dmin) / cell_angle;
    if (!(margin <= P.max_margin)) {
        fp.global = true;
        return fp;
    }
    // octagon around the cone's cross-section in the plane through the centre, perpendicular to the axis
    Vec axis = c * (1.0 / D);
    Vec t = std::fabs(axis.x) < 0.6 ? Vec{1, 0, 0} : Vec{0, 1, 0};
    Vec u = cross(axis, t);
    u = u * (1.0 / len(u));
    Vec w = cross(axis, u);
    double rad = D * tan_a * 1.002 / std::cos(M_PI / 8);
    fp.n = 8;
    for (int k = 0; k < 8; ++k) {
        double ang = 2 * M_PI * k / 8;
        fp.poly[k] = c + u * (rad * std::cos(ang)) + w * (rad * std::sin(ang));
    }
    fp.margin = margin;
    fp.mindist = round_down((dmin - P.ray_offset - P.abs_slack - 64.0 * kEps32 * D) * (1.0 - 1e-5));
    return fp;
}

// Clip polygon (in, n) against the half-space k . p >= 0 (a plane through the origin).
int clip_plane(const Vec* in, int n, Vec k, Vec* out) {
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
void cover_cells(const double* px, const double* py, int n, bool all, double m, uint32_t R, size_t base, Emit&& emit) {
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int i = 0; i < n && !all; ++i) {
        x0 = std::min(x0, px[i]);
        x1 = std::max(x1, px[i]);
        y0 = std::min(y0, py[i]);
        y1 = std::max(y1, py[i]);
    }
    if (all) {
        x0 = y0 = 0;
        x1 = y1 = R;
    }
    auto cell_lo = [&](double v) { return (uint32_t)std::min<double>(R - 1, std::max(0.0, std::floor(v - m))); };
    auto cell_hi = [&](double v) { return (uint32_t)std::min<double>(R - 1, std::max(0.0, std::floor(v + m))); };
    uint32_t ix0 = cell_lo(x0), ix1 = cell_hi(x1), iy0 = cell_lo(y0), iy1 = cell_hi(y1);
    double area2 = 0;
    if (!all)
        for (int i = 0; i < n; ++i) {
            int j = (i + 1) % n;
            area2 += px[i] * py[j] - px[j] * py[i];
        }
    const bool trim = !all && (ix1 - ix0 >= 2 || iy1 - iy0 >= 2) && std::fabs(area2) > 1e-6;
    const double orient = area2 > 0 ? 1.0 : -1.0;
    for (uint32_t iy = iy0; iy <= iy1; ++iy)
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
                    if (val < -1e-9 * (std::fabs(nx) + std::fabs(ny)) * R) outside = true;
                }
                if (outside) continue;
            }
            emit(base + (size_t)iy * R + ix);
        }
}

// Calls emit(cell) for every cell of every face whose (margin-grown) square the projection of the footprint may touch.
template <class Emit>
void rasterize(const GridParams& P, const Footprint& fp, Emit&& emit) {
    const uint32_t R = P.res;
    if (P.ortho) {   // parallel projection onto the plane (axis_u, axis_v): one "face"
        double px[kMaxPoly], py[kMaxPoly];
        for (int i = 0; i < fp.n; ++i) {
            px[i] = (dot(fp.poly[i], P.axis_u) - P.u0) * P.cells_per_unit;
            py[i] = (dot(fp.poly[i], P.axis_v) - P.v0) * P.cells_per_unit;
        }
        cover_cells(px, py, fp.n, false, fp.margin, R, 0, emit);
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
        cover_cells(px, py, n, all, m, R, (size_t)face * R * R, emit);
    }
}

struct Builder {
    const pt_scene_desc& d;
    GridParams P;
    std::vector<Footprint> fps;       // per primitive
    std::vector<uint32_t> prim_word;  // primitive id | sphere bit

    Builder(const pt_scene_desc& desc, const GridParams& params) : d(desc), P(params) {}

    void footprints() {
        uint64_t n_prims = pth_prim_count(&d);
        fps.resize(n_prims);
        prim_word.resize(n_prims);
        uint64_t prim = 0;
        struct Span {   // a model, or 4096 triangles of a mesh (a scene is often ONE big mesh: a span per model was one thread)
            uint64_t prim0;
            uint32_t model, tri0, tri1;
        };
        std::vector<Span> spans;
        for (uint32_t m = 0; m < d.n_models; ++m) {
            if (d.models[m].kind == PT_MODEL_MESH) {
                for (uint32_t t0 = 0; t0 < d.models[m].tri_count; t0 += 4096u)
                    spans.push_back({prim + t0, m, t0, std::min(d.models[m].tri_count, t0 + 4096u)});
                prim += d.models[m].tri_count;
            } else {
                spans.push_back({prim, m, 0u, 0u});
                prim += 1;
            }
        }
        parallel(spans.size(), [&](size_t si, size_t) {
            const pt_model& mo = d.models[spans[si].model];
            uint64_t p = spans[si].prim0;
            if (mo.kind == PT_MODEL_MESH) {
                for (uint32_t t = spans[si].tri0; t < spans[si].tri1; ++t, ++p) {
                    const float* tri = d.triangles + (size_t)(mo.tri_first + t) * 24;
                    fps[p] = P.ortho ? triangle_footprint_ortho(P, tri) : triangle_footprint(P, tri);
                    prim_word[p] = (uint32_t)p;
                }
            } else {
                fps[p] = P.ortho ? sphere_footprint_ortho(P, mo) : sphere_footprint(P, mo);
                prim_word[p] = (uint32_t)p | 0x80000000u;
            }
        });
    }

    // fn(index, thread) over [0, n) on every hardware thread, dynamic chunks
    template <class F>
    static void parallel(size_t n, F&& fn, size_t chunk = 1) {
        unsigned nt = std::max(1u, std::min(96u, std::thread::hardware_concurrency()));
        if (const char* e = getenv("PT_HOST_THREADS")) nt = std::max(1, atoi(e));
        if (n <= chunk || nt == 1) {
            for (size_t i = 0; i < n; ++i) fn(i, 0);
            return;
        }
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                while (true) {
                    size_t b = next.fetch_add(chunk);
                    if (b >= n) break;
                    for (size_t i = b; i < std::min(n, b + chunk); ++i) fn(i, t);
                }
            });
        for (auto& x : th) x.join();
    }
};

uint32_t auto_resolution(uint64_t n_prims) {
    if (const char* e = getenv("PT_OG_RES")) {
        int v = atoi(e);
        if (v > 0) return (uint32_t)std::min(8192, std::max(8, v));
    }
    // cells about half the edge of a typical triangle: ~4 sqrt(n) cells across the 90 degrees of a face (4096 from
    // 260 k primitives, 8192 from 1.05 M: the 4 M-triangle translucent 4K frame of config 5 renders 16 % faster with
    // 8192 than with 4096, for 8 GB instead of 3.8 GB of device memory and 4.4 s instead of 1.6 s of build time)
    double want = 4.0 * std::sqrt((double)std::max<uint64_t>(1, n_prims));
    uint32_t r = 32;
    while (r < want && r < 8192) r *= 2;
    return r;
}

// Footprints -> lists: count, scan, fill, sort.  Shared by the cube-map grids around a point and the orthographic
// grids along a direction.
void fill_lists(const pt_scene_desc& d, const GridParams& P, pth_origin_grid& g, std::chrono::steady_clock::time_point t0) {
    const uint64_t n_prims = pth_prim_count(&d);
    const bool dbg = getenv("PT_DEBUG_SETUP") != nullptr;
    auto t_ph = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!dbg) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[grid %u] %-12s %.3f s\n", P.res, name, std::chrono::duration<double>(now - t_ph).count());
        t_ph = now;
    };
    Builder B(d, P);
    B.footprints();
    phase("footprints");
    std::vector<uint32_t> global;
    for (uint64_t p = 0; p < n_prims; ++p)
        if (B.fps[p].global) global.push_back((uint32_t)p);
    g.n_global = (uint32_t)global.size();
    uint32_t max_global = 64;
    if (const char* e = getenv("PT_OG_MAX_GLOBAL")) max_global = (uint32_t)std::max(0, atoi(e));
    if (global.size() > max_global) {
        g.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return;   // enabled = 0: too many primitives every ray would have to test
    }

    // ---- pass 1: count the references of every cell.  Layout trick: the array has n_cells + 2 words and the per-cell
    // word lives at index cell + 1: after the scan it holds the cell's START, the fill pass advances it, and what is left
    // is the cell's END = the next cell's start - i.e. read from index 0 the array IS the offset table (word 0 = the
    // number of global entries in front), with no second cursor array and no copy.
    const size_t off_bytes = (g.n_cells + 2) * 4;
    struct BigPtr {
        void* p;
        size_t bytes;
        ~BigPtr() { big_free(p, bytes); }
        void* release() { void* q = p; p = nullptr; return q; }
    };
    BigPtr counts{big_alloc(off_bytes), off_bytes};
    uint32_t* cnt = (uint32_t*)counts.p + 1;
    const size_t chunk = 256;
    Builder::parallel((n_prims + chunk - 1) / chunk, [&](size_t ci, size_t) {
        for (uint64_t p = ci * chunk; p < std::min<uint64_t>(n_prims, (ci + 1) * chunk); ++p) {
            const Footprint& fp = B.fps[p];
            if (fp.skip || fp.global) continue;
            rasterize(P, fp, [&](size_t cell) { __atomic_fetch_add(&cnt[cell], 1u, __ATOMIC_RELAXED); });
        }
    });
    phase("count");
    // ---- exclusive scan (the global list sits in front), in parallel: sums of 1 Mi-cell chunks, their prefix, then every
    // chunk scans itself and writes the fill cursors beside the offsets (one thread over 100 M cells, two fresh 400 MB
    // arrays faulted in by it, was 0.7 of the grid's 1.1 s)
    const uint64_t scan_chunk = 1ull << 20, n_chunks = (g.n_cells + scan_chunk - 1) / scan_chunk;
    std::vector<uint64_t> chunk_sum(n_chunks + 1, 0);
    Builder::parallel(n_chunks, [&](size_t ci, size_t) {
        uint64_t sum = 0;
        for (uint64_t c = ci * scan_chunk; c < std::min<uint64_t>(g.n_cells, (ci + 1) * scan_chunk); ++c) sum += cnt[c];
        chunk_sum[ci + 1] = sum;
    });
    chunk_sum[0] = global.size();
    for (uint64_t ci = 0; ci < n_chunks; ++ci) chunk_sum[ci + 1] += chunk_sum[ci];
    const uint64_t total = chunk_sum[n_chunks];
    if (total > 0xffffffffull) fail(PT_ERR_UNSUPPORTED, "origin grid: more than 2^32 references (lower PT_OG_RES)");
    Builder::parallel(n_chunks, [&](size_t ci, size_t) {
        uint64_t run = chunk_sum[ci];
        for (uint64_t c = ci * scan_chunk; c < std::min<uint64_t>(g.n_cells, (ci + 1) * scan_chunk); ++c) {
            const uint32_t n = cnt[c];
            cnt[c] = (uint32_t)run;
            run += n;
        }
    });
    g.n_refs = total;
    const size_t ref_bytes = std::max<uint64_t>(1, total) * sizeof(pth_grid_ref);
    BigPtr refs{big_alloc(ref_bytes), ref_bytes};
    phase("scan + alloc");
    pth_grid_ref* rf = (pth_grid_ref*)refs.p;
    for (size_t i = 0; i < global.size(); ++i) rf[i] = pth_grid_ref{B.prim_word[global[i]], 0.f};
    // ---- pass 2: fill (advances the cells' start words to their ends)
    uint32_t* cur = cnt;
    Builder::parallel((n_prims + chunk - 1) / chunk, [&](size_t ci, size_t) {
        for (uint64_t p = ci * chunk; p < std::min<uint64_t>(n_prims, (ci + 1) * chunk); ++p) {
            const Footprint& fp = B.fps[p];
            if (fp.skip || fp.global) continue;
            const pth_grid_ref r{B.prim_word[p], fp.mindist};
            rasterize(P, fp, [&](size_t cell) { rf[__atomic_fetch_add(&cur[cell], 1u, __ATOMIC_RELAXED)] = r; });
        }
    });
    phase("fill");
    const uint32_t* off = (const uint32_t*)counts.p;   // off[c] = start of cell c, off[c + 1] = its end
    ((uint32_t*)counts.p)[0] = (uint32_t)global.size();
    // ---- every list in ascending (distance, primitive) order: deterministic, and what the early exits need
    const size_t cell_chunk = 4096;
    std::atomic<uint32_t> longest{0};
    Builder::parallel((g.n_cells + cell_chunk - 1) / cell_chunk, [&](size_t ci, size_t) {
        uint32_t local_max = 0;
        for (uint64_t c = ci * cell_chunk; c < std::min<uint64_t>(g.n_cells, (ci + 1) * cell_chunk); ++c) {
            uint32_t b = off[c], e = off[c + 1];
            local_max = std::max(local_max, e - b);
            if (e - b > 1)
                std::sort(rf + b, rf + e, [](const pth_grid_ref& x, const pth_grid_ref& y) {
                    return x.mindist < y.mindist || (x.mindist == y.mindist && x.prim < y.prim);
                });
        }
        uint32_t seen = longest.load();
        while (local_max > seen && !longest.compare_exchange_weak(seen, local_max)) {
        }
    });
    phase("sort");
    g.max_cell_refs = longest.load();
    g.cell_off = (uint32_t*)counts.release();
    g.refs = (pth_grid_ref*)refs.release();
    g.enabled = 1;
    g.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void build(const pt_scene_desc& d, const float origin[3], uint32_t res, float ray_offset, float max_dir_len,
           pth_origin_grid& g) {
    auto t0 = std::chrono::steady_clock::now();
    memset(&g, 0, sizeof g);
    memcpy(g.origin, origin, 12);
    const uint64_t n_prims = pth_prim_count(&d);
    if (res == 0) res = auto_resolution(n_prims);
    if (res > 8192) fail(PT_ERR_INVALID, "origin grid: resolution %u too large", res);
    g.res = res;
    g.n_cells = 6ull * res * res;
    g.ray_offset = ray_offset;
    if (!std::isfinite(origin[0]) || !std::isfinite(origin[1]) || !std::isfinite(origin[2])) return;  // enabled = 0

    // scene extent (for the absolute slack of the stored distances)
    double ext = 0;
    for (uint64_t t = 0; t < d.n_triangles; ++t)
        for (int k = 0; k < 3; ++k)
            for (int a = 0; a < 3; ++a) {
                double v = std::fabs((double)d.triangles[t * 24 + k * 8 + a] - origin[a]);
                if (std::isfinite(v)) ext = std::max(ext, v);
            }
    for (uint32_t m = 0; m < d.n_models; ++m)
        if (d.models[m].kind == PT_MODEL_SPHERE)
            for (int a = 0; a < 3; ++a) {
                double v = std::fabs((double)d.models[m].center[a] - origin[a]) + std::fabs((double)d.models[m].radius);
                if (std::isfinite(v)) ext = std::max(ext, v);
            }
    GridParams P;
    P.origin = Vec{origin[0], origin[1], origin[2]};
    P.res = res;
    P.base_margin = 0.125;
    P.max_margin = 128.0;
    P.max_dir_len = max_dir_len;
    if (const char* e = getenv("PT_OG_MAX_MARGIN")) P.max_margin = std::max(0.5, atof(e));
    // (the f32 rounding of the normalised shadow-ray direction moves the far end of the ray by ~1e-7 of its length)
    P.ray_offset = ray_offset > 0 ? ray_offset + 4e-7 * ext : 0.0;
    P.abs_slack = 1e-6 * ext + 1e-30;
    // a ray that misses O by ray_offset deviates by ray_offset / distance: at most 1/8 cell beyond near_radius
    P.near_radius = std::max(1e-5 * ext, P.ray_offset > 0 ? 3.0 * P.ray_offset / (0.125 * 2.0 / res) : 0.0);

    fill_lists(d, P, g, t0);
}

void build_ortho(const pt_scene_desc& d, const float direction[3], uint32_t res, pth_origin_grid& g) {
    auto t0 = std::chrono::steady_clock::now();
    memset(&g, 0, sizeof g);
    g.kind = 1;
    const uint64_t n_prims = pth_prim_count(&d);
    if (res == 0) res = auto_resolution(n_prims);
    if (res > 8192) fail(PT_ERR_INVALID, "origin grid: resolution %u too large", res);
    g.res = res;
    g.n_cells = (uint64_t)res * res;
    Vec w{direction[0], direction[1], direction[2]};
    const double wl = len(w);
    if (!std::isfinite(wl) || !(wl > 0) || wl > 1e6) return;   // enabled = 0
    GridParams P;
    P.ortho = true;
    P.ray_len = wl * (1.0 + 1e-6);
    P.axis_w = w * (1.0 / wl);
    const Vec t = std::fabs(P.axis_w.x) < 0.6 ? Vec{1, 0, 0} : Vec{0, 1, 0};
    P.axis_u = cross(P.axis_w, t);
    P.axis_u = P.axis_u * (1.0 / len(P.axis_u));
    P.axis_v = cross(P.axis_w, P.axis_u);
    // the device computes the cell with the f32 images of the axes: use exactly those here
    auto f32v = [](Vec v) { return Vec{(double)(float)v.x, (double)(float)v.y, (double)(float)v.z}; };
    P.axis_u = f32v(P.axis_u);
    P.axis_v = f32v(P.axis_v);
    P.axis_w = f32v(P.axis_w);
    // bounds of the scene in (u, v) and its diameter
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto grow = [&](Vec p, double r) {
        if (!std::isfinite(len(p)) || !std::isfinite(r)) return;
        const double c[3] = {dot(p, P.axis_u), dot(p, P.axis_v), dot(p, P.axis_w)};
        const double q[3] = {p.x, p.y, p.z};
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], c[k] - r);
            hi[k] = std::max(hi[k], c[k] + r);
            bmin[k] = std::min(bmin[k], q[k] - r);
            bmax[k] = std::max(bmax[k], q[k] + r);
        }
    };
    for (uint64_t tr = 0; tr < d.n_triangles; ++tr)
        for (int k = 0; k < 3; ++k) grow(Vec{d.triangles[tr * 24 + k * 8], d.triangles[tr * 24 + k * 8 + 1], d.triangles[tr * 24 + k * 8 + 2]}, 0.0);
    for (uint32_t m = 0; m < d.n_models; ++m)
        if (d.models[m].kind == PT_MODEL_SPHERE)
            grow(Vec{d.models[m].center[0], d.models[m].center[1], d.models[m].center[2]}, std::fabs((double)d.models[m].radius));
    if (!(hi[0] >= lo[0])) return;   // nothing finite in the scene: enabled = 0
    const double diag = std::sqrt((bmax[0] - bmin[0]) * (bmax[0] - bmin[0]) + (bmax[1] - bmin[1]) * (bmax[1] - bmin[1]) +
                                  (bmax[2] - bmin[2]) * (bmax[2] - bmin[2]));
    const double span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-6 * (diag + 1e-30));
    P.res = res;
    P.cells_per_unit = (res - 2.0) / span;   // one cell of border on every side
    P.u0 = lo[0] - 1.0 / P.cells_per_unit;
    P.v0 = lo[1] - 1.0 / P.cells_per_unit;
    P.base_margin = 0.125 + 8.0 * kEps32 * (std::fabs(lo[0]) + std::fabs(hi[0]) + std::fabs(lo[1]) + std::fabs(hi[1])) * P.cells_per_unit;
    P.max_margin = 128.0;
    if (const char* e = getenv("PT_OG_MAX_MARGIN")) P.max_margin = std::max(0.5, atof(e));
    P.max_dir_len = P.ray_len;
    P.ray_offset = 0.0;
    P.reach_max = diag * 1.01 + 1e-4;   // rays start on surfaces of the scene (+ normal * 1e-5)
    P.abs_slack = 1e-5 * diag + 1e-30;  // (covers the f32 rounding of the ray origin's depth on the device)
    P.near_radius = 0.0;
    P.origin = Vec{0, 0, 0};
    memcpy(g.axis_u, std::array<float, 3>{(float)P.axis_u.x, (float)P.axis_u.y, (float)P.axis_u.z}.data(), 12);
    memcpy(g.axis_v, std::array<float, 3>{(float)P.axis_v.x, (float)P.axis_v.y, (float)P.axis_v.z}.data(), 12);
    memcpy(g.axis_w, std::array<float, 3>{(float)P.axis_w.x, (float)P.axis_w.y, (float)P.axis_w.z}.data(), 12);
    g.u0 = (float)P.u0;
    g.v0 = (float)P.v0;
    g.cells_per_unit = (float)P.cells_per_unit;
    // (u0, v0, cells_per_unit as the device will use them)
    P.u0 = g.u0;
    P.v0 = g.v0;
    P.cells_per_unit = g.cells_per_unit;
    fill_lists(d, P, g, t0);
}

}  // namespace
}  // namespace pth

extern "C" {

int pth_origin_grid_build(const pt_scene_desc* desc, const float origin[3], uint32_t res, float ray_offset,
                          float max_dir_len, pth_origin_grid* out) {
    return pth::guarded([&] {
        if (!desc || !origin || !out) pth::fail(PT_ERR_INVALID, "pth_origin_grid_build: null argument");
        if (!(ray_offset >= 0.f)) pth::fail(PT_ERR_INVALID, "pth_origin_grid_build: ray_offset must be >= 0");
        if (!(max_dir_len > 0.f && max_dir_len < 1e6f)) pth::fail(PT_ERR_INVALID, "pth_origin_grid_build: bad max_dir_len");
        pth::build(*desc, origin, res, ray_offset, max_dir_len, *out);
    });
}

uint32_t pth_origin_grid_auto_resolution(uint64_t n_prims) { return pth::auto_resolution(n_prims); }

int pth_ortho_grid_build(const pt_scene_desc* desc, const float direction[3], uint32_t res, pth_origin_grid* out) {
    return pth::guarded([&] {
        if (!desc || !direction || !out) pth::fail(PT_ERR_INVALID, "pth_ortho_grid_build: null argument");
        pth::build_ortho(*desc, direction, res, *out);
    });
}

void pth_origin_grid_free(pth_origin_grid* g) {
    if (!g) return;
    pth::big_free(g->cell_off, (g->n_cells + 2) * 4);
    pth::big_free(g->refs, std::max<uint64_t>(1, g->n_refs) * sizeof(pth_grid_ref));
    g->cell_off = nullptr;
    g->refs = nullptr;
    g->enabled = 0;
}

}  // extern "C"

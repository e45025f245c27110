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
#include "og_raster.h"

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

using namespace og;

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

bool params_point(const pt_scene_desc& d, const float origin[3], uint32_t res, float ray_offset, float max_dir_len,
                  GridParams& P, pth_origin_grid& g) {
    memset(&g, 0, sizeof g);
    memcpy(g.origin, origin, 12);
    const uint64_t n_prims = pth_prim_count(&d);
    if (res == 0) res = auto_resolution(n_prims);
    if (res > 8192) fail(PT_ERR_INVALID, "origin grid: resolution %u too large", res);
    g.res = res;
    g.n_cells = 6ull * res * res;
    g.ray_offset = ray_offset;
    if (!std::isfinite(origin[0]) || !std::isfinite(origin[1]) || !std::isfinite(origin[2])) return false;  // enabled = 0

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
    P = GridParams();
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

    return true;
}

void build(const pt_scene_desc& d, const float origin[3], uint32_t res, float ray_offset, float max_dir_len,
           pth_origin_grid& g) {
    auto t0 = std::chrono::steady_clock::now();
    GridParams P;
    if (params_point(d, origin, res, ray_offset, max_dir_len, P, g)) fill_lists(d, P, g, t0);
}

bool params_ortho(const pt_scene_desc& d, const float direction[3], uint32_t res, GridParams& P, pth_origin_grid& g) {
    memset(&g, 0, sizeof g);
    g.kind = 1;
    const uint64_t n_prims = pth_prim_count(&d);
    if (res == 0) res = auto_resolution(n_prims);
    if (res > 8192) fail(PT_ERR_INVALID, "origin grid: resolution %u too large", res);
    g.res = res;
    g.n_cells = (uint64_t)res * res;
    Vec w{direction[0], direction[1], direction[2]};
    const double wl = len(w);
    if (!std::isfinite(wl) || !(wl > 0) || wl > 1e6) return false;   // enabled = 0
    P = GridParams();
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
    if (!(hi[0] >= lo[0])) return false;   // nothing finite in the scene: enabled = 0
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
    return true;
}

void build_ortho(const pt_scene_desc& d, const float direction[3], uint32_t res, pth_origin_grid& g) {
    auto t0 = std::chrono::steady_clock::now();
    GridParams P;
    if (params_ortho(d, direction, res, P, g)) fill_lists(d, P, g, t0);
}

}  // namespace

bool og_params_point(const pt_scene_desc& d, const float origin[3], uint32_t res, float ray_offset, float max_dir_len,
                     og::GridParams& P, pth_origin_grid& g) {
    return params_point(d, origin, res, ray_offset, max_dir_len, P, g);
}
bool og_params_ortho(const pt_scene_desc& d, const float direction[3], uint32_t res, og::GridParams& P, pth_origin_grid& g) {
    return params_ortho(d, direction, res, P, g);
}
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

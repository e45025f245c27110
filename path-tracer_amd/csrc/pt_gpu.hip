// libptgpu.so — HIP kernels and the C ABI of include/ptgpu.h (gfx950 only).
//
// Integrators
//   wavefront (default)  pt_wavefront.h: k_wf_rng, then per bounce k_wf_trace (persistent),
//                        k_wf_shade, k_wf_shadow (persistent, on a side stream beside the next trace);
//                        k_accumulate adds the staged per-sample radiance in the reference's sample order.
//                        pt_grid.h: camera rays and the shadow rays of point lights are cast through origin
//                        grids (k_og_primary, k_og_shadow) instead of the KD-tree (PT_FLAG_NO_GRIDS: KD only)
//   megakernel           k_render<COUNT> (PT_FLAG_MEGAKERNEL): one lane per pixel, samples looped inside the
//                        lane (renderer/mod.rs:105-130), KD-tree only; the second, independent implementation
//                        the parity tests cross-check the wavefront integrator with
// Other kernels
//   k_postprocess     Renderer::post_processing (mod.rs:335-353)
//   k_assemble        scatter all-gathered packed tiles into a row-major image
//   k_debug           --debug-textures G-buffer pass (debug_renderer.rs:64-105)
//   k_stream_copy     achievable-HBM yardstick of the roofline (pt_measure_copy_bandwidth)
//   k_trace / k_trace_all / k_isect / k_rng / k_math   parity-test hooks
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "pt_integrator.h"
#include "pt_wavefront.h"
#include "pt_grid_kernels.h"
#include "pt_escape_build.h"
#include "pt_grid_build.h"
#include "pthost.h"

// ------------------------------------------------------------------ errors
namespace {
thread_local std::string g_err;

struct GpuError {
    int code;
    std::string msg;
};

[[noreturn]] void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw GpuError{code, buf};
}

#define HIP_CHECK(expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) fail(PT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <class F>
int guarded(F&& fn) {
    try {
        fn();
        return PT_OK;
    } catch (const GpuError& e) {
        g_err = e.msg;
        return e.code;
    } catch (const std::bad_alloc&) {
        g_err = "out of host memory";
        return PT_ERR_INVALID;
    } catch (const std::exception& e) {
        g_err = e.what();
        return PT_ERR_INVALID;
    }
}
}  // namespace

// ------------------------------------------------------------------ pixel mapping
// Thread -> pixel.  Grid: one 256-thread workgroup per quarter of a tile_w x
// tile_h tile (4 wavefronts, each an 8x8 pixel block).  Local tile lt is global
// tile k = tile_table[tile_k_base + lt] (the rank's tiles in ascending order; unsharded: k = lt).
struct PixelRef {
    uint32_t x, y;
    uint32_t global_index;  // x + y*W  (seed formula, mod.rs:107-112)
    uint32_t out_index;     // position in the packed output
    bool valid;
};

__device__ __forceinline__ PixelRef map_pixel(const RenderParams& P, const uint32_t* __restrict__ tile_offsets) {
    PixelRef r;
    const uint32_t per_tile = P.tile_w * P.tile_h;
    const uint32_t blocks_per_tile = per_tile / 256u;
    uint32_t lt = blockIdx.x / blocks_per_tile;
    uint32_t q = (blockIdx.x % blocks_per_tile) * 256u + threadIdx.x;
    uint32_t wave = q >> 6, lane = q & 63u;
    uint32_t waves_x = P.tile_w >> 3;
    uint32_t tx = (wave % waves_x) * 8u + (lane & 7u);
    uint32_t ty = (wave / waves_x) * 8u + (lane >> 3);
    uint32_t k = P.tile_k_base ? tile_offsets[P.tile_k_base + lt] : lt;
    uint32_t tile_x = k % P.tiles_x, tile_y = k / P.tiles_x;
    r.x = tile_x * P.tile_w + tx;
    r.y = tile_y * P.tile_h + ty;
    r.valid = tile_y < P.tiles_y && r.x < P.width && r.y < P.height;
    r.global_index = r.x + r.y * P.width;
    if (P.shard_count <= 1) {
        r.out_index = r.global_index;
    } else {
        uint32_t cw = min(P.tile_w, P.width - tile_x * P.tile_w);
        r.out_index = tile_offsets[lt] + ty * cw + tx;
    }
    return r;
}

template <bool COUNT>
__global__ __launch_bounds__(256) void k_render(DevScene S, RenderParams P, const uint32_t* __restrict__ tile_offsets,
                                                float* __restrict__ accum, DevCounters* __restrict__ ctr) {
    __shared__ uint32_t slab[16 * PT_RNG_BLOCK];
    PixelRef px = map_pixel(P, tile_offsets);
    if (!px.valid) return;
    const uint32_t tid = threadIdx.x;
    float* out = accum + (size_t)px.out_index * 3;
    f3 acc = mk3(0.f, 0.f, 0.f);
    if (P.sample_begin != 0) acc = mk3(out[0], out[1], out[2]);
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    uint32_t draws = 0;
    for (uint32_t s = P.sample_begin + 1; s <= P.sample_end; ++s) {
        PtRng rng;
        pt_rng_seed(rng, (uint64_t)s + (uint64_t)px.global_index * (uint64_t)P.samples);
        float r1 = pt_rng_f32(rng, slab, tid);
        float r2 = pt_rng_f32(rng, slab, tid);
        f3 o, d;
        primary_ray(S, px.x, px.y, P.width, P.height, r1, r2, o, d);
        f3 color = render_path<COUNT>(S, P.bounces, o, d, rng, slab, tid, lc);
        acc = acc + color;
        if (COUNT) draws += rng.draws;
    }
    out[0] = acc.x;
    out[1] = acc.y;
    out[2] = acc.z;
    if (COUNT) {
        atomicAdd(&ctr->samples, (unsigned long long)(P.sample_end - P.sample_begin));
        atomicAdd(&ctr->segments, (unsigned long long)lc.segments);
        atomicAdd(&ctr->shadow_rays, (unsigned long long)lc.shadow_rays);
        atomicAdd(&ctr->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&ctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&ctr->shaded_hits, (unsigned long long)lc.shaded);
        atomicAdd(&ctr->rng_draws, (unsigned long long)draws);
        atomicAdd(&ctr->restarts, (unsigned long long)lc.restarts);
    }
}

// accum[p] (+)= staging[0][p] + staging[1][p] + ... in sample order: the reference's
// `*pixel += color` once per sample pass (mod.rs:105,130).
__global__ __launch_bounds__(256) void k_accumulate(const float* __restrict__ staging, float* __restrict__ accum,
                                                    uint32_t n_local, uint32_t batch, int first,
                                                    const uint8_t* __restrict__ pixel_empty, float bg_r, float bg_g, float bg_b) {
    uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_local) return;
    f3 acc = mk3(0.f, 0.f, 0.f);
    if (!first) acc = mk3(accum[3 * (size_t)p], accum[3 * (size_t)p + 1], accum[3 * (size_t)p + 2]);
    if (pixel_empty != nullptr && pixel_empty[p]) {
        // camera-grid cull (k_cam_block_mask): every sample of this pixel is the background - the value the bounce-0
        // kernel would have staged (mod.rs:184-186 with the initial throughput and colour), added once per sample
        const f3 c = mk3(0.f, 0.f, 0.f) + mul_ew(mk3(1.f, 1.f, 1.f), mk3(bg_r, bg_g, bg_b));
        for (uint32_t s = 0; s < batch; ++s) acc = acc + c;
    } else {
        for (uint32_t s = 0; s < batch; ++s) {
            const float* v = staging + ((size_t)s * n_local + p) * 3;
            acc = acc + mk3(v[0], v[1], v[2]);
        }
    }
    accum[3 * (size_t)p] = acc.x;
    accum[3 * (size_t)p + 1] = acc.y;
    accum[3 * (size_t)p + 2] = acc.z;
}

__global__ __launch_bounds__(256) void k_postprocess(const float* __restrict__ accum, uint8_t* __restrict__ rgb8,
                                                     uint32_t n, uint32_t samples, int tonemap_type) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f3 c = mk3(accum[3 * (size_t)i], accum[3 * (size_t)i + 1], accum[3 * (size_t)i + 2]) / (float)samples;
    c = tonemap(tonemap_type, c);
    rgb8[3 * (size_t)i] = as_u8(pt_pow_inv_gamma(c.x) * 255.f);
    rgb8[3 * (size_t)i + 1] = as_u8(pt_pow_inv_gamma(c.y) * 255.f);
    rgb8[3 * (size_t)i + 2] = as_u8(pt_pow_inv_gamma(c.z) * 255.f);
}

// gathered: shard_count slices of slice_pixels packed pixels; tile_src[k] = rank (high 8 bits) and packed offset of
// global tile k inside its rank's slice... as two words: tile_src[2k] = rank, tile_src[2k + 1] = offset
__global__ __launch_bounds__(256) void k_assemble(const uint8_t* __restrict__ gathered, uint8_t* __restrict__ image,
                                                  const uint32_t* __restrict__ tile_src,
                                                  uint32_t width, uint32_t height,
                                                  uint32_t tile_w, uint32_t tile_h, uint32_t tiles_x,
                                                  uint64_t slice_pixels, uint32_t elem_bytes) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= width * height) return;
    uint32_t x = i % width, y = i / width;
    uint32_t tile_x = x / tile_w, tile_y = y / tile_h;
    uint32_t k = tile_y * tiles_x + tile_x;
    uint32_t cw = min(tile_w, width - tile_x * tile_w);
    uint64_t src = (uint64_t)tile_src[2 * (size_t)k] * slice_pixels + tile_src[2 * (size_t)k + 1] +
                   (uint64_t)(y - tile_y * tile_h) * cw + (x - tile_x * tile_w);
    const uint8_t* s = gathered + src * elem_bytes;
    uint8_t* d = image + (uint64_t)i * elem_bytes;
    for (uint32_t b = 0; b < elem_bytes; ++b) d[b] = s[b];
}

// The counts of one chunk, gathered into one line of the frame's statistics (render_device, frame plan).
__global__ void k_wf_stats(const WfCounters* __restrict__ ctr, uint32_t levels, uint32_t* __restrict__ out) {
    const uint32_t b = threadIdx.x;
    if (b >= levels) return;
    out[4 * b + 0] = ctr[b].queue_count;
    out[4 * b + 1] = ctr[b].shadow_count;
    out[4 * b + 2] = ctr[b].exact_count;
    out[4 * b + 3] = b == 0 ? ctr[0].overflow : ctr[b].offgrid_count;
}

// ------------------------------------------------------------------ test-hook kernels
__device__ __forceinline__ void store_hit(pt_hit& o, const RawHit& h) {
    o.prim = (int32_t)PT_PRIM_INDEX(h.pid);
    o.flags = (int32_t)h.flags;
    o.dist = h.key;
    o.u = (h.flags & 2u) ? 0.f : h.u;
    o.v = (h.flags & 2u) ? 0.f : h.v;
}

__global__ __launch_bounds__(256) void k_trace(DevScene S, const float* __restrict__ rays, uint64_t n,
                                               pt_hit* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f3 o = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    f3 d = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    RawHit h;
    if (next_hit<false>(S, o, d, -INFINITY, 0u, h, lc)) store_hit(out[i], h);
    else out[i] = pt_hit{-1, 0, 0.f, 0.f, 0.f};
}

__global__ __launch_bounds__(256) void k_trace_all(DevScene S, const float* __restrict__ rays, uint64_t n,
                                                   uint32_t max_hits, pt_hit* __restrict__ out,
                                                   uint32_t* __restrict__ counts) {
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f3 o = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    f3 d = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    RawHit h;
    float t_prev = -INFINITY;
    uint32_t ord_prev = 0, cnt = 0;
    while (cnt < 4096u && next_hit<false>(S, o, d, t_prev, ord_prev, h, lc)) {
        if (cnt < max_hits) store_hit(out[i * max_hits + cnt], h);
        ++cnt;
        t_prev = h.key;
        ord_prev = h.ord;
    }
    counts[i] = cnt;
    for (uint32_t j = cnt; j < max_hits; ++j) out[i * max_hits + j] = pt_hit{-1, 0, 0.f, 0.f, 0.f};
}

// render_debug_pixels (src/renderer/debug_renderer.rs:64-105): first hit of the pixel-centre ray
__global__ __launch_bounds__(256) void k_debug(DevScene S, uint32_t width, uint32_t height, uint8_t* __restrict__ planes,
                                               int* __restrict__ any_hit) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= width * height) return;
    uint32_t x = i % width, y = i / width;
    f3 o, d;
    primary_ray(S, x, y, width, height, 0.5f, 0.5f, o, d);  // screen = x + 0.5 (debug_renderer.rs:24-30)
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    RawHit h;
    if (!next_hit<false>(S, o, d, -INFINITY, 0u, h, lc)) return;
    Surface sf;
    make_surface(S, o, d, h, sf);
    MatSample ms;
    material_sample(S, sf.model, sf.sphere, sf.uv, ms);
    f3 n = shading_normal(S, sf);
    float ior = S.materials[sf.model].ior;
    const f3 one = mk3(1.f, 1.f, 1.f);
    f3 v[PT_DEBUG_PLANES] = {mk3(n.x * 0.5f + 0.5f, n.y * 0.5f + 0.5f, n.z * 0.5f + 0.5f), ms.albedo, one * ms.opacity,
                             one * ms.metalness, one * ms.roughness, ms.emissive, (one * ior) / 3.f};
    size_t npix = (size_t)width * height;
#pragma unroll
    for (int p = 0; p < PT_DEBUG_PLANES; ++p) {
        uint8_t* out = planes + ((size_t)p * npix + i) * 3;
        out[0] = as_u8(v[p].x * 255.f);
        out[1] = as_u8(v[p].y * 255.f);
        out[2] = as_u8(v[p].z * 255.f);
    }
    *any_hit = 1;
}

__global__ __launch_bounds__(256) void k_isect(const float* __restrict__ rays, const float* __restrict__ tris,
                                               uint64_t n, pt_hit* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    f3 o = mk3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    f3 d = mk3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    f3 v0 = ld3(tris + 9 * i), v1 = ld3(tris + 9 * i + 3), v2 = ld3(tris + 9 * i + 6);
    float dist, u, v;
    bool bf;
    if (isect_triangle(o, d, v0, v1 - v0, v2 - v0, dist, u, v, bf)) {
        // tex_coords of the unit-test triangle (uv0=(0,0), uv1=(1,0), uv2=(0,1); triangle.rs:165-184)
        f2 uv0 = {0.f, 0.f}, uv1 = {1.f, 0.f}, uv2 = {0.f, 1.f};
        f2 tc = uv0 + u * (uv1 - uv0) + v * (uv2 - uv0);
        out[i] = pt_hit{0, bf ? 1 : 0, dist, tc.x, tc.y};
    } else {
        out[i] = pt_hit{-1, 0, 0.f, 0.f, 0.f};
    }
}

__global__ __launch_bounds__(256) void k_rng(const uint64_t* __restrict__ seeds, uint64_t n_seeds, uint32_t n_words,
                                             uint32_t* __restrict__ out) {
    __shared__ uint32_t slab[16 * PT_RNG_BLOCK];
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_seeds) return;
    PtRng rng;
    pt_rng_seed(rng, seeds[i]);
    for (uint32_t w = 0; w < n_words; ++w) out[i * n_words + w] = pt_rng_next_u32(rng, slab, threadIdx.x);
}

// plain streaming copy, 16 B per lane per step: the achievable-HBM yardstick of the roofline
typedef float pt_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const float4* __restrict__ src4, float4* __restrict__ dst4, uint64_t n) {
    const pt_v4f* src = (const pt_v4f*)src4;
    pt_v4f* dst = (pt_v4f*)dst4;
    uint64_t i = (uint64_t)blockIdx.x * 1024u + threadIdx.x;   // 4 independent 16-byte loads in flight per lane
    if (i + 768u < n) {
        pt_v4f a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + 256u);
        pt_v4f c = __builtin_nontemporal_load(src + i + 512u), d = __builtin_nontemporal_load(src + i + 768u);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + 256u);
        __builtin_nontemporal_store(c, dst + i + 512u);
        __builtin_nontemporal_store(d, dst + i + 768u);
    } else {
        for (; i < n; i += 256u) dst[i] = src[i];
    }
}

// scattered 8- / 16-byte loads, eight independent ones in flight per lane (pt_measure_gather_rate)
template <int BYTES>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, uint32_t mask, uint32_t rounds,
                                                uint32_t* __restrict__ sink) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t idx[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x = x * 1664525u + 1013904223u;      // LCG: a different 16-byte slot per lane and load
            idx[k] = (x >> 8) & mask;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (BYTES == 16) {
                uint4 v = table[idx[k]];
                acc += v.x ^ v.w;
            } else {
                uint2 v = *(const uint2*)(table + idx[k]);
                acc += v.x ^ v.y;
            }
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;   // (keeps the loads alive)
}

__global__ __launch_bounds__(256) void k_math(int fn, const float* __restrict__ x, uint64_t n, float* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float v = x[i], r;
    switch (fn) {
        case 0: r = pt_pow_inv_gamma(v); break;
        case 1: r = pt_acosf(v); break;
        case 2: r = pt_sinf(v); break;
        case 3: r = pt_cosf(v); break;
        default: r = NAN;
    }
    out[i] = r;
}

// ------------------------------------------------------------------ host side
namespace {

struct TileMap {
    uint32_t tile_w, tile_h, tiles_x, tiles_y, count, rank;
    uint32_t n_local_tiles;
    uint64_t n_local;
    std::vector<uint32_t> offsets;  // n_local_tiles + 1 (only used when count > 1)
    std::vector<uint32_t> tiles;    // global number (ty * tiles_x + tx) of every local tile, ascending
};

void normalise_opts(const pt_profile& p, const pt_opts* in, pt_opts& o) {
    if (in) o = *in;
    else memset(&o, 0, sizeof o), o.device = -1;
    if (o.shard_count == 0) o.shard_count = 1;
    if (o.tile_w == 0) o.tile_w = 32;
    if (o.tile_h == 0) o.tile_h = 32;
    if (o.shard_rank >= o.shard_count) fail(PT_ERR_INVALID, "shard_rank %u >= shard_count %u", o.shard_rank, o.shard_count);
    if ((o.tile_w & 7u) || (o.tile_h & 7u) || ((o.tile_w * o.tile_h) & 255u))
        fail(PT_ERR_INVALID, "tile_w/tile_h must be multiples of 8 with tile_w*tile_h a multiple of 256");
    if (p.width == 0 || p.height == 0) fail(PT_ERR_INVALID, "profile resolution must be non-zero");
    if ((uint64_t)p.width * p.height >= (1ull << 31)) fail(PT_ERR_UNSUPPORTED, "image too large");
}

// Which rank renders tile (tx, ty): (tx + ty * stride) mod count - diagonal stripes, so that every rank takes tiles
// from every column and every row of the image.  (k mod count, the obvious rule, degenerates into vertical stripes
// whenever the tile columns are a multiple of count / 2: at 1920x1080 with 32x32 tiles and 8 ranks it gave each rank
// every fourth column, and the per-rank frame times of config 3 ranged from 6.7 to 8.1 ms.)  stride = the smallest
// odd number >= 3 that is coprime to count (1 for count <= 2: a checkerboard).
uint32_t tile_rank_stride(uint32_t count) {
    if (count <= 2) return 1;
    for (uint32_t s = 3;; s += 2) {
        uint32_t a = s, b = count;
        while (b) {
            uint32_t t = a % b;
            a = b;
            b = t;
        }
        if (a == 1) return s;
    }
}
inline uint32_t tile_rank(uint32_t tx, uint32_t ty, uint32_t count, uint32_t stride) {
    return (uint32_t)(((uint64_t)tx + (uint64_t)ty * stride) % count);
}

TileMap make_tile_map(const pt_profile& p, const pt_opts& o, uint32_t rank) {
    TileMap m;
    m.tile_w = o.tile_w;
    m.tile_h = o.tile_h;
    m.tiles_x = (p.width + o.tile_w - 1) / o.tile_w;
    m.tiles_y = (p.height + o.tile_h - 1) / o.tile_h;
    m.count = o.shard_count;
    m.rank = rank;
    const uint32_t stride = tile_rank_stride(m.count);
    uint64_t off = 0;
    for (uint32_t ty = 0; ty < m.tiles_y; ++ty)
        for (uint32_t tx = 0; tx < m.tiles_x; ++tx) {
            if (m.count > 1 && tile_rank(tx, ty, m.count, stride) != rank) continue;
            uint32_t cw = std::min(o.tile_w, p.width - tx * o.tile_w);
            uint32_t ch = std::min(o.tile_h, p.height - ty * o.tile_h);
            m.tiles.push_back(ty * m.tiles_x + tx);
            m.offsets.push_back((uint32_t)off);
            off += (uint64_t)cw * ch;
        }
    m.n_local_tiles = (uint32_t)m.tiles.size();
    m.offsets.push_back((uint32_t)off);
    m.n_local = off;
    return m;
}

struct DeviceBuffer {
    void* p = nullptr;
    size_t bytes = 0;
    void ensure(size_t n) {
        if (!try_ensure(n)) fail(PT_ERR_DEVICE, "hipMalloc of %zu bytes failed: out of device memory", n);
    }
    bool try_ensure(size_t n) {  // false (buffer released) when the device cannot provide n bytes
        if (n <= bytes) return true;
        release();
        if (hipMalloc(&p, n) != hipSuccess) {
            (void)hipGetLastError();  // clear the sticky error
            p = nullptr;
            return false;
        }
        bytes = n;
        return true;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
};

}  // namespace

struct pt_scene {
    int device = 0;
    DevScene dev{};
    std::vector<void*> allocations;
    pt_scene_info info{};
    bool ortho_light_grids = false;   // some light grid is orthographic (a directional light): kernel variants DIRL
    std::vector<uint32_t> host_prim_entry;
    std::vector<pth_origin_grid> grid_headers;   // device-built grids: [0] camera, [1 + i] light i (enabled = 0: none)
    std::vector<DevGrid> host_light_grids;
    struct BuiltGrids {   // what grids_on_device() produced for this scene (possibly while the KD-tree was still being built)
        bool done = false, all_lights = false, ortho = false;
        DevGrid cam{};
        std::vector<DevGrid> lights;
        uint32_t cam_res = 0, light_grids = 0;
        uint64_t refs = 0, bytes = 0;
        float seconds = 0.f;
    } built;
    mutable pt_timing timing{};
    mutable pt_counters counters{};
    mutable DeviceBuffer accum_scratch, counter_buf, staging_buf;
    // The queues of the chunk of work items in flight.  The shadow casts of bounce b run on a side stream
    // beside the trace of bounce b+1, so the tail of one persistent launch is filled by the other's head.
    struct WfPipe {
        DeviceBuffer queue[2], hits, shadow, contrib, ctr, rng[2], draws, offgrid, deferred, exact[2], block_mask;
        hipStream_t side = nullptr, side_wide = nullptr, side_exact = nullptr;
        hipEvent_t ev_shade = nullptr, ev_shadow = nullptr, ev_rng = nullptr, ev_chunk = nullptr, ev_trace = nullptr, ev_wide = nullptr,
                   ev_exact = nullptr, ev_exact_go = nullptr;
    };
    mutable WfPipe pipe;
    // What a frame of one configuration produced: records per queue and bounce, per chunk of work items.  A frame is a pure
    // function of (scene, profile, options) - the seeds are the pixels' - so the counts of one frame are those of every later
    // one: the FIRST frame of a configuration runs in chunks small enough for a fixed budget with every queue as long as the
    // chunk, the later ones get queues as long as the records that exist (render_device, "frame plan").
    struct FrameStats {
        uint32_t cap_items = 0, n_slots = 0, levels = 0;   // the chunking the numbers were taken with; (batch, chunk) slots; bounces + 2
        uint32_t* host = nullptr;                         // pinned: n_slots x levels x 4 words (queue, shadow, exact, offgrid | overflow)
        hipEvent_t done = nullptr;
        bool pending = false, valid = false, planned = false;
        bool stats_planned = false;   // the counts on their way were taken by a frame that ran the plan
        std::vector<uint32_t> first_item_of_slot;         // (which chunk a slot was)
        // the plan made from them: work items per chunk, records per queue / hit / shadow / exact array, the bounces whose
        // shadow casts go inline; fresh until its buffers have been allocated once (buffers much larger are given back then)
        uint32_t plan_cap = 0, plan_q[2] = {0, 0}, plan_h = 0, plan_s = 0, plan_e = 0;
        std::vector<uint8_t> plan_inline;
        std::vector<uint32_t> plan_last;   // per chunk of the plan: the last bounce that has a ray (later ones are not launched)
        bool plan_fresh = false, plan_failed = false;   // (failed: the device could not provide the plan's buffers)
        ~FrameStats() {
            if (host) (void)hipHostFree(host);
            if (done) (void)hipEventDestroy(done);
        }
    };
    mutable std::map<std::vector<uint64_t>, std::unique_ptr<FrameStats>> frame_stats;
    mutable DeviceBuffer stats_dev;
    mutable uint64_t queue_bytes_last = 0;   // bytes of the path queues of the last frame (pt_scene_get_info)
    mutable uint32_t queue_chunk_last = 0, frame_planned_last = 0;
    // escape masks: wanted (PT_ESCAPE), built when the scene has rendered `escape_after` frames of the default pipeline
    bool escape_wanted = false, escape_tried = false;
    uint32_t escape_after = 2;
    mutable uint32_t frames_rendered = 0;
    mutable int trace_blocks = 0, shadow_blocks = 0, n_cu = 0;
    // (experiment, pt_scene_set_cu_mask: the scene's own streams confined to these CUs, grids sized for their number)
    std::vector<uint32_t> cu_mask;
    mutable uint32_t last_mask_blocks = 0;   // blocks of the last frame's camera-grid cull table (0: no cull in that frame)
    mutable uint32_t wf_cap_ok = 0;   // largest queue capacity the device provided so far (0: not tried)
    // tile tables of the sharded renders, one per configuration (image size, rank, count, tile size) and never rewritten:
    // pt_render_device is asynchronous, two calls for different ranks may be in flight on the caller's streams at once
    typedef std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t> TileKey;
    mutable std::map<TileKey, std::unique_ptr<DeviceBuffer>> tile_tables;
    mutable std::vector<hipEvent_t> events;
    // One frame's launches as an instantiated hipGraph, per configuration (profile, options, output and queue addresses):
    // the ~35 launches, memsets and cross-stream waits of a frame are captured once and replayed with ONE hipGraphLaunch
    mutable std::map<std::vector<uint64_t>, hipGraphExec_t> graphs;
    mutable hipStream_t capture_stream = nullptr;

    ~pt_scene() {
        (void)hipSetDevice(device);
        frame_stats.clear();
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
        if (capture_stream) (void)hipStreamDestroy(capture_stream);
        for (void* p : allocations) (void)hipFree(p);
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipEvent_t e : {pipe.ev_shade, pipe.ev_shadow, pipe.ev_rng, pipe.ev_chunk, pipe.ev_trace, pipe.ev_wide, pipe.ev_exact, pipe.ev_exact_go})
            if (e) (void)hipEventDestroy(e);
        if (pipe.side_exact) (void)hipStreamDestroy(pipe.side_exact);
        if (pipe.side) (void)hipStreamDestroy(pipe.side);
        if (pipe.side_wide) (void)hipStreamDestroy(pipe.side_wide);
    }
    template <class T>
    const T* upload(const T* host, size_t count) {
        size_t bytes = std::max<size_t>(16, count * sizeof(T));
        void* d = nullptr;
        HIP_CHECK(hipMalloc(&d, bytes));
        allocations.push_back(d);
        if (count) HIP_CHECK(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
        info.device_bytes += bytes;
        return (const T*)d;
    }
};

namespace {

void select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) fail(PT_ERR_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device >= n) fail(PT_ERR_DEVICE, "device %d out of range (%d devices)", device, n);
    if (device >= 0) HIP_CHECK(hipSetDevice(device));
}

}  // namespace

// Everything pt_scene_create computes on the HOST: validated copies of the small tables, the KD-tree in device
// layout, the per-primitive arrays, the origin grids.  Built once per process and uploaded to as many devices as
// the caller renders on (the CLI's --devices, a multi-GPU host): the KD build and the grid builds are seconds of
// CPU work that do not depend on the device.
struct pt_prep {
    std::vector<pth_kd_node> nodes;            // treelet order (see below)
    std::vector<float4> leaf, attr, pos;
    std::vector<uint2> entry_lists;            // entry lists of the KD-tree (csrc/pt_wavefront.h trav_enter) ...
    std::vector<uint32_t> prim_entry;          // ... and the word (list offset << 6 | entries) of every primitive
    std::vector<pt_material> model_mat;
    std::vector<pt_texture> textures;
    std::vector<uint8_t> texels;
    std::vector<DevLight> lights;
    float lut[256];
    DevScene dev{};                            // scalars filled in; pointers are set per device
    pt_scene_info info{};
    struct Grid {
        pth_origin_grid g{};
        Grid() = default;
        Grid(const Grid&) = delete;
        Grid& operator=(const Grid&) = delete;
        ~Grid() { pth_origin_grid_free(&g); }
    };
    std::unique_ptr<Grid> cam_grid;
    std::vector<std::unique_ptr<Grid>> light_grids;
    bool all_lights_gridded = false;
    // grids built on the device at upload time (csrc/pt_grid_build.h; PT_OG_HOST=1: built here, on the host, as above):
    // what every grid needs beside the primitives - its parameters and header - and the primitives' geometry as the
    // footprints take it (9 floats: three positions, or centre + radius), with the id | sphere-bit word
    struct GridJob {
        pth::og::GridParams params;
        pth_origin_grid hdr{};
        bool valid = false;
    };
    bool device_grids = false;
    GridJob cam_job;
    std::vector<GridJob> light_jobs;
    std::vector<float> og_geom;
    std::vector<uint32_t> og_words;
    double og_budget = 0;
};

namespace {

void grids_on_device(const pt_prep& P, pt_scene& s);

// `early` / `early_device`: the scene this prep is made for, when there is exactly one (pt_scene_create): its origin grids
// are then built on the device by the grid thread WHILE the KD-tree is being built on the host.
void prep_create(const pt_scene_desc& d, pt_prep& P, pt_scene* early = nullptr, int early_device = -1) {
    const bool dbg_setup = getenv("PT_DEBUG_SETUP") != nullptr;
    auto t_sec = std::chrono::steady_clock::now();
    auto section = [&](const char* name) {
        if (!dbg_setup) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[prep] %-28s %.3f s\n", name, std::chrono::duration<double>(now - t_sec).count());
        t_sec = now;
    };
    // ---- validate
    for (uint32_t m = 0; m < d.n_models; ++m) {
        const pt_model& mo = d.models[m];
        if (mo.material < 0 || (uint32_t)mo.material >= d.n_materials) fail(PT_ERR_INVALID, "model %u: bad material index", m);
        if (mo.kind == PT_MODEL_MESH && (uint64_t)mo.tri_first + mo.tri_count > d.n_triangles)
            fail(PT_ERR_INVALID, "model %u: triangle range out of bounds", m);
        if (mo.kind != PT_MODEL_MESH && mo.kind != PT_MODEL_SPHERE) fail(PT_ERR_INVALID, "model %u: bad kind", m);
    }
    for (uint32_t m = 0; m < d.n_materials; ++m) {
        const pt_material& ma = d.materials[m];
        const int32_t tx[6] = {ma.tex_albedo, ma.tex_emissive, ma.tex_opacity, ma.tex_metalness, ma.tex_roughness, ma.tex_normal};
        const uint32_t ch[6] = {3, 3, 1, 1, 1, 3};
        for (int k = 0; k < 6; ++k) {
            if (tx[k] < 0) continue;
            if ((uint32_t)tx[k] >= d.n_textures) fail(PT_ERR_INVALID, "material %u: texture index out of range", m);
            const pt_texture& t = d.textures[tx[k]];
            if (t.channels != ch[k]) fail(PT_ERR_INVALID, "material %u: texture %d has %u channels, expected %u", m, tx[k], t.channels, ch[k]);
            if (t.width == 0 || t.height == 0 || t.offset + (uint64_t)t.width * t.height * t.channels > d.n_texel_bytes)
                fail(PT_ERR_INVALID, "texture %d: bad extent", tx[k]);
        }
    }

    // ---- per-primitive arrays
    uint64_t n_prims = pth_prim_count(&d);
    std::vector<float4>&attr = P.attr, &pos = P.pos;
    attr.resize(n_prims * 4);
    pos.resize(n_prims * 3);
    P.og_geom.resize(n_prims * 9);
    P.og_words.resize(n_prims);
    P.model_mat.resize(d.n_models);
    bool translucent = false;
    uint64_t prim = 0;
    for (uint32_t m = 0; m < d.n_models; ++m) {
        const pt_model& mo = d.models[m];
        P.model_mat[m] = d.materials[mo.material];
        if (P.model_mat[m].opacity != 1.0f || P.model_mat[m].tex_opacity >= 0) translucent = true;
        float mbits;
        memcpy(&mbits, &m, 4);
        if (mo.kind == PT_MODEL_MESH) {
            for (uint32_t t = 0; t < mo.tri_count; ++t, ++prim) {
                const float* v = d.triangles + (size_t)(mo.tri_first + t) * 24;
                const float *a = v, *b = v + 8, *c = v + 16;
                attr[prim * 4 + 0] = make_float4(a[3], a[4], a[5], a[6]);
                attr[prim * 4 + 1] = make_float4(b[3], b[4], b[5], a[7]);
                attr[prim * 4 + 2] = make_float4(c[3], c[4], c[5], b[6]);
                attr[prim * 4 + 3] = make_float4(b[7], c[6], c[7], mbits);
                float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
                float e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
                uint32_t pid = (uint32_t)prim;
                float pbits;
                memcpy(&pbits, &pid, 4);
                pos[prim * 3 + 0] = make_float4(a[0], a[1], a[2], pbits);
                pos[prim * 3 + 1] = make_float4(e1[0], e1[1], e1[2], e2[0]);
                pos[prim * 3 + 2] = make_float4(e2[1], e2[2], 0.f, 0.f);
                float* og = &P.og_geom[prim * 9];
                og[0] = a[0]; og[1] = a[1]; og[2] = a[2];
                og[3] = b[0]; og[4] = b[1]; og[5] = b[2];
                og[6] = c[0]; og[7] = c[1]; og[8] = c[2];
                P.og_words[prim] = pid;
            }
        } else {
            attr[prim * 4 + 0] = make_float4(mo.center[0], mo.center[1], mo.center[2], mo.radius);
            attr[prim * 4 + 1] = attr[prim * 4 + 2] = make_float4(0, 0, 0, 0);
            attr[prim * 4 + 3] = make_float4(0, 0, 0, mbits);
            uint32_t pid = (uint32_t)prim | PT_PRIM_SPHERE;
            float pbits;
            memcpy(&pbits, &pid, 4);
            pos[prim * 3 + 0] = make_float4(mo.center[0], mo.center[1], mo.center[2], pbits);
            pos[prim * 3 + 1] = make_float4(mo.radius, 0, 0, 0);
            pos[prim * 3 + 2] = make_float4(0, 0, 0, 0);
            float* og = &P.og_geom[prim * 9];
            og[0] = mo.center[0]; og[1] = mo.center[1]; og[2] = mo.center[2]; og[3] = mo.radius;
            og[4] = og[5] = og[6] = og[7] = og[8] = 0.f;
            P.og_words[prim] = pid;
            ++prim;
        }
    }
    // ---- kdtree-ray's slab test (scene_slab, pt_integrator.h): the exact bounding box of the scene - the union of
    // Model::bound() (model.rs:76-86: the positions' bounds for a mesh, centre -+ radius for a sphere).  A cast whose origin
    // is not strictly inside it runs the test once, at its end (hit_passes_slab).
    {
        DevScene& D = P.dev;
        for (int a = 0; a < 3; ++a) {
            D.slab_min[a] = INFINITY;
            D.slab_max[a] = -INFINITY;
        }
        for (uint32_t m = 0; m < d.n_models; ++m) {
            const pt_model& mo = d.models[m];
            const uint32_t cnt = mo.kind == PT_MODEL_MESH ? mo.tri_count : 1u;
            for (uint32_t t = 0; t < cnt; ++t) {
                for (int a = 0; a < 3; ++a) {
                    float lo, hi;
                    if (mo.kind == PT_MODEL_MESH) {
                        const float* v = d.triangles + (size_t)(mo.tri_first + t) * 24;
                        lo = fminf(fminf(v[a], v[8 + a]), v[16 + a]);
                        hi = fmaxf(fmaxf(v[a], v[8 + a]), v[16 + a]);
                    } else {
                        lo = mo.center[a] - mo.radius;
                        hi = mo.center[a] + mo.radius;
                    }
                    D.slab_min[a] = fminf(D.slab_min[a], lo);
                    D.slab_max[a] = fmaxf(D.slab_max[a], hi);
                }
            }
        }
        if (n_prims >= (1ull << 28)) fail(PT_ERR_UNSUPPORTED, "more than 2^28 primitives");   // (pack_hit's index bits)
    }
    section("primitive arrays, scene box");
    const uint64_t n_prims_early = pth_prim_count(&d);
    // ---- origin grids (host/origin_grid.cpp, csrc/pt_grid.h): camera rays, shadow rays of point lights.  They depend on the
    // scene description only, not on the KD-tree: built on a thread of their own BESIDE the KD build (both are seconds of
    // multi-threaded host work; setup of config 3: 3.2 -> 2.3 s).
    auto build_grids = [&P, &d, n_prims_early, early, early_device]() {
        const uint64_t n_prims = n_prims_early;
        const float* M = d.camera.transform;
        DevScene& D = P.dev;

        auto t_grid = std::chrono::steady_clock::now();
        static const bool grids_on = [] {
            const char* e = getenv("PT_OG");
            return !(e && *e && atoi(e) == 0);
        }();
        // longest camera-ray direction: |M dir| <= ||M||_F for the unit vector dir (mod.rs:122-123)
        double fro = 0;
        for (int k = 0; k < 3; ++k)
            for (int r = 0; r < 3; ++r) fro += (double)M[4 * k + r] * M[4 * k + r];
        fro = std::sqrt(fro);
        // Byte budget over ALL grids of the scene (PT_OG_BUDGET_GIB, default 48 of the 288 GB): the grids are an optional
        // accelerator in front of the KD-tree, one per camera and per light, 6 res^2 cells of 4 B plus ~1.5x that in list
        // entries each (8192^2: ~4 GB a grid) - a scene with many lights must not run the host or the device out of
        // memory over them.  The resolution is halved (down to 512) until the estimate fits; if it still does not, or
        // the grids as built exceed the budget, the lights go without (their shadow rays take the KD-tree).
        static const double budget = [] {
            const char* e = getenv("PT_OG_BUDGET_GIB");
            const double g = e && *e ? atof(e) : 48.0;
            return (g > 0 ? g : 48.0) * 1073741824.0;
        }();
        auto estimate = [](uint32_t r) { return 6.0 * r * r * 4.0 * 2.5; };
        uint32_t res = pth_origin_grid_auto_resolution(n_prims);
        const double n_grids = 1.0 + d.n_lights;
        while (res > 512u && estimate(res) * n_grids > budget) res >>= 1;
        bool lights_fit = estimate(res) * n_grids <= budget;
        // (PT_OG_RES_LIGHT: experiments - the light grids at a resolution of their own)
        const uint32_t light_res = [&] {
            const char* e = getenv("PT_OG_RES_LIGHT");
            return e && *e && atoi(e) >= 32 ? (uint32_t)atoi(e) : res;
        }();
        if (!lights_fit) {   // the camera grid alone, at the resolution it is worth having
            res = pth_origin_grid_auto_resolution(n_prims);
            while (res > 512u && estimate(res) > budget) res >>= 1;
        }
        double grid_bytes = 0;
        static const bool host_grids = [] {
            const char* e = getenv("PT_OG_HOST");
            return e && *e && atoi(e) != 0;
        }();
        if (!host_grids) {   // the device builds them at upload time: only the parameters are derived here
            P.device_grids = true;
            P.og_budget = budget;
            const float max_normal = 1.5f;
            if (grids_on && n_prims > 0 && fro > 0 && fro < 64.0 && estimate(res) <= budget)
                P.cam_job.valid = pth::og_params_point(d, M + 12, res, 0.f, (float)(fro * 1.001), P.cam_job.params, P.cam_job.hdr);
            bool all = grids_on && n_prims > 0 && lights_fit;
            for (uint32_t i = 0; i < d.n_lights && all; ++i) {
                pt_prep::GridJob job;
                if (d.lights[i].kind == PT_LIGHT_POINT) {
                    job.valid = pth::og_params_point(d, d.lights[i].vec, light_res, 1.05e-5f * max_normal, 1.001f, job.params, job.hdr);
                } else {
                    const float sd[3] = {-1.f * d.lights[i].vec[0], -1.f * d.lights[i].vec[1], -1.f * d.lights[i].vec[2]};
                    job.valid = pth::og_params_ortho(d, sd, light_res, job.params, job.hdr);
                }
                if (!job.valid) all = false;
                P.light_jobs.push_back(job);
            }
            if (!all) P.light_jobs.clear();
            P.all_lights_gridded = all;
            D.all_lights_gridded = all ? 1u : 0u;
            D.light_grid_max_normal2 = max_normal * max_normal;
            P.info.grid_build_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_grid).count();
            if (early) {
                select_device(early_device);
                int cur = 0;
                HIP_CHECK(hipGetDevice(&cur));
                early->device = cur;
                grids_on_device(P, *early);
            }
            return;
        }
        // (the camera grid on a thread of its own beside the light grids: every grid is its own count / scan / fill / sort)
        std::future<void> cam_done;
        if (grids_on && n_prims > 0 && fro > 0 && fro < 64.0 && estimate(res) <= budget) {
            P.cam_grid = std::make_unique<pt_prep::Grid>();
            cam_done = std::async(std::launch::async, [&P, &d, M, res, fro] {
                if (pth_origin_grid_build(&d, M + 12, res, 0.f, (float)(fro * 1.001), &P.cam_grid->g) != PT_OK)
                    fail(PT_ERR_INVALID, "origin grid (camera): %s", pth_last_error());
            });
        }
        struct JoinCam {
            std::future<void>& f;
            ~JoinCam() { if (f.valid()) f.wait(); }
        } join_cam{cam_done};
        // lights: the shadow queue is consumed by ONE kernel, so the grids serve the shadow rays only when EVERY
        // light has one - a cube map around a point light, an orthographic grid along a directional light.  A
        // point light's shadow ray starts n * 1e-5 off the line through the light (mod.rs:319): the grids' margin
        // covers |n| <= 1.5, longer normals take the KD-tree per surface.
        bool all = grids_on && n_prims > 0 && lights_fit;   // (no lights at all: vacuously)
        const float max_normal = 1.5f;
        for (uint32_t i = 0; i < d.n_lights && all; ++i) {
            P.light_grids.push_back(std::make_unique<pt_prep::Grid>());
            int rc;
            if (d.lights[i].kind == PT_LIGHT_POINT) {
                rc = pth_origin_grid_build(&d, d.lights[i].vec, light_res, 1.05e-5f * max_normal, 1.001f, &P.light_grids.back()->g);
            } else {   // the shadow rays run along -direction (mod.rs:291), as it is
                const float sd[3] = {-1.f * d.lights[i].vec[0], -1.f * d.lights[i].vec[1], -1.f * d.lights[i].vec[2]};
                rc = pth_ortho_grid_build(&d, sd, light_res, &P.light_grids.back()->g);
            }
            if (rc != PT_OK) fail(PT_ERR_INVALID, "origin grid (light %u): %s", i, pth_last_error());
            if (!P.light_grids.back()->g.enabled) all = false;
            grid_bytes += 4.0 * P.light_grids.back()->g.n_cells + 8.0 * P.light_grids.back()->g.n_refs;
            if (grid_bytes > budget) all = false;   // (the lists came out longer than estimated)
        }
        if (cam_done.valid()) {
            cam_done.get();
            if (P.cam_grid->g.enabled) {
                P.info.cam_grid_res = P.cam_grid->g.res;
                P.info.grid_refs += P.cam_grid->g.n_refs;
                grid_bytes += 4.0 * P.cam_grid->g.n_cells + 8.0 * P.cam_grid->g.n_refs;
                if (grid_bytes > budget) all = false;
            }
        }
        if (!all) P.light_grids.clear();
        for (auto& g : P.light_grids) P.info.grid_refs += g->g.n_refs;
        P.all_lights_gridded = all;
        D.all_lights_gridded = all ? 1u : 0u;
        D.light_grid_max_normal2 = max_normal * max_normal;
        P.info.light_grids = all ? d.n_lights : 0u;
        P.info.grid_build_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_grid).count();
    };
    std::future<void> grids_done = std::async(std::launch::async, build_grids);
    struct JoinGrids {   // (an exception on the way out must not leave the thread behind)
        std::future<void>& f;
        ~JoinGrids() { if (f.valid()) f.wait(); }
    } join_grids{grids_done};

    // ---- KD-tree
    pth_kdtree kd;
    if (pth_kd_build(&d, &kd) != PT_OK) fail(PT_ERR_INVALID, "KD build failed: %s", pth_last_error());
    std::unique_ptr<pth_kdtree, void (*)(pth_kdtree*)> kd_guard(&kd, pth_kd_free);
    if (kd.depth >= PT_KD_STACK) fail(PT_ERR_UNSUPPORTED, "KD-tree depth %u exceeds the traversal stack", kd.depth);

    section("KD build");
    section("slab box + edge marks");
    // leaf records in leaf-reference order
    P.leaf.resize(kd.n_refs * 3);
    for (uint64_t r = 0; r < kd.n_refs; ++r) {
        uint32_t p = kd.refs[r];
        P.leaf[r * 3 + 0] = pos[(size_t)p * 3 + 0];
        P.leaf[r * 3 + 1] = pos[(size_t)p * 3 + 1];
        P.leaf[r * 3 + 2] = pos[(size_t)p * 3 + 2];
    }
    // sRGB -> linear table: (c as f32 / 255.0).powf(2.2) with the host libm (material.rs:137-141)
    for (int c = 0; c < 256; ++c) P.lut[c] = powf((float)c / 255.0f, 2.2f);
    P.lights.resize(d.n_lights);
    for (uint32_t i = 0; i < d.n_lights; ++i) {
        P.lights[i].kind = d.lights[i].kind;
        memcpy(P.lights[i].vec, d.lights[i].vec, 12);
        memcpy(P.lights[i].color, d.lights[i].color, 12);
        P.lights[i].tame = 1u;
        for (int k = 0; k < 3; ++k)
            if (!(fabsf(d.lights[i].color[k]) < 1e30f)) P.lights[i].tame = 0u;  // also catches NaN
    }
    P.textures.assign(d.textures, d.textures + d.n_textures);
    P.texels.assign(d.texels, d.texels + d.n_texel_bytes);

    section("leaf records, tables");
    // ---- device node layout.  The builder emits DFS order (below child = next node); on the GPU the
    // walk is bound by cache-line round trips (a wave waits for the slowest of ~43 scattered node
    // fetches), so the nodes are re-laid out in treelets: sibling PAIRS are adjacent (children of a
    // node = pair, pair + 1) and the pairs of a 4-level subtree are packed consecutively, so that
    // one 128-byte line serves up to four steps of a walk.  Node words: interior (split,
    // pair << 2 | axis), leaf (first record, n << 2 | 3) as before.
    std::vector<pth_kd_node>& tre = P.nodes;
    tre.assign(std::max<uint64_t>(kd.n_nodes, 1) + 1, pth_kd_node{0u, 0u});
    std::vector<uint32_t> new_index(kd.n_nodes, 0xffffffffu);   // builder (DFS) node number -> device slot
    {
        const pth_kd_node* N = kd.nodes;
        const int H = 4;  // treelet height: 2 + 4 + 8 = 14 nodes = 112 B below the treelet root pair
        if (kd.n_nodes == 0) {
            tre[0] = pth_kd_node{0u, 3u};
        } else {
            new_index[0] = 0;
            uint32_t next = 2;  // pairs start at even indices; slot 1 pads the root
            std::vector<uint32_t> cluster_roots{0}, frontier, level;
            size_t cr = 0;
            while (cr < cluster_roots.size()) {
                level.assign(1, cluster_roots[cr++]);
                for (int depth = 0; depth < H && !level.empty(); ++depth) {
                    frontier.clear();
                    for (uint32_t n : level) {
                        if ((N[n].w1 & 3u) == 3u) continue;
                        uint32_t below = n + 1, above = N[n].w1 >> 2;
                        new_index[below] = next;
                        new_index[above] = next + 1;
                        next += 2;
                        frontier.push_back(below);
                        frontier.push_back(above);
                    }
                    level.swap(frontier);
                }
                // whatever is left at the bottom of this treelet starts new treelets
                for (uint32_t n : level)
                    if ((N[n].w1 & 3u) != 3u) cluster_roots.push_back(n);
            }
            if (next > tre.size()) tre.resize(next);
            if (next >= (1u << 29)) fail(PT_ERR_UNSUPPORTED, "KD-tree has too many nodes");
            for (uint64_t n = 0; n < kd.n_nodes; ++n) {
                pth_kd_node nd = N[n];
                if ((nd.w1 & 3u) != 3u) nd.w1 = (new_index[n + 1] << 2) | (nd.w1 & 3u);
                tre[new_index[n]] = nd;
            }
            tre[1] = pth_kd_node{0u, 3u};
        }
    }
    section("treelet layout");
    // ---- entry lists (trav_enter, csrc/pt_wavefront.h).  A path's next ray starts ON the primitive it just hit
    // (origin = hit point + interpolated normal * 1e-5, mod.rs:266-268), deep inside the tree: of the ~23 nodes such a
    // cast visits, the first ~15 are the descent from the root to the small node around its origin - a chain of
    // dependent 8-byte fetches during which nothing is decided that the origin's whereabouts do not already say, except
    // which far children the ray will come back to.  So every primitive gets its HOME NODE - the deepest node whose box
    // holds every origin a hit on the primitive can produce - and the list of the home node's ancestors, root first:
    // (split, far child << 3 | near child is the below child << 2 | axis).  The cast reads that list (contiguous, all
    // loads in flight together), pushes the far children its ray reaches, and starts walking at the home node.  The
    // lists are shared by the primitives of a home node (a few MB in all).  Nothing here is load-bearing for
    // correctness: the cast checks that its origin lies on the near side of every listed plane and starts at the root
    // otherwise, so the region below is an estimate that only has to be right most of the time.
    P.prim_entry.assign(n_prims, 0u);
    P.entry_lists.clear();
    if (kd.n_nodes > 0 && n_prims > 0) {
        const pth_kd_node* N = kd.nodes;
        double ext = 0;
        for (int a = 0; a < 3; ++a)
            ext = std::max({ext, (double)fabsf(kd.bounds_min[a]), (double)fabsf(kd.bounds_max[a]), (double)fabsf(d.camera.transform[12 + a])});
        const float eps = (float)(2.5e-7 * std::max(ext, 1e-3));   // ~2 ulps of the largest coordinate: the rounding of o + d * t
        std::unordered_map<uint32_t, uint32_t> word_of_home;     // home node (builder numbering) -> entry word
        uint32_t path[PT_KD_STACK + 1];
        uint64_t q = 0;
        for (uint32_t m = 0; m < d.n_models; ++m) {
            const pt_model& mo = d.models[m];
            const uint32_t cnt = mo.kind == PT_MODEL_MESH ? mo.tri_count : 1u;
            for (uint32_t t = 0; t < cnt; ++t, ++q) {
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    if (mo.kind == PT_MODEL_MESH) {
                        const float* v = d.triangles + (size_t)(mo.tri_first + t) * 24;
                        const float nl = fminf(fminf(v[3 + a], v[11 + a]), v[19 + a]), nh = fmaxf(fmaxf(v[3 + a], v[11 + a]), v[19 + a]);
                        lo[a] = fminf(fminf(v[a], v[8 + a]), v[16 + a]) + 1.05e-5f * nl - eps;
                        hi[a] = fmaxf(fmaxf(v[a], v[8 + a]), v[16 + a]) + 1.05e-5f * nh + eps;
                    } else {   // a sphere's hits: on the surface, pushed 1e-5 out (entry hit) or in (exit hit, model.rs:49-62)
                        lo[a] = mo.center[a] - mo.radius - 1.05e-5f - eps;
                        hi[a] = mo.center[a] + mo.radius + 1.05e-5f + eps;
                    }
                }
                uint32_t node = 0, depth = 0;
                while (lo[0] == lo[0] && hi[0] == hi[0]) {   // (a NaN region stays at the root)
                    const pth_kd_node nd = N[node];
                    const uint32_t axis = nd.w1 & 3u;
                    if (axis == 3u || depth >= 63u) break;
                    float split;
                    memcpy(&split, &nd.w0, 4);
                    uint32_t next;
                    if (hi[axis] < split) next = node + 1u;            // below child = next node (builder layout)
                    else if (lo[axis] > split) next = nd.w1 >> 2;      // above child
                    else break;
                    path[depth++] = node;
                    node = next;
                }
                auto it = word_of_home.find(node);
                if (it == word_of_home.end()) {
                    uint32_t word = 0u;
                    if (depth > 0) {
                        if (P.entry_lists.size() & 1u) P.entry_lists.push_back(make_uint2(0u, 0u));   // 16-byte aligned lists
                        const uint64_t off = P.entry_lists.size();
                        if (off + depth >= (1ull << 26)) fail(PT_ERR_UNSUPPORTED, "entry lists exceed 2^26 entries");
                        for (uint32_t k = 0; k < depth; ++k) {
                            const uint32_t anc = path[k], child = k + 1 < depth ? path[k + 1] : node;
                            const bool near_below = child == anc + 1u;
                            const uint32_t far_host = near_below ? (N[anc].w1 >> 2) : anc + 1u;
                            P.entry_lists.push_back(make_uint2(N[anc].w0, (new_index[far_host] << 3) | (near_below ? 4u : 0u) | (N[anc].w1 & 3u)));
                        }
                        word = (uint32_t)(off << 6) | depth;
                    }
                    it = word_of_home.emplace(node, word).first;
                }
                P.prim_entry[q] = it->second;
            }
        }
    }
    if (P.entry_lists.size() & 1u) P.entry_lists.push_back(make_uint2(0u, 0u));
    for (int k = 0; k < 8; ++k) P.entry_lists.push_back(make_uint2(0u, 0u));   // (the batched loads of trav_enter read up to 8 entries past a list's start)
    DevScene& D = P.dev;
    D.n_lights = d.n_lights;
    D.n_prims = (uint32_t)n_prims;
    D.n_nodes = (uint32_t)kd.n_nodes;
    D.n_node_slots = (uint32_t)tre.size();
    D.has_translucent = translucent ? 1u : 0u;
    for (int a = 0; a < 3; ++a) {
        float pad = 1e-4f * std::max(fabsf(kd.bounds_min[a]), fabsf(kd.bounds_max[a])) + 1e-5f;
        D.bounds_min[a] = kd.bounds_min[a] - pad;
        D.bounds_max[a] = kd.bounds_max[a] + pad;
    }
    const float* M = d.camera.transform;
    memcpy(D.cam_c0, M, 12);
    memcpy(D.cam_c1, M + 4, 12);
    memcpy(D.cam_c2, M + 8, 12);
    memcpy(D.cam_c3, M + 12, 12);
    D.tan_half_fov = tanf(d.camera.fov / 2.f);  // Rad::tan(fov / 2.) (mod.rs:116,120)
    memcpy(D.background, d.background, 12);


    section("entry lists");
    grids_done.get();   // (rethrows what the grid thread threw)
    section("waiting for the grids");
    P.info.n_prims = n_prims;
    P.info.n_kd_nodes = kd.n_nodes;
    P.info.n_kd_leaves = kd.n_leaves;
    P.info.n_leaf_refs = kd.n_refs;
    P.info.kd_depth = kd.depth;
    P.info.has_translucent = translucent;
    P.info.kd_build_seconds = (float)kd.build_seconds;
}

// One origin grid built on the device (csrc/pt_grid_build.h) from the parameters / header prep_create derived.  Returns
// false when the grid is not to be had - too many primitives every ray would have to test, more than 2^32 list entries,
// the byte budget of the scene's grids exceeded, or the device out of memory: the casts it would have served take the
// KD-tree.  On success the two arrays belong to the scene (s.allocations).
bool device_grid_build(pt_scene& s, const pt_prep::GridJob& job, const float* d_geom, const uint32_t* d_words, uint32_t n_prims,
                       const std::vector<uint32_t>& words, double& bytes_used, double budget, DevGrid& out, pth_origin_grid& hdr_out,
                       uint64_t& dev_bytes) {
    memset(&out, 0, sizeof out);
    if (!job.valid || n_prims == 0) return false;
    static const uint32_t max_global = [] {
        const char* e = getenv("PT_OG_MAX_GLOBAL");
        return (uint32_t)(e && *e ? std::max(0, atoi(e)) : 64);
    }();
    const pth::og::GridParams& G = job.params;
    pth_origin_grid hdr = job.hdr;
    const uint64_t n_cells = hdr.n_cells;
    const bool dbg = getenv("PT_DEBUG_SETUP") != nullptr;
    auto t_ph = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (!dbg) return;
        (void)hipDeviceSynchronize();
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[device grid %u] %-10s %.4f s\n", hdr.res, name, std::chrono::duration<double>(now - t_ph).count());
        t_ph = now;
    };
    uint32_t *cnt_base = nullptr, *d_glob = nullptr, *d_bsum = nullptr;
    uint2* d_refs = nullptr;
    auto cleanup = [&]() {
        for (void* q : {(void*)cnt_base, (void*)d_glob, (void*)d_bsum, (void*)d_refs})
            if (q) (void)hipFree(q);
        (void)hipGetLastError();
    };
    try {
        const size_t off_bytes = (n_cells + 2) * 4;
        if (bytes_used + (double)off_bytes > budget) return false;
        HIP_CHECK(hipMalloc((void**)&cnt_base, off_bytes));
        HIP_CHECK(hipMemsetAsync(cnt_base, 0, off_bytes, 0));
        const uint32_t glob_cap = max_global + 1u;
        HIP_CHECK(hipMalloc((void**)&d_glob, (glob_cap + 2u) * 4));   // [0] counter, [1] longest list, [2 ...] the list
        HIP_CHECK(hipMemsetAsync(d_glob, 0, (glob_cap + 2u) * 4, 0));
        phase("alloc");
        const dim3 pg((n_prims + 3u) / 4u);   // a wavefront per primitive
        hipLaunchKernelGGL((ogb::k_og_raster<0>), pg, dim3(256), 0, 0, G, d_geom, d_words, n_prims, cnt_base + 1, (uint2*)nullptr,
                           d_glob + 2, d_glob, glob_cap);
        HIP_CHECK(hipGetLastError());
        phase("count");
        const uint32_t n_blocks = (uint32_t)((n_cells + OG_SCAN_BLOCK - 1) / OG_SCAN_BLOCK);
        HIP_CHECK(hipMalloc((void**)&d_bsum, (size_t)n_blocks * 4));
        hipLaunchKernelGGL(ogb::k_og_block_sum, dim3(n_blocks), dim3(256), 0, 0, (const uint32_t*)(cnt_base + 1), n_cells, d_bsum);
        HIP_CHECK(hipGetLastError());
        std::vector<uint32_t> glob(glob_cap + 2u), bsum(n_blocks);
        HIP_CHECK(hipMemcpy(glob.data(), d_glob, glob.size() * 4, hipMemcpyDeviceToHost));
        const uint32_t n_global = glob[0];
        hdr.n_global = n_global;
        if (n_global > max_global) {   // (enabled = 0, as the host builder)
            cleanup();
            return false;
        }
        HIP_CHECK(hipMemcpy(bsum.data(), d_bsum, (size_t)n_blocks * 4, hipMemcpyDeviceToHost));
        uint64_t run = n_global;
        for (uint32_t b = 0; b < n_blocks; ++b) {   // exclusive prefix of the block sums, 64-bit
            const uint32_t v = bsum[b];
            if (run > 0xffffffffull) break;
            bsum[b] = (uint32_t)run;
            run += v == 0xffffffffu ? 0x100000000ull : v;   // (a saturated block sum: the grid is given up below)
        }
        const uint64_t total = run;
        if (total > 0xffffffffull || bytes_used + (double)off_bytes + 8.0 * (double)total > budget) {
            cleanup();
            return false;
        }
        HIP_CHECK(hipMemcpy(d_bsum, bsum.data(), (size_t)n_blocks * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ogb::k_og_block_scan, dim3(n_blocks), dim3(256), 0, 0, cnt_base + 1, n_cells, (const uint32_t*)d_bsum);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMalloc((void**)&d_refs, std::max<uint64_t>(1, total) * 8));
        phase("scan");
        if (n_global) {   // the global block in front: ascending primitive, bound 0 (as the host builder)
            std::vector<uint32_t> g(glob.begin() + 2, glob.begin() + 2 + n_global);
            std::sort(g.begin(), g.end());
            std::vector<uint2> front(n_global);
            for (uint32_t i = 0; i < n_global; ++i) front[i] = make_uint2(words[g[i]], 0u);
            HIP_CHECK(hipMemcpy(d_refs, front.data(), (size_t)n_global * 8, hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL((ogb::k_og_raster<1>), pg, dim3(256), 0, 0, G, d_geom, d_words, n_prims, cnt_base + 1, d_refs, d_glob + 2,
                           d_glob, glob_cap);
        HIP_CHECK(hipGetLastError());
        phase("fill");
        HIP_CHECK(hipMemcpy(cnt_base, &n_global, 4, hipMemcpyHostToDevice));   // word 0: where cell 0 starts
        hipLaunchKernelGGL(ogb::k_og_sort, dim3((uint32_t)((n_cells + 255) / 256)), dim3(256), 0, 0, (const uint32_t*)cnt_base, n_cells,
                           d_refs, d_glob + 1);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(&hdr.max_cell_refs, d_glob + 1, 4, hipMemcpyDeviceToHost));   // (also the synchronisation point)
        phase("sort");
        (void)hipFree(d_glob);
        (void)hipFree(d_bsum);
        d_glob = d_bsum = nullptr;
        s.allocations.push_back(cnt_base);
        s.allocations.push_back(d_refs);
        const double bytes = (double)off_bytes + 8.0 * (double)std::max<uint64_t>(1, total);
        dev_bytes += (uint64_t)bytes;
        bytes_used += bytes;
        hdr.n_refs = total;
        hdr.enabled = 1;
        out.cell_off = cnt_base;
        out.refs = d_refs;
        out.res = hdr.res;
        out.n_global = n_global;
        out.half_res = 0.5f * (float)hdr.res;
        out.kind = hdr.kind;
        memcpy(out.axis_u, hdr.axis_u, 12);
        memcpy(out.axis_v, hdr.axis_v, 12);
        memcpy(out.axis_w, hdr.axis_w, 12);
        out.u0 = hdr.u0;
        out.v0 = hdr.v0;
        out.cells_per_unit = hdr.cells_per_unit;
        hdr_out = hdr;
        return true;
    } catch (const GpuError&) {
        cleanup();
        memset(&out, 0, sizeof out);
        return false;
    }
}

// Every origin grid of the scene, built on the device the calling thread has selected, into s.built (the arrays go to
// s.allocations).  Needs of the prep only what prep_create has ready BEFORE the KD build: the grid jobs and the
// primitives' geometry - so pt_scene_create runs it on a thread of its own beside the KD build.
void grids_on_device(const pt_prep& P, pt_scene& s) {
    auto t_grid = std::chrono::steady_clock::now();
    pt_scene::BuiltGrids& B = s.built;
    const size_t n_lights = P.light_jobs.size();
    B = pt_scene::BuiltGrids();
    B.lights.assign(n_lights, DevGrid{});
    for (auto& g : B.lights) memset(&g, 0, sizeof g);
    memset(&B.cam, 0, sizeof B.cam);
    s.grid_headers.assign(1 + std::max(n_lights, (size_t)0), pth_origin_grid{});
    const uint32_t n_prims = (uint32_t)P.og_words.size();
    float* d_geom = nullptr;
    uint32_t* d_words = nullptr;
    double used = 0;
    if ((P.cam_job.valid || n_lights) && n_prims > 0 && hipMalloc((void**)&d_geom, P.og_geom.size() * 4) == hipSuccess &&
        hipMalloc((void**)&d_words, P.og_words.size() * 4) == hipSuccess) {
        bool copied = hipMemcpy(d_geom, P.og_geom.data(), P.og_geom.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                      hipMemcpy(d_words, P.og_words.data(), P.og_words.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        if (copied && device_grid_build(s, P.cam_job, d_geom, d_words, n_prims, P.og_words, used, P.og_budget, B.cam, s.grid_headers[0], B.bytes)) {
            B.cam_res = s.grid_headers[0].res;
            B.refs += s.grid_headers[0].n_refs;
        }
        // the lights: all or none (the shadow queue is consumed by ONE kernel)
        bool all = copied && P.all_lights_gridded;
        const size_t mark = s.allocations.size();
        const uint64_t bytes_mark = B.bytes;
        uint64_t light_refs = 0;
        for (size_t i = 0; i < n_lights && all; ++i) {
            if (!device_grid_build(s, P.light_jobs[i], d_geom, d_words, n_prims, P.og_words, used, P.og_budget, B.lights[i], s.grid_headers[1 + i], B.bytes))
                all = false;
            else {
                light_refs += s.grid_headers[1 + i].n_refs;
                if (B.lights[i].kind != 0) B.ortho = true;
            }
        }
        if (!all) {   // drop whatever light grids were built
            for (size_t k = mark; k < s.allocations.size(); ++k) (void)hipFree(s.allocations[k]);
            s.allocations.resize(mark);
            B.bytes = bytes_mark;
            for (auto& g : B.lights) memset(&g, 0, sizeof g);
            for (size_t i = 1; i < s.grid_headers.size(); ++i) s.grid_headers[i] = pth_origin_grid{};
            B.ortho = false;
            light_refs = 0;
        }
        B.all_lights = all;
        B.light_grids = all ? (uint32_t)n_lights : 0u;
        B.refs += light_refs;
    } else {
        (void)hipGetLastError();
        B.all_lights = n_lights == 0 && P.all_lights_gridded;   // (no lights at all: vacuously)
    }
    if (d_geom) (void)hipFree(d_geom);
    if (d_words) (void)hipFree(d_words);
    B.seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_grid).count();
    B.done = true;
}

// Copy a prepared scene to `device`.
// The escape masks of a scene (pt_escape.h), built on its device from the uploaded arrays: one wavefront per primitive.  Blocks
// until they are there (every frame in flight has completed by then).  A device without the memory for them goes without.
void escape_masks_build(pt_scene& s) {
    s.escape_tried = true;
    DevScene& D = s.dev;
    const uint64_t n_prims = D.n_prims;
    auto t_esc = std::chrono::steady_clock::now();
    const size_t mark = s.allocations.size();
    const uint64_t bytes_mark = s.info.device_bytes;
    try {
        HIP_CHECK(hipSetDevice(s.device));
        void* buf = nullptr;
        HIP_CHECK(hipMalloc(&buf, n_prims * 80));
        s.allocations.push_back(buf);
        s.info.device_bytes += n_prims * 80;
        uint32_t* d_stats = nullptr;
        HIP_CHECK(hipMalloc((void**)&d_stats, 16));
        HIP_CHECK(hipMemset(d_stats, 0, 16));
        // the largest distance a ray of this scene covers before it hits anything: the box diagonal, or camera to far corner
        double diag2 = 0, cam2 = 0, amax = 0;
        for (int a = 0; a < 3; ++a) {
            const double w = (double)D.bounds_max[a] - D.bounds_min[a], c = D.cam_c3[a];
            const double far = std::max(std::fabs(c - D.bounds_min[a]), std::fabs(c - D.bounds_max[a]));
            diag2 += w * w;
            cam2 += far * far;
            amax = std::max({amax, std::fabs((double)D.bounds_min[a]), std::fabs((double)D.bounds_max[a]), std::fabs(c)});
        }
        const double reach = std::sqrt(std::max(diag2, cam2));
        EscBuildParams E{};
        E.delta_in = (float)((double)PT_SLACK_K * reach + 4.0 * 5.9604645e-8 * amax);
        E.slop_far = E.delta_in;
        E.alpha_stop = [] { const char* e = getenv("PT_ESCAPE_ALPHA"); return e && *e ? (float)atof(e) : 0.04f; }();
        E.r_near_scale = 2.0f;
        const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_prims + 3) / 4, 256u * 64u);
        hipLaunchKernelGGL(k_escape_build, dim3(blocks), dim3(256), 0, 0, D, E, (float4*)buf, (uint32_t)n_prims, d_stats);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        uint32_t st[4] = {0, 0, 0, 0};
        HIP_CHECK(hipMemcpy(st, d_stats, 16, hipMemcpyDeviceToHost));
        (void)hipFree(d_stats);
        D.escape = (const float4*)buf;
        s.info.escape_prims = st[0];
        s.info.escape_clear_fraction = st[0] ? (float)((double)st[1] / (384.0 * st[0])) : 0.f;
    } catch (const GpuError&) {   // (no memory for them: the casts are simply made)
        while (s.allocations.size() > mark) {
            (void)hipFree(s.allocations.back());
            s.allocations.pop_back();
        }
        s.info.device_bytes = bytes_mark;
        D.escape = nullptr;
    }
    s.info.escape_build_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_esc).count();
    // the counts of the frames rendered so far are upper bounds now (the masks end paths): the next frame counts again
    for (auto& kv : s.frame_stats) {
        pt_scene::FrameStats& f = *kv.second;
        f.pending = f.valid = f.planned = f.plan_failed = false;
    }
}

void scene_upload(const pt_prep& P, int device, pt_scene& s) {
    select_device(device);
    int cur = 0;
    HIP_CHECK(hipGetDevice(&cur));
    s.device = cur;
    auto t_up = std::chrono::steady_clock::now();
    s.info = P.info;
    s.info.device_bytes = 0;
    DevScene& D = s.dev;
    D = P.dev;
    D.kd_nodes = (const uint2*)s.upload(P.nodes.data(), P.nodes.size());
    D.leaf_prims = s.upload(P.leaf.data(), P.leaf.size());
    D.prim_attr = s.upload(P.attr.data(), P.attr.size());
    D.prim_pos = s.upload(P.pos.data(), P.pos.size());
    D.entry_lists = s.upload(P.entry_lists.data(), P.entry_lists.size());
    D.prim_entry = s.upload(P.prim_entry.data(), P.prim_entry.size());
    s.host_prim_entry = P.prim_entry;   // (test hook pt_trace_rays_wavefront)
    D.materials = s.upload(P.model_mat.data(), P.model_mat.size());
    D.textures = s.upload(P.textures.data(), P.textures.size());
    D.texels = s.upload(P.texels.data(), P.texels.size());
    D.srgb_lut = s.upload(P.lut, 256);
    D.lights = s.upload(P.lights.data(), P.lights.size());
    // ---- escape masks (pt_escape.h): built on the device from these arrays - not here: when the scene is about to render its
    // THIRD frame (escape_masks_build below).  They take 0.14 s for the 0.5 M primitives of config 3 and 1.3 s for the 4 M of
    // config 5 and return 3 ms and ~0.1 s per frame: a one-shot render (the CLI) is better off without them.
    D.escape = nullptr;
    s.info.escape_build_seconds = 0.f;
    s.info.escape_prims = 0;
    s.info.escape_clear_fraction = 0.f;
    {
        const char* esc_env = getenv("PT_ESCAPE");   // (read per scene: the tests switch it)
        const bool esc_on = !(esc_env && *esc_env && atoi(esc_env) == 0);
        const uint64_t n_prims = P.pos.size() / 3;
        s.escape_wanted = esc_on && n_prims > 0 && n_prims < (1ull << 28);
        const char* after = getenv("PT_ESCAPE_AFTER");   // frames a scene renders without them (0: built for the first frame)
        s.escape_after = after && *after ? (uint32_t)atoi(after) : 2u;
    }
    auto upload_grid = [&](const pth_origin_grid& g, DevGrid& out) {
        memset(&out, 0, sizeof out);
        if (!g.enabled) return;
        out.cell_off = s.upload(g.cell_off, g.n_cells + 1);
        out.refs = (const uint2*)s.upload(g.refs, std::max<uint64_t>(1, g.n_refs));
        out.res = g.res;
        out.n_global = g.n_global;
        out.half_res = 0.5f * (float)g.res;
        out.kind = g.kind;
        memcpy(out.axis_u, g.axis_u, 12);
        memcpy(out.axis_v, g.axis_v, 12);
        memcpy(out.axis_w, g.axis_w, 12);
        out.u0 = g.u0;
        out.v0 = g.v0;
        out.cells_per_unit = g.cells_per_unit;
    };
    if (P.device_grids) {   // built on this device (a prep is shared by the devices of a multi-GPU host: each builds its own)
        if (!s.built.done) grids_on_device(P, s);   // (pt_scene_create has them built beside the KD-tree already)
        const pt_scene::BuiltGrids& B = s.built;
        D.cam_grid = B.cam;
        std::vector<DevGrid> lgrids = B.lights;
        lgrids.resize(P.lights.size());
        D.all_lights_gridded = B.all_lights ? 1u : 0u;
        s.ortho_light_grids = B.ortho;
        s.info.cam_grid_res = B.cam_res;
        s.info.light_grids = B.light_grids;
        s.info.grid_refs = B.refs;
        s.info.device_bytes += B.bytes;
        s.info.grid_build_seconds += B.seconds;
        s.host_light_grids = lgrids;
        D.light_grids = s.upload(lgrids.data(), lgrids.size());
        s.info.upload_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_up).count();
        return;
    }
    // The grids are optional: a device that cannot hold them renders through the KD-tree (the light grids go first,
    // then the camera grid) instead of failing the scene.
    auto drop_allocations_from = [&](size_t mark, uint64_t bytes_mark) {
        for (size_t k = mark; k < s.allocations.size(); ++k) (void)hipFree(s.allocations[k]);
        s.allocations.resize(mark);
        s.info.device_bytes = bytes_mark;
        (void)hipGetLastError();
    };
    memset(&D.cam_grid, 0, sizeof D.cam_grid);
    if (P.cam_grid) {
        const size_t mark = s.allocations.size();
        const uint64_t bytes_mark = s.info.device_bytes;
        try {
            upload_grid(P.cam_grid->g, D.cam_grid);
        } catch (const GpuError&) {
            drop_allocations_from(mark, bytes_mark);
            memset(&D.cam_grid, 0, sizeof D.cam_grid);
            s.info.grid_refs -= P.cam_grid->g.enabled ? P.cam_grid->g.n_refs : 0;
            s.info.cam_grid_res = 0;
        }
    }
    std::vector<DevGrid> lgrids(P.lights.size());
    for (auto& g : lgrids) memset(&g, 0, sizeof g);
    {
        const size_t mark = s.allocations.size();
        const uint64_t bytes_mark = s.info.device_bytes;
        try {
            for (size_t i = 0; i < P.light_grids.size(); ++i) {
                upload_grid(P.light_grids[i]->g, lgrids[i]);
                if (P.light_grids[i]->g.kind != 0) s.ortho_light_grids = true;
            }
        } catch (const GpuError&) {
            drop_allocations_from(mark, bytes_mark);
            for (auto& g : lgrids) memset(&g, 0, sizeof g);
            for (auto& g : P.light_grids) s.info.grid_refs -= g->g.n_refs;
            s.ortho_light_grids = false;
            D.all_lights_gridded = 0u;
            s.info.light_grids = 0;
        }
    }
    D.light_grids = s.upload(lgrids.data(), lgrids.size());
    s.info.upload_seconds = std::chrono::duration<float>(std::chrono::steady_clock::now() - t_up).count();
}

hipEvent_t get_event(const pt_scene& s, size_t i) {
    while (s.events.size() <= i) {
        hipEvent_t e;
        HIP_CHECK(hipEventCreate(&e));
        s.events.push_back(e);
    }
    return s.events[i];
}

#ifndef PT_GRAPH_DEFAULT
// PT_GRAPH=1: frames replayed from a captured hipGraph (render_device).  Measured (MI355X, ROCm 7.2; config 3): a whole
// frame 34.47 -> 34.40 ms, one shard of eight 5.49 -> 5.60 ms - the replay of the ~35 nodes on three streams is no faster
// than their launches (the host is 30 frames ahead of the device either way; the 7 us between dependent kernels stay): off.
#define PT_GRAPH_DEFAULT false
#endif
void render_device(const pt_scene& s, const pt_profile& p, const pt_opts* opts_in, void* d_rgb8, void* d_accum,
                   hipStream_t stream, bool allow_preview = false) {
    pt_opts o;
    normalise_opts(p, opts_in, o);
    if (p.samples == 0) fail(PT_ERR_INVALID, "profile.samples must be > 0");
    if (p.brdf != PT_BRDF_COOK_TORRANCE) fail(PT_ERR_INVALID, "unknown brdf %d", p.brdf);
    if (p.tonemap < 0 || p.tonemap > 2) fail(PT_ERR_INVALID, "unknown tonemap %d", p.tonemap);
    HIP_CHECK(hipSetDevice(s.device));
    TileMap tm = make_tile_map(p, o, o.shard_rank);
    if (tm.n_local == 0) return;
    // escape masks: from the scene's (escape_after + 1)th frame of the default pipeline on (scene_upload has the reason)
    if (s.escape_wanted && !s.escape_tried && !(o.flags & (PT_FLAG_NO_GRIDS | PT_FLAG_MEGAKERNEL))) {
        if (s.frames_rendered >= s.escape_after) escape_masks_build(const_cast<pt_scene&>(s));
        ++s.frames_rendered;
    }
    // the scene as the kernels of THIS frame see it: the KD-tree pipeline (PT_FLAG_NO_GRIDS) - the cross-check of the parity
    // tests - knows no escape masks either
    DevScene dev = s.dev;
    if (o.flags & PT_FLAG_NO_GRIDS) dev.escape = nullptr;

    // tile tables (cached per configuration: no host sync in steady state): the packed offset of every local tile
    // (sharded renders), then - PT_TILE_ORDER=morton - the order in which the wavefront integrator visits the local
    // tiles: along a Z curve over the tile grid instead of row by row
    static const bool morton = [] {
        const char* e = getenv("PT_TILE_ORDER");
        return e && !strcmp(e, "morton");
    }();
    const uint32_t* d_tiles = nullptr;
    uint32_t tile_order_base = 0, tile_k_base = 0;
    if (o.shard_count > 1 || morton) {
        auto key = std::make_tuple(p.width, p.height, o.shard_rank, o.shard_count, o.tile_w, o.tile_h);
        auto found = s.tile_tables.find(key);
        if (found == s.tile_tables.end()) {
            if (s.tile_tables.size() >= 256) {   // (a caller cycling through hundreds of configurations: start over, idle)
                HIP_CHECK(hipDeviceSynchronize());
                s.tile_tables.clear();
            }
            std::vector<uint32_t> table(tm.offsets);
            table.insert(table.end(), tm.tiles.begin(), tm.tiles.end());   // global tile numbers (tile_k_base)
            if (morton) {
                auto spread = [](uint32_t v) {   // bits of v to the even positions
                    uint64_t x = v;
                    x = (x | (x << 16)) & 0x0000ffff0000ffffull;
                    x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
                    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
                    x = (x | (x << 2)) & 0x3333333333333333ull;
                    x = (x | (x << 1)) & 0x5555555555555555ull;
                    return x;
                };
                std::vector<std::pair<uint64_t, uint32_t>> order(tm.n_local_tiles);
                for (uint32_t lt = 0; lt < tm.n_local_tiles; ++lt) {
                    const uint32_t k = tm.tiles[lt];
                    order[lt] = {spread(k % tm.tiles_x) | (spread(k / tm.tiles_x) << 1), lt};
                }
                std::sort(order.begin(), order.end());
                for (auto& e : order) table.push_back(e.second);
            }
            auto buf = std::make_unique<DeviceBuffer>();
            buf->ensure(table.size() * 4);
            HIP_CHECK(hipMemcpy(buf->p, table.data(), table.size() * 4, hipMemcpyHostToDevice));   // (a fresh buffer: nobody reads it yet)
            found = s.tile_tables.emplace(key, std::move(buf)).first;
        }
        d_tiles = (const uint32_t*)found->second->p;
        tile_k_base = (uint32_t)tm.offsets.size();
        if (morton) tile_order_base = (uint32_t)(tm.offsets.size() + tm.tiles.size());
    }
    float* accum = (float*)d_accum;
    if (!accum) {
        s.accum_scratch.ensure(tm.n_local * 12);
        accum = (float*)s.accum_scratch.p;
    }
    const bool timing = o.flags & PT_FLAG_TIMING, counting = o.flags & PT_FLAG_COUNTERS;
#ifdef WF_EXIT_TIMES
    const bool exit_times = true;    // diagnostic build: the stamps are taken by the plain (non-counting) kernels too
#else
    const bool exit_times = false;
#endif
    if (counting || exit_times) {
        s.counter_buf.ensure(sizeof(DevCounters));
        HIP_CHECK(hipMemsetAsync(s.counter_buf.p, 0, sizeof(DevCounters), stream));
#ifdef WF_EXIT_TIMES
        HIP_CHECK(hipMemsetAsync(&((DevCounters*)s.counter_buf.p)->launch_start, 0xff, sizeof(((DevCounters*)nullptr)->launch_start), stream));
#endif
    }

    RenderParams P{};
    P.width = p.width;
    P.height = p.height;
    P.samples = p.samples;
    P.bounces = p.bounces;
    P.tonemap = p.tonemap;
    P.shard_rank = o.shard_rank;
    P.shard_count = o.shard_count;
    P.tile_w = o.tile_w;
    P.tile_h = o.tile_h;
    P.tiles_x = tm.tiles_x;
    P.tiles_y = tm.tiles_y;
    P.n_local = (uint32_t)tm.n_local;
    P.tile_order_base = tile_order_base;
    P.tile_k_base = tile_k_base;
    pt_fastdiv_make((o.tile_w >> 3) * (o.tile_h >> 3), P.div_tile_blocks);
    pt_fastdiv_make(o.tile_w >> 3, P.div_tile_cols);
    pt_fastdiv_make(tm.tiles_x, P.div_tiles_x);

    // ---- integrator selection: wavefront (default, mode 2), or the one-lane-per-pixel megakernel (mode 0)
    const int mode = (o.flags & PT_FLAG_MEGAKERNEL) ? 0 : 2;
    // the 16-bit draw index / bounce fields of the queue records, the per-bounce counter table
    if (p.bounces > 4096u) fail(PT_ERR_INVALID, "profile.bounces %u is out of range (at most 4096)", p.bounces);
    const bool use_cam_grid = !(o.flags & PT_FLAG_NO_GRIDS) && s.dev.cam_grid.res != 0;
    const bool use_light_grids = !(o.flags & PT_FLAG_NO_GRIDS) && s.dev.all_lights_gridded != 0;
    // bounce 0 as ONE kernel (k_wf_shade<GRID >= 2>: camera cast through the camera grid, shadow casts through the light grids).
    // The camera-grid cull belongs to that kernel: its mask is computed, its wavefronts skip empty blocks and k_accumulate
    // adds the background for their pixels under this one condition (round-3 advisory: three places derived it separately).
    const bool bounce0_fused = use_cam_grid && use_light_grids;
    const uint32_t blocks64 = tm.n_local_tiles * (o.tile_w / 8u) * (o.tile_h / 8u);
    // staging budget (radiance 12 B + RNG block 64 B per work item [+ queues]); default 32 GiB of 288 GB
    static const uint64_t budget = [] {
        const char* e = getenv("PT_STAGING_GIB");
        return (uint64_t)((e && *e ? atof(e) : 32.0) * 1024.0 * 1024.0 * 1024.0);
    }();
    static const uint32_t wf_cap = [] {
        const char* e = getenv("PT_WF_CHUNK");
        // the most work items one pass over the bounces may take, whatever the budgets of the frame plan below allow
        // (every extra chunk repeats the ~13 persistent launches and their drain phases: 60.9 ms against 63.0 ms for
        // two chunks of 128 Mi - round 2)
        return (uint32_t)(e && *e ? atof(e) : 320.0 * 1024 * 1024);
    }();
    static const bool wf_overlap = [] {    // shadow(b) on a side stream beside trace(b+1); PT_WF_OVERLAP=0 serialises
        const char* e = getenv("PT_WF_OVERLAP");
        return e && *e ? atoi(e) != 0 : true;
    }();
    static const uint32_t wf_refill = [] {
        // idle lanes that trigger a refill of a persistent wavefront.  Round 3, after the split shade pass and the slack
        // change (config 3, trace stage): 2 / 4 / 6 / 8 / 12 / 16 / 24 / 32 -> 13.36 / 13.31 / 13.32 / 13.31 / 13.41 / 13.57 /
        // 13.98 / 14.63 ms (16 was round 1's optimum)
        const char* e = getenv("PT_WF_REFILL");
        return (uint32_t)(e && *e ? atoi(e) : 8);
    }();
    static const uint32_t wf_walk = [] {
        const char* e = getenv("PT_WF_WALK");
        // 0 = default: 20 for the coherent camera rays (bounce 0 on the KD-tree), 12 for the incoherent rays of the
        // later bounces (7.9 against 8.2 ms per 64 spp, MI355X, config 3)
        return (uint32_t)(e && *e ? atoi(e) : 0);
    }();
    // The shade pass of bounces >= 1 runs beside k_wf_trace_wide (see the launch below).  Measured (MI355X, config 3;
    // profiles/r03_experiments.txt item 6): frame 34.26 -> 34.14 ms, one shard of eight 5.85 -> 5.69 ms - the two kernels
    // slow each other down (beside the 4 workgroups per CU of k_wf_trace_wide a SIMD has registers for one shade wavefront
    // instead of four), so only part of the shorter one is hidden.
    static const bool wf_split = [] {
        const char* e = getenv("PT_WF_SPLIT");
        return e && *e ? atoi(e) != 0 : true;
    }();
    static const bool wf_allwide = [] {   // experiment: every cast of the bounces >= 1 of an opaque scene through k_wf_trace_wide
        const char* e = getenv("PT_WF_ALLWIDE");
        return e && *e ? atoi(e) != 0 : false;
    }();
    // k_wf_trace / k_wf_shadow hand the rays their walker's slack does not cover to k_wf_trace_exact / k_og_shadow_offgrid
    // (csrc/pt_integrator.h, slop model).  PT_WF_EXACT=0 is for A/B measurements of what that costs only: the capped walk.
    // 1: k_wf_trace lists them when it fetches them, k_wf_trace_exact runs behind it (beside the shade pass over the queue);
    // 2: k_wf_shade lists them when it makes the rays, k_wf_trace_exact runs beside k_wf_trace.
    static const uint32_t wf_exact = [] {
        const char* e = getenv("PT_WF_EXACT");
        return (uint32_t)(e && *e ? std::min(2, std::max(0, atoi(e))) : 1);
    }();
    static const uint32_t wf_defer = [] {   // k_wf_trace: age (loop iterations) at which a cast leaves a drained wavefront
        const char* e = getenv("PT_WF_DEFER");
        return (uint32_t)(e && *e ? atoi(e) : 16);
    }();
    // k_wf_shade: workgroups per CU in the grid.  The kernel's loops are grid-stride, but a grid of just the resident
    // workgroups (3 per CU) keeps the whole chip on ONE window of the image at a time - everybody in the ChaCha-bound
    // background together, then everybody waiting for casts into the model together.  Many more workgroups than are
    // resident, each with a short loop, mix the two (and balance the end): bounce-0 kernel of config 3, 3 / 16 / 64 /
    // 256 / 1024 / 4096 workgroups per CU: 20.9 / 17.3 / 16.1 / 15.5 / 15.4 / 16.4 ms; the later bounces (queues of
    // unknown, shrinking length: every extra workgroup is a dispatch that may find nothing) are best at 32
    // (16 / 32 / 64 / 128: frame 37.6 / 36.2 / 36.2 / 36.2 ms, one shard of eight 5.97 / 5.92 / 6.03 / 6.13 ms).
    static const uint32_t shade_blocks_b0 = [] {
        const char* e = getenv("PT_SHADE_BLOCKS_B0");
        return (uint32_t)(e && *e ? atoi(e) : 256);
    }();
    static const uint32_t shade_blocks_later = [] {
        const char* e = getenv("PT_SHADE_BLOCKS");
        return (uint32_t)(e && *e ? atoi(e) : 32);
    }();
    static const uint32_t wf_refill_shadow = [] {
        const char* e = getenv("PT_WF_REFILL_SHADOW");
        return (uint32_t)(e && *e ? atoi(e) : 0);
    }();
    static const uint32_t wf_walk_shadow = [] {
        const char* e = getenv("PT_WF_WALK_SHADOW");
        return (uint32_t)(e && *e ? atoi(e) : 12);
    }();
    if (s.n_cu == 0) {
        HIP_CHECK(hipDeviceGetAttribute(&s.n_cu, hipDeviceAttributeMultiprocessorCount, s.device));
        if (!s.cu_mask.empty()) {
            int bits = 0;
            for (uint32_t w : s.cu_mask) bits += __builtin_popcount(w);
            s.n_cu = std::max(1, std::min(s.n_cu, bits));
        }
    }
    auto side_stream = [&s](hipStream_t* st, int priority) {
        if (!s.cu_mask.empty()) HIP_CHECK(hipExtStreamCreateWithCUMask(st, (uint32_t)s.cu_mask.size(), s.cu_mask.data()));
        else HIP_CHECK(hipStreamCreateWithPriority(st, hipStreamNonBlocking, priority));
    };
    uint32_t batch = o.sample_batch ? o.sample_batch : p.samples;
    const bool alpha = s.dev.has_translucent != 0;
    if (mode >= 1) {
        uint64_t per_sample = tm.n_local * 12u;
        uint64_t max_batch = std::max<uint64_t>(1, budget / std::max<uint64_t>(1, per_sample));
        max_batch = std::min<uint64_t>(max_batch, 0x7fffffffull / ((uint64_t)blocks64 * 64u));
        if (max_batch == 0) fail(PT_ERR_UNSUPPORTED, "image too large for one sample batch");
        batch = (uint32_t)std::min<uint64_t>(batch, max_batch);
        s.staging_buf.ensure((size_t)batch * tm.n_local * 12);
    }
    uint32_t cap = 0;
    // capacities (records) of what is indexed by a queue position: the two path queues, the hit records (+ the alpha walk's draw
    // counts), the shadow records (+ contrib planes, off-grid list), the exact lists
    uint32_t cap_q[2] = {0, 0}, cap_h = 0, cap_s = 0, cap_e = 0;
    std::vector<uint8_t> inline_at(p.bounces + 2, 0);   // bounces >= 1 whose shadow casts run inside the shade kernel
    pt_scene::FrameStats* fs = nullptr;
    bool multi_chunk = false, rng_one_plane = false, skip_dead = false;
    uint32_t stats_slots = 0;
    if (mode == 2) {
        uint64_t items_per_batch = (uint64_t)blocks64 * 64u * batch;
        if (s.trace_blocks == 0) {
            int a = 0, b = 0, c = 0, d = 0;
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_wf_trace<false, false, false>, WF_THREADS, 0));
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_wf_trace<true, false, false>, WF_THREADS, 0));
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, k_wf_shadow<false, false>, WF_THREADS, 0));
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, k_wf_shadow<true, false>, WF_THREADS, 0));
            s.trace_blocks = std::max(1, alpha ? b : a) * s.n_cu;
            s.shadow_blocks = std::max(1, alpha ? d : c) * s.n_cu;
            if (const char* e = getenv("PT_WF_BLOCKS_PER_CU")) {  // experiments: fewer resident workgroups
                int k = atoi(e);
                if (k > 0) {
                    s.trace_blocks = std::min(s.trace_blocks, k * s.n_cu);
                    s.shadow_blocks = std::min(s.shadow_blocks, k * s.n_cu);
                }
            }
        }
        pt_scene::WfPipe& w = s.pipe;
        const uint32_t levels = p.bounces + 3;
        w.ctr.ensure(sizeof(WfCounters) * levels);
        const size_t lights = std::max(1u, s.dev.n_lights);
        static const bool fuse_rng_env = [] {
            const char* e = getenv("PT_OG_FUSE_RNG");
            return e && *e ? atoi(e) != 0 : true;
        }();
        // (the fused bounce-0 kernel keeps words 0-3 of the items' ChaCha blocks in registers: one 16-byte plane per item)
        rng_one_plane = fuse_rng_env && bounce0_fused;
        // ---- The frame plan.  Bytes per work item when nothing is known (every queue as long as the chunk): two path queues of
        // 64 B (+ the entry word), 20 B hit, 64 B shadow record + 16 B per light, the RNG plane(s), draws, off-grid and exact lists.
        const uint64_t per_item = 68u * 2u + 20u + 64u + 16u * lights + (rng_one_plane ? 16u : 32u) + (alpha ? 4u : 0u) + 4u + 16u;
        // (read per frame: the tests change them)
        // First frame 8 GiB: what the CLI pays for in allocation time (path-tracer render of the 500 k-triangle scene, 1080p x 128 spp,
        // render_s with 2 / 4 / 8 / 16 / 64 GiB: 0.121 / 0.095 / 0.080 / 0.136-0.29 / 1.69 s - the frame itself is 0.04-0.05 s).
        // Later frames: what their records need, up to 32 GiB - config 3 takes 14.1 GiB (one pass), the closed room 31.4 GiB in 2
        // passes (170.2 ms in 5 passes of 12.7 GiB, 165.8 in 2, 164.5 in one of 51.7 GiB), the KD-tree pipeline of config 3 29.7 GiB.
        const double first_gib = [] { const char* e = getenv("PT_QUEUE_GIB"); return e && *e ? atof(e) : 8.0; }();
        const double one_pass_gib = [] { const char* e = getenv("PT_QUEUE_ONE_PASS_GIB"); return e && *e ? atof(e) : 32.0; }();
        const double steady_gib = [first_gib] { const char* e = getenv("PT_QUEUE_STEADY_GIB"); return e && *e ? atof(e) : std::max(first_gib, 32.0); }();
        static const bool skip_dead_env = [] { const char* e = getenv("PT_PLAN_SKIP"); return !(e && *e && atoi(e) == 0); }();
        skip_dead = skip_dead_env;
        static const bool inline_auto = [] { const char* e = getenv("PT_OG_INLINE_AUTO"); return !(e && *e && atoi(e) == 0); }();
        const uint32_t max_items = (uint32_t)std::min<uint64_t>(items_per_batch, std::max<uint32_t>(64u, wf_cap & ~63u));
        std::vector<uint64_t> stat_key = {p.width, p.height, p.samples, p.bounces, (uint64_t)p.brdf,
                                          o.flags & (PT_FLAG_NO_GRIDS | PT_FLAG_MEGAKERNEL | PT_FLAG_COUNTERS), (uint64_t)bounce0_fused, o.shard_rank, o.shard_count, o.tile_w, o.tile_h, batch, (uint64_t)max_items};
        static const bool graphs_on = [] {   // (frames replayed from a captured graph keep the first frame's chunking: no plan)
            const char* e = getenv("PT_GRAPH");
            return e && *e ? atoi(e) != 0 : PT_GRAPH_DEFAULT;
        }();
        // the chunk of the first frame: what fits the first-frame budget
        uint32_t cap_a = (uint32_t)std::min<uint64_t>(max_items, std::max<uint64_t>(1u << 20, (uint64_t)(first_gib * 1073741824.0) / per_item) & ~63ull);
        if (s.wf_cap_ok) cap_a = std::min(cap_a, s.wf_cap_ok);
        // counts per (batch, chunk of fs->cap_items items) -> the plan: m consecutive chunks become one, every buffer as long as
        // the largest group's counts need; the largest m whose buffers fit the steady budget
        auto make_plan = [&](pt_scene::FrameStats& f) {
            const uint32_t ca = f.cap_items, lv = f.levels;
            std::vector<std::vector<uint32_t>> batches;   // slots of each batch, in order
            for (uint32_t k = 0; k < f.n_slots; ++k) {
                if (f.first_item_of_slot[k] == 0u) batches.emplace_back();
                if (batches.empty()) return false;
                batches.back().push_back(k);
            }
            if (batches.empty() || ca == 0) return false;
            uint32_t m_max = 1;
            for (auto& bt : batches) m_max = std::max<uint32_t>(m_max, (uint32_t)bt.size());
            // (PT_WF_CHUNK caps a chunk; a batch that may be one chunk needs no multiple of ca)
            m_max = std::min<uint32_t>(m_max, (uint64_t)max_items >= items_per_batch ? (uint32_t)((items_per_batch + ca - 1u) / ca) : std::max(1u, max_items / ca));
            // by the number of passes over a batch: m = the fewest first-frame chunks per pass that make that many passes, so the
            // passes come out even (nine chunks in three passes: 3 + 3 + 3, not 4 + 4 + 1 at a third more memory)
            uint32_t s_max = 1;
            for (auto& bt : batches) s_max = std::max<uint32_t>(s_max, (uint32_t)bt.size());
            uint32_t m_prev = 0;
            for (uint32_t n_pass = 1; n_pass <= s_max; ++n_pass) {
                const uint32_t m = (s_max + n_pass - 1u) / n_pass;
                if (m > m_max || m == m_prev) continue;
                m_prev = m;
                uint64_t q1 = 0, q0 = 0, hh = 0, ss = 0, ee = 0;
                std::vector<uint64_t> tot_q(lv, 0), tot_s(lv, 0);
                std::vector<uint32_t> last;
                for (auto& bt : batches)
                    for (size_t g0 = 0; g0 < bt.size(); g0 += m) {
                        last.push_back(0u);
                        for (uint32_t b = 0; b < lv; ++b) {
                            uint64_t nq = 0, ns = 0, ne = 0;
                            for (size_t k = g0; k < std::min(bt.size(), g0 + m); ++k) {
                                const uint32_t* row = f.host + ((size_t)bt[k] * lv + b) * 4;
                                nq += row[0];
                                ns += row[1];
                                ne += row[2];
                            }
                            tot_q[b] += nq;
                            tot_s[b] += ns;
                            if (b >= 1 && nq != 0) last.back() = b;
                            if (b >= 1) {   // (queue b lives in queue[b & 1]; the hits of its casts by queue position)
                                uint64_t& q = (b & 1u) ? q1 : q0;
                                q = std::max(q, nq);
                                hh = std::max(hh, nq);
                            }
                            ss = std::max(ss, ns);
                            ee = std::max(ee, ne);
                        }
                    }
                const uint64_t items = std::min<uint64_t>((uint64_t)ca * m, items_per_batch);
                if (!bounce0_fused) hh = std::max(hh, items);   // the casts of bounce 0 too: hits and draw counts by work item
                auto pad = [](uint64_t n) { return (std::max<uint64_t>(n, 1024) + 1023) & ~1023ull; };
                q0 = pad(q0), q1 = pad(q1), hh = pad(hh), ss = pad(ss), ee = pad(ee);
                const uint64_t bytes = 68u * (q0 + q1) + 20u * hh + (alpha ? 4u * hh : 0u) + (64u + 16u * lights + 4u) * ss + 16u * ee +
                                       (rng_one_plane ? 16u : (wf_overlap && items < items_per_batch) ? 64u : 32u) * items;   // (two copies: the next chunk's are made ahead)
                // (a batch that fits `one_pass_gib` as ONE chunk takes it: every extra pass repeats the persistent launches and their
                // drains - the KD-tree pipeline of config 3, 29.7 GiB in one pass: 62.4 ms, in three passes of 16 GiB 75.3 ms)
                const bool fits = bytes <= (uint64_t)(steady_gib * 1073741824.0) ||
                                  (items >= items_per_batch && bytes <= (uint64_t)(one_pass_gib * 1073741824.0));
                if ((fits || m == 1) && std::max({q0, q1, hh, ss, ee, items}) < 0xffffffffull) {
                    f.plan_cap = (uint32_t)items;
                    f.plan_q[0] = (uint32_t)q0;
                    f.plan_q[1] = (uint32_t)q1;
                    f.plan_h = (uint32_t)hh;
                    f.plan_s = (uint32_t)ss;
                    f.plan_e = (uint32_t)ee;
                    // shadow casts inside the shade kernel where (nearly) every ray of a bounce reaches a lit surface
                    // (PT_OG_INLINE_ALL=1 in the closed room: +6.6 %; in an open scene: -6 %)
                    f.plan_inline.assign(p.bounces + 2, 0);
                    if (inline_auto && use_light_grids)
                        for (uint32_t b = 1; b < lv && b < f.plan_inline.size(); ++b) f.plan_inline[b] = tot_q[b] > 0 && tot_s[b] * 10 >= tot_q[b] * 6;
                    f.plan_last = last;
                    f.plan_fresh = true;
                    if (const char* e = getenv("PT_PLAN_DEBUG"); e && *e && atoi(e)) {
                        fprintf(stderr, "[ptgpu] frame plan: %u chunks of %u -> 1 of %llu items; queues %llu / %llu, hits %llu, shadow %llu, exact %llu records; %.3f GiB\n",
                                m, ca, (unsigned long long)items, (unsigned long long)q0, (unsigned long long)q1, (unsigned long long)hh,
                                (unsigned long long)ss, (unsigned long long)ee, bytes / 1073741824.0);
                        for (uint32_t b = 0; b < lv; ++b)
                            fprintf(stderr, "[ptgpu]   bounce %u: %llu rays, %llu shadow records%s\n", b, (unsigned long long)tot_q[b],
                                    (unsigned long long)tot_s[b], b < f.plan_inline.size() && f.plan_inline[b] ? " (inline)" : "");
                    }
                    return true;
                }
            }
            return false;
        };
        if (!graphs_on) {
            auto& slot = s.frame_stats[stat_key];
            if (!slot) {
                if (s.frame_stats.size() > 64) {   // (a caller cycling through configurations: start over)
                    HIP_CHECK(hipDeviceSynchronize());
                    s.frame_stats.clear();
                }
                s.frame_stats[stat_key] = std::make_unique<pt_scene::FrameStats>();
            }
            fs = s.frame_stats[stat_key].get();
            if (fs->pending) {
                const hipError_t q = hipEventQuery(fs->done);
                (void)hipGetLastError();   // (hipErrorNotReady is no error)
                if (q == hipSuccess) {
                    fs->pending = false;
                    bool overflow = false;
                    for (uint32_t k = 0; k < fs->n_slots; ++k) {
                        overflow = overflow || fs->host[(size_t)k * fs->levels * 4 + 3] != 0u;
                        if (fs->planned && fs->stats_planned && k < fs->plan_last.size())   // (a ray at a bounce the plan did not launch)
                            for (uint32_t b = fs->plan_last[k] + 1u; b < fs->levels; ++b)
                                overflow = overflow || fs->host[((size_t)k * fs->levels + b) * 4] != 0u;
                    }
                    if (overflow) {
                        // cannot happen (the counts of a configuration do not change): a queue sized from them ran full
                        fs->valid = fs->planned = false;
                        fail(PT_ERR_DEVICE, "internal error: a path queue sized from an earlier frame's counts ran full; the previous frame of "
                                            "this configuration is not to be trusted");
                    }
                    if (!fs->planned) {
                        fs->valid = true;
                        fs->planned = make_plan(*fs);
                    }
                }
            }
        }
        bool exact = fs && fs->planned && !fs->plan_failed;
        if (exact) {
            cap = fs->plan_cap;
            cap_q[0] = fs->plan_q[0], cap_q[1] = fs->plan_q[1], cap_h = fs->plan_h, cap_s = fs->plan_s, cap_e = fs->plan_e;
            inline_at = fs->plan_inline;
            inline_at.resize(p.bounces + 2, 0);
            // test hook (tests/test_frame_plan.py): a plan that is WRONG - every array shorter than its records - so that what
            // cannot happen does: the kernels must drop the records that do not fit without writing past an array, flag the
            // frame, and the next call of the configuration must say so
            if (const char* e = getenv("PT_PLAN_TEST_SHRINK"); e && *e) {
                const double f = std::min(1.0, std::max(0.01, atof(e)));
                for (uint32_t* c : {&cap_q[0], &cap_q[1], &cap_s, &cap_e}) *c = std::max<uint32_t>(64u, (uint32_t)(*c * f));
                cap_h = std::max({(uint32_t)(cap_h * f), cap_q[0], cap_q[1]});
            }
        }
        if (!exact) {
            cap = cap_a;
            cap_q[0] = cap_q[1] = cap_h = cap_s = cap_e = cap;
        }
        // allocation; a device that cannot provide the buffers (shared GPU) gets half-size chunks with every queue as long as
        // the chunk, and so on, down to 1 Mi items
        const bool give_back = exact && fs->plan_fresh;   // (a new plan gives memory back, once)
        bool grew = false;
        auto fit = [give_back, &grew](DeviceBuffer& b, size_t n) {
            if (give_back && b.bytes > n + n / 4 + (64u << 20)) b.release();
            grew = grew || b.bytes < n;
            return b.try_ensure(n);
        };
        // (what the queues must leave free: the runtime allocates the kernels' scratch - up to 592 B per lane of every wave slot
        // of the device, per hardware queue: ~1.2 GB - when they are first launched, and dies if it cannot)
        const double reserve_gib = [] { const char* e = getenv("PT_QUEUE_RESERVE_GIB"); return e && *e ? atof(e) : 2.0; }();
        while (true) {
            multi_chunk = (uint64_t)cap < items_per_batch;
            const bool two_rng = !rng_one_plane && multi_chunk && wf_overlap;
            bool ok = fit(w.queue[0], (size_t)cap_q[0] * 68u) && fit(w.queue[1], (size_t)cap_q[1] * 68u) &&   // (64 B + the entry word)
                      fit(w.hits, (size_t)cap_h * 20u) && fit(w.shadow, (size_t)cap_s * 64u) &&
                      fit(w.contrib, (size_t)cap_s * 16u * lights) && fit(w.rng[0], (size_t)cap * (rng_one_plane ? 16u : 32u)) &&
                      (!two_rng || fit(w.rng[1], (size_t)cap * 32u)) &&
                      (!alpha || fit(w.draws, (size_t)cap_h * 4u)) &&   // RNG draw index of the alpha walk
                      fit(w.offgrid, (size_t)cap_s * 4u) &&   // shadow jobs left to k_og_shadow_offgrid (long normals; rays the wavefront walker does not take)
                      fit(w.exact[0], (size_t)cap_e * 8u) && fit(w.exact[1], (size_t)cap_e * 8u) &&   // casts left to k_wf_trace_exact: queue index + hit word
                      (!bounce0_fused || w.block_mask.try_ensure((size_t)blocks64 * 4u + 4u + tm.n_local)) &&
                      // casts left to k_wf_trace_wide: at most one per lane in flight when the queue runs dry
                      // (4 B the queue index + 20 B a hit + 4 B the progress of the walk: wf_list_* in pt_wavefront.h)
                      (!wf_defer ||
                       w.deferred.try_ensure((wf_allwide && !alpha ? (size_t)std::max(cap_q[0], cap_q[1]) : (size_t)s.trace_blocks * WF_THREADS) * 4u *
                                             (alpha ? WF_LIST_WORDS_ALPHA : WF_LIST_WORDS_OPAQUE)));
            if (ok && grew && cap > (1u << 20)) {
                size_t free_b = 0, total_b = 0;
                if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)free_b < reserve_gib * 1073741824.0) ok = false;
                grew = false;
            }
            if (ok) {
                if (exact) fs->plan_fresh = false;
                break;
            }
            for (DeviceBuffer* b : {&w.queue[0], &w.queue[1], &w.hits, &w.shadow, &w.contrib, &w.rng[0], &w.rng[1], &w.draws, &w.offgrid, &w.deferred, &w.exact[0], &w.exact[1], &w.block_mask})
                b->release();
            if (exact) {   // (no room for the planned sizes: as a first frame)
                exact = false;
                fs->plan_failed = true;
                cap = cap_a;
            } else {
                if (cap <= (1u << 20))
                    fail(PT_ERR_DEVICE, "out of device memory: the path queues need %zu bytes for %u work items", (size_t)cap * per_item, cap);
                cap = std::max<uint32_t>(1u << 20, (cap / 2u) & ~63u);
                s.wf_cap_ok = cap;
            }
            cap_q[0] = cap_q[1] = cap_h = cap_s = cap_e = cap;
            std::fill(inline_at.begin(), inline_at.end(), 0);
        }
        s.queue_bytes_last = w.queue[0].bytes + w.queue[1].bytes + w.hits.bytes + w.shadow.bytes + w.contrib.bytes + w.rng[0].bytes +
                             w.rng[1].bytes + w.draws.bytes + w.offgrid.bytes + w.exact[0].bytes + w.exact[1].bytes + w.deferred.bytes;
        s.queue_chunk_last = cap;
        s.frame_planned_last = exact ? 1u : 0u;
        if (fs) {
            // this frame's counts, one line per (batch, chunk): taken every frame (a few KB) - the first frame's feed the plan,
            // the later ones only say whether a queue ran full
            stats_slots = 0;
            for (uint32_t s0 = 0; s0 < p.samples; s0 += batch) {
                const uint64_t tot = (uint64_t)blocks64 * 64u * std::min(batch, p.samples - s0);
                stats_slots += (uint32_t)((tot + cap - 1u) / cap);
            }
            if (!fs->pending) {
                if (fs->n_slots != stats_slots || fs->levels != levels || !fs->host) {
                    if (fs->host) (void)hipHostFree(fs->host);
                    fs->host = nullptr;
                    HIP_CHECK(hipHostMalloc((void**)&fs->host, (size_t)stats_slots * levels * 16u));
                    fs->n_slots = stats_slots;
                    fs->levels = levels;
                }
                if (!fs->done) HIP_CHECK(hipEventCreateWithFlags(&fs->done, hipEventDisableTiming));
                fs->cap_items = cap;
                fs->stats_planned = exact;
                fs->first_item_of_slot.assign(stats_slots, 0u);
                s.stats_dev.ensure((size_t)stats_slots * levels * 16u);
            } else {
                stats_slots = 0;   // (an earlier frame's line is still on its way)
            }
        }
        // (the next chunk's RNG planes are produced on the SHADOW stream, idle while a chunk's bounce 0 is traced - not on a stream
        // of their own: a FIFTH stream - the caller's, shadow, wide, exact and that one - makes two of them share one of the
        // device's four hardware queues (ROCm's default), and every launch of the frame then waits ~45 us longer for its turn:
        // config 3, 40.3 -> 41.8 ms a frame with the stream merely existing, config 5 465 -> 508 ms)
        if (multi_chunk && wf_overlap && !rng_one_plane && !w.ev_rng) {
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_rng, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_chunk, hipEventDisableTiming));
        }
        if (wf_overlap && !w.side) {
            side_stream(&w.side, 0);
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_shade, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_shadow, hipEventDisableTiming));
        }
        if (wf_overlap && !w.side_wide) {
            // (high priority: their few workgroups - the long casts of the drain, the casts the wavefront walker does not take -
            // run beside the shade pass over the queue, whose thousands of short workgroups would otherwise take every slot
            // that comes free before them; PT_WF_SIDE_PRIORITY=0: default priority)
            int prio_lo = 0, prio_hi = 0;
            HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
            const char* pe = getenv("PT_WF_SIDE_PRIORITY");
            const int prio = (pe && *pe && atoi(pe) == 0) ? prio_lo : prio_hi;
            side_stream(&w.side_wide, prio);
            side_stream(&w.side_exact, prio);
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_trace, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_wide, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_exact, hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&w.ev_exact_go, hipEventDisableTiming));
        }
    }
    uint32_t blocks = tm.n_local_tiles * (o.tile_w * o.tile_h / 256u);
    size_t ev = 0;
    uint32_t launches = 0, stage_launches = 0;
    // (stage id, first event index) of every timed launch: 0 generate 1 trace 2 shade 3 shadow 4 accumulate 5 fused
    std::vector<std::pair<int, size_t>> marks;
    std::vector<size_t> fused_marks;   // first event index of every fused bounce-0 launch (k_wf_shade<GRID >= 2>)
    hipStream_t stage_stream = stream;
    auto stage_begin = [&](int stage) {
        if (!timing) return;
        marks.emplace_back(stage, ev);
        HIP_CHECK(hipEventRecord(get_event(s, ev++), stage_stream));
    };
    auto stage_end = [&]() {
        ++stage_launches;
        if (timing) HIP_CHECK(hipEventRecord(get_event(s, ev++), stage_stream));
    };
    DevCounters* gctr = (counting || exit_times) ? (DevCounters*)s.counter_buf.p : nullptr;
    // ---- hipGraph: a plain frame (no timing events, counters, callbacks) is captured once per configuration - on a
    // stream of the scene's own, the caller's may be the null stream - and replayed on the caller's stream afterwards.
    // Everything the launches below depend on is in the key; the side streams join the capture through the events they
    // wait for and are joined again before it ends.
    static const bool wf_graph = [] {
        const char* e = getenv("PT_GRAPH");
        return e && *e ? atoi(e) != 0 : PT_GRAPH_DEFAULT;
    }();
    const hipStream_t user_stream = stream;
    struct CaptureGuard {   // (an exception between begin and end must not leave the stream capturing)
        hipStream_t st = nullptr;
        ~CaptureGuard() {
            if (!st) return;
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(st, &g);
            if (g) (void)hipGraphDestroy(g);
            (void)hipGetLastError();
        }
    } capture;
    std::vector<uint64_t> graph_key;
    if (wf_graph && mode == 2 && !timing && !counting && !exit_times && !o.progress && !(allow_preview && o.preview)) {
        const pt_scene::WfPipe& w = s.pipe;
        graph_key = {p.width, p.height, p.samples, p.bounces, (uint64_t)p.tonemap, o.flags, o.shard_rank, o.shard_count, o.tile_w,
                     o.tile_h, o.sample_batch, (uint64_t)d_rgb8, (uint64_t)accum, (uint64_t)d_tiles, cap, cap_q[0], cap_q[1], cap_h, cap_s, cap_e, batch,
                     (uint64_t)s.staging_buf.p, (uint64_t)w.queue[0].p, (uint64_t)w.queue[1].p, (uint64_t)w.hits.p,
                     (uint64_t)w.shadow.p, (uint64_t)w.contrib.p, (uint64_t)w.ctr.p, (uint64_t)w.rng[0].p, (uint64_t)w.rng[1].p,
                     (uint64_t)w.draws.p, (uint64_t)w.offgrid.p, (uint64_t)w.deferred.p, (uint64_t)w.exact[0].p, (uint64_t)w.exact[1].p, (uint64_t)s.trace_blocks,
                     (uint64_t)s.shadow_blocks, (uint64_t)tm.n_local, (uint64_t)tm.n_local_tiles, (uint64_t)w.block_mask.p};
        auto hit = s.graphs.find(graph_key);
        if (hit != s.graphs.end()) {
            HIP_CHECK(hipGraphLaunch(hit->second, user_stream));
            return;
        }
        if (!s.capture_stream) HIP_CHECK(hipStreamCreateWithFlags(&s.capture_stream, hipStreamNonBlocking));
        stream = s.capture_stream;
        stage_stream = stream;
        HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
        capture.st = stream;
    }
    // bounce 0 through the camera grid: the 8x8 pixel blocks no camera ray can hit anything in (k_cam_block_mask) are
    // found once per frame; their wavefronts write the background without a ChaCha block or a cast.  PT_CAM_CULL=0: off
    static const bool cam_cull = [] {
        const char* e = getenv("PT_CAM_CULL");
        return e && *e ? atoi(e) != 0 : true;
    }();
    const uint32_t* block_empty = nullptr;
    if (cam_cull && mode == 2 && bounce0_fused && !counting) {
        RenderParams P1 = P;
        P1.sample_begin = 0;
        P1.sample_end = 1;
        pt_fastdiv_make(1u, P1.div_batch);
        HIP_CHECK(hipMemsetAsync((uint32_t*)s.pipe.block_mask.p + blocks64, 0, 4, stream));
        hipLaunchKernelGGL(k_cam_block_mask, dim3((blocks64 * 64u + 255u) / 256u), dim3(256), 0, stream, dev, P1, d_tiles, blocks64,
                           (uint32_t*)s.pipe.block_mask.p, (uint8_t*)((uint32_t*)s.pipe.block_mask.p + blocks64 + 1u));
        HIP_CHECK(hipGetLastError());
        block_empty = (const uint32_t*)s.pipe.block_mask.p;
    }
    s.last_mask_blocks = block_empty ? blocks64 : 0u;
    uint32_t stats_line = 0, chunk_slot = 0;
    const std::vector<uint32_t>* plan_last = (fs && s.frame_planned_last && skip_dead) ? &fs->plan_last : nullptr;
    for (uint32_t s0 = 0; s0 < p.samples; s0 += batch) {
        P.sample_begin = s0;
        P.sample_end = std::min(p.samples, s0 + batch);
        uint32_t nb = P.sample_end - P.sample_begin;
        pt_fastdiv_make(nb, P.div_batch);
        if (mode == 0) {
            stage_begin(5);
            if (counting)
                hipLaunchKernelGGL(k_render<true>, dim3(blocks), dim3(256), 0, stream, dev, P, d_tiles, accum, gctr);
            else
                hipLaunchKernelGGL(k_render<false>, dim3(blocks), dim3(256), 0, stream, dev, P, d_tiles, accum, gctr);
            HIP_CHECK(hipGetLastError());
            stage_end();
            ++launches;
        } else {
            uint32_t total_items = blocks64 * 64u * nb;
            const pt_scene::WfPipe& pipe = s.pipe;
            hipStream_t st_main = stream;
            hipStream_t st_shadow = wf_overlap ? pipe.side : st_main;   // (not const: see the shadow stage)
            // opaque scenes, several chunks: the RNG planes of chunk c+1 are produced on their own stream while
            // chunk c runs its bounces (k_wf_rng is pure integer ALU work; the traversal kernels leave ~40 % of
            // the issue slots idle and end in a drain phase)
            // both kinds of grid: the bounce-0 kernel computes the ChaCha block itself (GRID 3)
            const bool fused_rng = rng_one_plane;   // (PT_OG_FUSE_RNG=0: off)
            const bool rng_ahead = !fused_rng && wf_overlap && total_items > cap && pipe.side != nullptr && pipe.ev_rng != nullptr;
            uint32_t chunk_no = 0;
            for (uint32_t base = 0; base < total_items; base += cap, ++chunk_no) {
                WfParams W{};
                W.P = P;
                W.item_base = base;
                W.n_items = std::min(cap, total_items - base);
                W.cap = cap;
                W.hcap = cap_h;
                W.scap = cap_s;
                W.ecap = cap_e;
                W.qcap_in = cap_q[0];
                W.qcap_out = cap_q[1];
                W.rng_first_plane = rng_one_plane ? 1u : 0u;
                W.n_mask_blocks = blocks64;
                uint4* rng_planes = (uint4*)pipe.rng[rng_ahead ? (chunk_no & 1u) : 0u].p;
                // coherence sorting of the survivors by direction octant: measured (MI355X, config 3) trace of bounce 1
                // 9.50 -> 9.25 ms, but the bounce-0 kernel 19.0 -> 20.2 ms: off by default
                static const uint32_t wf_sort = [] {
                    const char* e = getenv("PT_WF_SORT");
                    return (uint32_t)(e && *e ? atoi(e) : 0);
                }();
                W.sort_octants = wf_sort;
                // PT_WF_ENTRY=1: casts of bounces >= 1 start at the home node of the primitive their ray leaves (trav_enter).
                // Measured (profiles/r03_experiments.txt item 2): 31 % fewer node visits, the same time - off by default
                static const uint32_t wf_entry = [] {
                    const char* e = getenv("PT_WF_ENTRY");
                    return (uint32_t)(e && *e ? atoi(e) != 0 : 0);
                }();
                W.use_entry = wf_entry;
                W.exact_handover = wf_exact ? 1u : 0u;   // (per bounce below: 1 k_wf_trace lists, 2 k_wf_shade listed)
                W.exact_shade_lists = wf_exact == 2u ? 1u : 0u;

                W.refill_min = std::max(1u, std::min(64u, wf_refill));
                W.walk_steps = wf_walk ? wf_walk : 20u;
                WfCounters* wctr = (WfCounters*)pipe.ctr.p;
                HIP_CHECK(hipMemsetAsync(wctr, 0, sizeof(WfCounters) * (p.bounces + 3), st_main));
                // bounce 0 derives the camera rays in place from the staged screen positions (no queue[0])
                const bool fused_primary = true;
                if (fused_rng) {
                    // (no k_wf_rng launch)
                } else if (!rng_ahead || chunk_no == 0) {
                    stage_begin(0);
                    hipLaunchKernelGGL(k_wf_rng, dim3((W.n_items + 255u) / 256u), dim3(256), 0, st_main, dev, W, d_tiles,
                                       rng_planes);
                    HIP_CHECK(hipGetLastError());
                    stage_end();
                } else {
                    HIP_CHECK(hipStreamWaitEvent(st_main, pipe.ev_rng, 0));  // produced underneath the previous chunk
                }
                if (rng_ahead && base + cap < total_items) {
                    // the other copy was last read by chunk c-1, which has completed on st_main by now
                    WfParams Wn = W;
                    Wn.item_base = base + cap;
                    Wn.n_items = std::min(cap, total_items - Wn.item_base);
                    HIP_CHECK(hipEventRecord(pipe.ev_chunk, st_main));
                    HIP_CHECK(hipStreamWaitEvent(pipe.side, pipe.ev_chunk, 0));
                    stage_stream = pipe.side;
                    stage_begin(0);
                    hipLaunchKernelGGL(k_wf_rng, dim3((Wn.n_items + 255u) / 256u), dim3(256), 0, pipe.side, dev, Wn,
                                       d_tiles, (uint4*)pipe.rng[(chunk_no + 1u) & 1u].p);
                    HIP_CHECK(hipGetLastError());
                    stage_end();
                    stage_stream = st_main;
                    HIP_CHECK(hipEventRecord(pipe.ev_rng, pipe.side));
                }
                // (the plan knows where this chunk's last ray ends: the launches of the bounces behind it - ~85 us each for nothing -
                // are not made; PT_PLAN_SKIP=0: all of them)
                const uint32_t b_end = (plan_last && chunk_slot < plan_last->size()) ? std::min(p.bounces, (*plan_last)[chunk_slot]) : p.bounces;
                ++chunk_slot;
                for (uint32_t b = 0; b <= b_end; ++b) {
                    W.bounce = b;
                    W.qcap_in = cap_q[b & 1u];          // (queue b lives in pipe.queue[b & 1])
                    W.qcap_out = cap_q[(b + 1u) & 1u];
                    // (node steps per walking phase: 12 until the escape masks took the short casts out of the queues; re-swept on the rays that
                    // are left - 12 / 14 / 16 / 18: config 3 40.3-40.8 / 40.3-40.4 / 39.7-39.8 / 39.8-40.0 ms, closed room 173.9 -> 171.4)
                    W.walk_steps = wf_walk ? wf_walk : (b == 0 ? 20u : 16u);
                    float4* q_in = (float4*)pipe.queue[b & 1].p;
                    float4* q_out = (float4*)pipe.queue[(b + 1) & 1].p;
                    const bool prim = fused_primary && b == 0;
                    // origin grids (pt_grid.h): 2 = camera cast + shadow casts inside the shade kernel (bounce 0, both
                    // kinds of grid), 1 = shadow casts inside the shade kernel, 0 = none
                    // (bounces >= 1 keep the shade kernel lean and cast their shadow rays in k_og_shadow: measured faster)
                    static const int inline_later = [] {
                        const char* e = getenv("PT_OG_INLINE_ALL");
                        return e && *e ? atoi(e) : 0;
                    }();
                    // (... or where the frame plan found (nearly) every ray of the bounce reaching a lit surface)
                    const int grid_mode = (prim && bounce0_fused) ? (fused_rng ? 3 : 2)
                                                                  : (use_light_grids && (inline_later || (b < inline_at.size() && inline_at[b])) ? 1 : 0);
#define PT_LAUNCH_ACP(kernel, grid, threads, ...)                                                                                      \
    do {                                                                                                                               \
        if (prim && alpha && counting)                                                                                                  \
            hipLaunchKernelGGL((kernel<true, true, true>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__);                        \
        else if (prim && alpha) hipLaunchKernelGGL((kernel<true, false, true>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__);   \
        else if (prim && counting) hipLaunchKernelGGL((kernel<false, true, true>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__); \
        else if (prim) hipLaunchKernelGGL((kernel<false, false, true>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__);           \
        else if (alpha && counting) hipLaunchKernelGGL((kernel<true, true, false>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__); \
        else if (alpha) hipLaunchKernelGGL((kernel<true, false, false>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__);          \
        else if (counting) hipLaunchKernelGGL((kernel<false, true, false>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__);       \
        else hipLaunchKernelGGL((kernel<false, false, false>), dim3(grid), dim3(threads), 0, st_main, __VA_ARGS__);                    \
        HIP_CHECK(hipGetLastError());                                                                                                  \
    } while (0)
#define PT_LAUNCH_AC(kernel, grid, ...)                                                                                  \
    do {                                                                                                                 \
        if (alpha && counting) hipLaunchKernelGGL((kernel<true, true>), dim3(grid), dim3(WF_THREADS), 0, st_shadow, __VA_ARGS__); \
        else if (alpha) hipLaunchKernelGGL((kernel<true, false>), dim3(grid), dim3(WF_THREADS), 0, st_shadow, __VA_ARGS__);       \
        else if (counting) hipLaunchKernelGGL((kernel<false, true>), dim3(grid), dim3(WF_THREADS), 0, st_shadow, __VA_ARGS__);    \
        else hipLaunchKernelGGL((kernel<false, false>), dim3(grid), dim3(WF_THREADS), 0, st_shadow, __VA_ARGS__);                 \
        HIP_CHECK(hipGetLastError());                                                                                    \
    } while (0)
// k_wf_shade<ALPHA, COUNT, PRIMARY, GRID>
#define PT_SHADE_ARGS                                                                                                         \
    dev, W, d_tiles, (const float4*)q_in, shade_hits, (const uint4*)rng_planes, (const uint32_t*)pipe.draws.p, \
        q_out, (float4*)pipe.shadow.p, (float4*)pipe.contrib.p, (float*)s.staging_buf.p, shade_list,                 \
        (const uint32_t*)pipe.exact[b & 1].p, (const uint4*)pipe.hits.p, (uint32_t*)pipe.exact[(b + 1) & 1].p,      \
        (grid_mode >= 2 ? block_empty : (const uint32_t*)nullptr), wctr, gctr
#define PT_LAUNCH_SHADE(A, C, P, G)                                                                                    \
    hipLaunchKernelGGL((k_wf_shade<A, C, P, G>), dim3(shade_grid), dim3(WF_SHADE_THREADS), 0, st_main, PT_SHADE_ARGS)
#define PT_LAUNCH_SHADE_G(G)                                                   \
    do {                                                                       \
        if (prim && alpha && counting) PT_LAUNCH_SHADE(true, true, true, G);   \
        else if (prim && alpha) PT_LAUNCH_SHADE(true, false, true, G);         \
        else if (prim && counting) PT_LAUNCH_SHADE(false, true, true, G);      \
        else if (prim) PT_LAUNCH_SHADE(false, false, true, G);                 \
        else if (alpha && counting) PT_LAUNCH_SHADE(true, true, false, G);     \
        else if (alpha) PT_LAUNCH_SHADE(true, false, false, G);                \
        else if (counting) PT_LAUNCH_SHADE(false, true, false, G);             \
        else PT_LAUNCH_SHADE(false, false, false, G);                          \
    } while (0)
#define PT_LAUNCH_SHADE_P(G)                                            \
    do {                                                                \
        if (alpha && counting) PT_LAUNCH_SHADE(true, true, true, G);    \
        else if (alpha) PT_LAUNCH_SHADE(true, false, true, G);          \
        else if (counting) PT_LAUNCH_SHADE(false, true, true, G);       \
        else PT_LAUNCH_SHADE(false, false, true, G);                    \
    } while (0)
                    bool split_shade = false;
                    if (grid_mode < 2) {   // (grid_mode >= 2: k_wf_shade casts the camera rays itself)
                        stage_begin(1);
                        if (prim && use_cam_grid) {   // camera rays: one grid lookup instead of a KD walk (pt_grid_kernels.h)
                            const dim3 g((W.n_items + 255u) / 256u);
#define PT_LAUNCH_OGP(A, C)                                                                                                   \
    hipLaunchKernelGGL((k_og_primary<A, C>), g, dim3(256), 0, st_main, dev, W, d_tiles, (uint4*)pipe.hits.p, \
                       (const uint4*)rng_planes, (uint32_t*)pipe.draws.p, gctr)
                            if (alpha && counting) PT_LAUNCH_OGP(true, true);
                            else if (alpha) PT_LAUNCH_OGP(true, false);
                            else if (counting) PT_LAUNCH_OGP(false, true);
                            else PT_LAUNCH_OGP(false, false);
#undef PT_LAUNCH_OGP
                            HIP_CHECK(hipGetLastError());
                        } else {
                            W.defer_age = prim ? 0u : wf_defer;
                            // the hand-over list: queue indices, then (split shade pass) the plane of their hits
                            const uint32_t list_cap = (uint32_t)s.trace_blocks * WF_THREADS;
                            uint4* list_hits = (uint4*)((uint32_t*)pipe.deferred.p + list_cap);
                            split_shade = wf_split && W.defer_age != 0u && pipe.side_wide != nullptr;
                            W.list_cap = list_cap;
                            W.split_deferred = split_shade ? 1u : 0u;
                            const bool allwide = wf_allwide && !alpha && W.defer_age != 0u;
                            if (allwide) {
                                split_shade = false;
                                W.list_cap = std::max(cap_q[0], cap_q[1]);
                                W.split_deferred = 0u;
                            }
                            // The casts the wavefront walker does not take (slack_is_capped): one lane each on the grown-box walker.
                            // Bounces >= 1: k_wf_shade listed them when it made the rays, so the launch goes out BEFORE the
                            // persistent kernel, on a stream of its own, and runs beside it (its few hundred workgroups take
                            // their slots first; the persistent grid's last workgroups start as those free up).  Camera rays of
                            // the KD-tree pipeline: k_wf_trace lists them, the launch follows it.
                            W.exact_handover = (wf_exact && !allwide) ? (prim ? 1u : wf_exact) : 0u;
                            // (beside the persistent kernel: a SMALL grid - every workgroup of it takes a slot from k_wf_trace for as
                            // long as it runs, and 2048 workgroups held most of the chip for milliseconds: closed room 185 -> 199 ms.
                            // PT_WF_EXACT_BLOCKS: workgroups per 8 CUs)
                            static const uint32_t exact_blocks = [] {
                                const char* e = getenv("PT_WF_EXACT_BLOCKS");
                                return (uint32_t)(e && *e ? atoi(e) : 4);
                            }();
                            auto launch_exact = [&](hipStream_t st_exact) {
                                const dim3 eg(W.exact_handover == 2u ? std::max(1u, (uint32_t)s.n_cu * exact_blocks / 8u) : (uint32_t)s.n_cu * 16u);
#define PT_LAUNCH_EXACT(C, A, P)                                                                                               \
    hipLaunchKernelGGL((k_wf_trace_exact<C, A, P>), eg, dim3(WF_EXACT_THREADS), 0, st_exact, dev, W, d_tiles, (const float4*)q_in, \
                       (uint4*)pipe.hits.p, (const uint4*)rng_planes, (uint32_t*)pipe.draws.p, (uint32_t*)pipe.exact[b & 1].p, \
                       (const WfCounters*)wctr, gctr)
                                if (prim) {
                                    if (alpha && counting) PT_LAUNCH_EXACT(true, true, true);
                                    else if (alpha) PT_LAUNCH_EXACT(false, true, true);
                                    else if (counting) PT_LAUNCH_EXACT(true, false, true);
                                    else PT_LAUNCH_EXACT(false, false, true);
                                } else {
                                    if (alpha && counting) PT_LAUNCH_EXACT(true, true, false);
                                    else if (alpha) PT_LAUNCH_EXACT(false, true, false);
                                    else if (counting) PT_LAUNCH_EXACT(true, false, false);
                                    else PT_LAUNCH_EXACT(false, false, false);
                                }
#undef PT_LAUNCH_EXACT
                                HIP_CHECK(hipGetLastError());
                            };
                            if (W.exact_handover == 2u) {
                                if (split_shade) {
                                    HIP_CHECK(hipEventRecord(pipe.ev_exact_go, st_main));   // (the shade pass that listed them is done)
                                    HIP_CHECK(hipStreamWaitEvent(pipe.side_exact, pipe.ev_exact_go, 0));
                                    launch_exact(pipe.side_exact);
                                    HIP_CHECK(hipEventRecord(pipe.ev_exact, pipe.side_exact));
                                } else {
                                    launch_exact(st_main);
                                }
                            }
                            if (allwide) {
                                hipLaunchKernelGGL(k_wf_list_identity, dim3((uint32_t)s.n_cu * 8u), dim3(256), 0, st_main,
                                                   (uint32_t*)pipe.deferred.p, W.list_cap, wctr, b);
                                HIP_CHECK(hipGetLastError());
                            } else {
                                PT_LAUNCH_ACP(k_wf_trace, s.trace_blocks, WF_THREADS, dev, W, d_tiles, q_in, (uint4*)pipe.hits.p,
                                              (const uint4*)rng_planes, (uint32_t*)pipe.draws.p, (uint32_t*)pipe.deferred.p,
                                              (uint32_t*)pipe.exact[b & 1].p, wctr, gctr);
                            }
                            if (W.exact_handover == 1u) {   // behind k_wf_trace: beside the shade pass over the queue if that is split off
                                if (split_shade) {
                                    HIP_CHECK(hipEventRecord(pipe.ev_exact_go, st_main));
                                    HIP_CHECK(hipStreamWaitEvent(pipe.side_exact, pipe.ev_exact_go, 0));
                                    launch_exact(pipe.side_exact);
                                    HIP_CHECK(hipEventRecord(pipe.ev_exact, pipe.side_exact));
                                } else {
                                    launch_exact(st_main);
                                }
                            }
                            if (W.defer_age) {   // the casts the drained wavefronts handed over (pt_wavefront.h)
                                // split shade pass: on a stream of its own, underneath k_wf_shade's pass over the queue
                                hipStream_t st_wide = split_shade ? pipe.side_wide : st_main;
                                // (16 Ki casts per pass; a workgroup without a cast returns at once)
                                const uint32_t wide_grid = allwide ? (uint32_t)s.trace_blocks * 4u
                                                                   : (uint32_t)s.n_cu * 4u * (WF_WIDE_LANES / 16u > 0u ? WF_WIDE_LANES / 16u : 1u);
                                uint4* wide_hits = split_shade ? list_hits : (uint4*)pipe.hits.p;
                                if (split_shade) {
                                    HIP_CHECK(hipEventRecord(pipe.ev_trace, st_main));
                                    HIP_CHECK(hipStreamWaitEvent(st_wide, pipe.ev_trace, 0));
                                }
#define PT_LAUNCH_WIDE(C, A)                                                                                                  \
    hipLaunchKernelGGL((k_wf_trace_wide<C, A>), dim3(wide_grid), dim3(WF_THREADS), 0, st_wide, dev, W, d_tiles, (const float4*)q_in, \
                       wide_hits, (const uint4*)rng_planes, (uint32_t*)pipe.draws.p, (const uint32_t*)pipe.deferred.p,             \
                       (const WfCounters*)wctr, gctr)
                                if (alpha && counting) PT_LAUNCH_WIDE(true, true);
                                else if (alpha) PT_LAUNCH_WIDE(false, true);
                                else if (counting) PT_LAUNCH_WIDE(true, false);
                                else PT_LAUNCH_WIDE(false, false);
#undef PT_LAUNCH_WIDE
                                HIP_CHECK(hipGetLastError());
                            }
                            if (split_shade) HIP_CHECK(hipEventRecord(pipe.ev_wide, pipe.side_wide));
                        }
                        stage_end();
                        ++launches;
                    }
                    // shade(b) reads the colours shadow(b-1) patched and refills the shadow queue it consumed
                    if (st_shadow != st_main && b > 0) HIP_CHECK(hipStreamWaitEvent(st_main, pipe.ev_shadow, 0));
                    // (no more workgroups than the chunk has 256-entry steps)
                    uint32_t shade_grid = std::max(1u, std::min((uint32_t)s.n_cu * (prim ? shade_blocks_b0 : shade_blocks_later),
                                                                (W.n_items + WF_SHADE_THREADS - 1u) / WF_SHADE_THREADS));
                    if (timing && grid_mode >= 2) fused_marks.push_back(ev);
                    stage_begin(2);
                    // (+4: the variants with the orthographic branch for directional lights compiled in)
                    const bool dirl = s.ortho_light_grids;
                    const uint4* shade_hits = (const uint4*)pipe.hits.p;
                    const uint32_t* shade_list = nullptr;
                    auto launch_shade = [&]() {
                        if (grid_mode == 3) {
                            if (dirl) PT_LAUNCH_SHADE_P(7);
                            else PT_LAUNCH_SHADE_P(3);
                        } else if (grid_mode == 2) {
                            if (dirl) PT_LAUNCH_SHADE_P(6);
                            else PT_LAUNCH_SHADE_P(2);
                        } else if (grid_mode == 1) {
                            if (dirl) PT_LAUNCH_SHADE_G(5);
                            else PT_LAUNCH_SHADE_G(1);
                        } else {
                            PT_LAUNCH_SHADE_G(0);
                        }
                        HIP_CHECK(hipGetLastError());
                    };
                    launch_shade();
                    if (split_shade) {
                        // ... and the casts that were with k_wf_trace_wide meanwhile: the hand-over list (at most one entry per
                        // lane of the trace grid; usually a few thousand - a launch that finds an empty list returns)
                        HIP_CHECK(hipStreamWaitEvent(st_main, pipe.ev_wide, 0));
                        if (W.exact_handover) HIP_CHECK(hipStreamWaitEvent(st_main, pipe.ev_exact, 0));
                        shade_hits = (const uint4*)((const uint32_t*)pipe.deferred.p + W.list_cap);
                        shade_list = (const uint32_t*)pipe.deferred.p;
                        shade_grid = std::min(shade_grid, (uint32_t)s.n_cu);
                        launch_shade();
                    }
                    HIP_CHECK(hipGetLastError());
                    stage_end();
                    // (inline casts leave at most a few records in the shadow queue: that launch stays on the main stream,
                    // behind a persistent trace grid on another stream it would wait milliseconds for a free slot)
                    hipStream_t st_shadow_main = st_shadow;
                    if (grid_mode != 0) st_shadow = st_main;
                    if (st_shadow != st_main) {
                        HIP_CHECK(hipEventRecord(pipe.ev_shade, st_main));
                        HIP_CHECK(hipStreamWaitEvent(st_shadow, pipe.ev_shade, 0));
                    }
                    stage_stream = st_shadow;
                    stage_begin(3);
                    WfParams Ws = W;
                    if (wf_refill_shadow) Ws.refill_min = std::min(64u, wf_refill_shadow);
                    if (wf_walk_shadow) Ws.walk_steps = wf_walk_shadow;
#define PT_OGS_ARGS                                                                                                    \
    dev, Ws, (const float4*)pipe.shadow.p, (const float4*)pipe.contrib.p, q_out, (float*)s.staging_buf.p, \
        (uint32_t*)pipe.offgrid.p, wctr, gctr
                    if (grid_mode != 0) {
                        // what is left in the shadow queue: surfaces with a normal too long for the grids' margin
                        // (normally none: the launch finds an empty queue and returns)
                        if (counting) hipLaunchKernelGGL((k_og_shadow_offgrid<true, false>), dim3((uint32_t)s.n_cu), dim3(256), 0, st_shadow, PT_OGS_ARGS);
                        else hipLaunchKernelGGL((k_og_shadow_offgrid<false, false>), dim3((uint32_t)s.n_cu), dim3(256), 0, st_shadow, PT_OGS_ARGS);
                        HIP_CHECK(hipGetLastError());
                    } else if (use_light_grids) {   // every light a point light with a grid: plain grid-stride kernel
                        static const uint32_t ogs_blocks = [] {   // workgroups per CU of k_og_shadow's grid-stride launch
                            const char* e = getenv("PT_OGS_BLOCKS");
                            return (uint32_t)(e && *e ? atoi(e) : 16);
                        }();
                        const dim3 g((uint32_t)s.n_cu * std::max(1u, ogs_blocks));
#define PT_LAUNCH_OGSH(A, C)                                                                                   \
    do {                                                                                                       \
        if (s.ortho_light_grids) hipLaunchKernelGGL((k_og_shadow<A, C, true>), g, dim3(256), 0, st_shadow, PT_OGS_ARGS); \
        else hipLaunchKernelGGL((k_og_shadow<A, C, false>), g, dim3(256), 0, st_shadow, PT_OGS_ARGS);          \
    } while (0)
                        if (alpha && counting) PT_LAUNCH_OGSH(true, true);
                        else if (alpha) PT_LAUNCH_OGSH(true, false);
                        else if (counting) PT_LAUNCH_OGSH(false, true);
                        else PT_LAUNCH_OGSH(false, false);
#undef PT_LAUNCH_OGSH
                        HIP_CHECK(hipGetLastError());
                        if (counting) hipLaunchKernelGGL((k_og_shadow_offgrid<true, true>), dim3((uint32_t)s.n_cu), dim3(256), 0, st_shadow, PT_OGS_ARGS);
                        else hipLaunchKernelGGL((k_og_shadow_offgrid<false, true>), dim3((uint32_t)s.n_cu), dim3(256), 0, st_shadow, PT_OGS_ARGS);
                        HIP_CHECK(hipGetLastError());
                    } else {
                        PT_LAUNCH_AC(k_wf_shadow, s.shadow_blocks, dev, Ws, (float4*)pipe.shadow.p,
                                     (const float4*)pipe.contrib.p, q_out, (float*)s.staging_buf.p, (uint32_t*)pipe.offgrid.p, wctr, gctr);
                        // ... and the jobs it set aside: a shadow ray the wavefront walker does not take (normally a few per mille)
                        if (Ws.exact_handover) {
                            if (counting) hipLaunchKernelGGL((k_og_shadow_offgrid<true, true>), dim3((uint32_t)s.n_cu * 4u), dim3(256), 0, st_shadow, PT_OGS_ARGS);
                            else hipLaunchKernelGGL((k_og_shadow_offgrid<false, true>), dim3((uint32_t)s.n_cu * 4u), dim3(256), 0, st_shadow, PT_OGS_ARGS);
                            HIP_CHECK(hipGetLastError());
                        }
                    }
                    stage_end();
                    stage_stream = st_main;
                    if (st_shadow != st_main) HIP_CHECK(hipEventRecord(pipe.ev_shadow, st_shadow));
                    else if (st_shadow_main != st_main) HIP_CHECK(hipEventRecord(pipe.ev_shadow, st_main));   // (keeps the waits below valid)
                    st_shadow = st_shadow_main;
#undef PT_OGS_ARGS
#undef PT_LAUNCH_ACP
#undef PT_LAUNCH_AC
#undef PT_LAUNCH_SHADE_P
#undef PT_LAUNCH_SHADE_G
#undef PT_LAUNCH_SHADE
#undef PT_SHADE_ARGS
                }
                // the next chunk clears the counters and reuses the queues, accumulate reads the staging area:
                // join the side stream
                if (st_shadow != st_main) HIP_CHECK(hipStreamWaitEvent(st_main, pipe.ev_shadow, 0));
                if (fs && stats_slots && stats_line < stats_slots) {   // this chunk's counts: one line of the frame's statistics
                    hipLaunchKernelGGL(k_wf_stats, dim3(1), dim3(64u * ((p.bounces + 3u + 63u) / 64u)), 0, st_main, (const WfCounters*)wctr,
                                       p.bounces + 3u, (uint32_t*)s.stats_dev.p + (size_t)stats_line * (p.bounces + 3u) * 4u);
                    HIP_CHECK(hipGetLastError());
                    fs->first_item_of_slot[stats_line] = base;
                    ++stats_line;
                }
            }
        }
        if (mode >= 1) {
            stage_begin(4);
            hipLaunchKernelGGL(k_accumulate, dim3(((uint32_t)tm.n_local + 255u) / 256u), dim3(256), 0, stream,
                               (const float*)s.staging_buf.p, accum, (uint32_t)tm.n_local, nb, s0 == 0 ? 1 : 0,
                               block_empty ? (const uint8_t*)(block_empty + blocks64 + 1u) : (const uint8_t*)nullptr,
                               s.dev.background[0], s.dev.background[1], s.dev.background[2]);
            HIP_CHECK(hipGetLastError());
            stage_end();
        }
        if (allow_preview && o.preview && d_rgb8) {
            // viewer feed (mod.rs:133-141): post_processing(pixel / current_sample) of the samples so far
            hipLaunchKernelGGL(k_postprocess, dim3(((uint32_t)tm.n_local + 255u) / 256u), dim3(256), 0, stream, accum,
                               (uint8_t*)d_rgb8, (uint32_t)tm.n_local, P.sample_end, p.tonemap);
            HIP_CHECK(hipGetLastError());
            std::vector<uint8_t> host(tm.n_local * 3);
            HIP_CHECK(hipMemcpyAsync(host.data(), d_rgb8, host.size(), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipStreamSynchronize(stream));
            o.preview(host.data(), tm.n_local, P.sample_end, p.samples, o.preview_user);
        }
        if (o.progress) {
            HIP_CHECK(hipStreamSynchronize(stream));
            o.progress(P.sample_end, p.samples, o.progress_user);
        }
    }
    if (fs && stats_slots && stats_line == stats_slots) {   // the frame's counts on their way to the host (read when the next frame is planned)
        HIP_CHECK(hipMemcpyAsync(fs->host, s.stats_dev.p, (size_t)stats_slots * (p.bounces + 3u) * 16u, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipEventRecord(fs->done, stream));
        fs->pending = true;
    }
    const size_t ev_post = ev;
    if (timing) HIP_CHECK(hipEventRecord(get_event(s, ev++), stream));
    if (d_rgb8) {
        hipLaunchKernelGGL(k_postprocess, dim3(((uint32_t)tm.n_local + 255u) / 256u), dim3(256), 0, stream, accum,
                           (uint8_t*)d_rgb8, (uint32_t)tm.n_local, p.samples, p.tonemap);
        HIP_CHECK(hipGetLastError());
    }
    if (timing) HIP_CHECK(hipEventRecord(get_event(s, ev++), stream));
    if (capture.st) {
        hipGraph_t graph = nullptr;
        capture.st = nullptr;
        HIP_CHECK(hipStreamEndCapture(stream, &graph));
        hipGraphExec_t exec = nullptr;
        const hipError_t rc = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_CHECK(rc);
        if (s.graphs.size() >= 16) {   // (configurations come and go with the caller's buffers: start over)
            for (auto& g : s.graphs) (void)hipGraphExecDestroy(g.second);
            s.graphs.clear();
        }
        s.graphs.emplace(std::move(graph_key), exec);
        HIP_CHECK(hipGraphLaunch(exec, user_stream));
        return;
    }

    if (timing || counting || exit_times) HIP_CHECK(hipStreamSynchronize(stream));
    if (timing) {
        pt_timing t{};
        t.launches = launches;
        t.stage_launches = stage_launches;
        float* slot[6] = {&t.generate_ms, &t.trace_ms, &t.shade_ms, &t.shadow_ms, &t.accumulate_ms, &t.integrate_ms};
        static const bool dump = getenv("PT_DEBUG_TIMES") != nullptr;
        for (auto& m : marks) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, s.events[m.second], s.events[m.second + 1]));
            *slot[m.first] += ms;
            if (dump) fprintf(stderr, "[pt] stage %d  %.3f ms\n", m.first, ms);
        }
        if (mode == 2) t.integrate_ms = t.trace_ms;  // k_wf_trace launches of the wavefront integrator
        for (size_t m : fused_marks) {
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, s.events[m], s.events[m + 1]));
            t.bounce0_ms += ms;
            ++t.bounce0_launches;
        }
        HIP_CHECK(hipEventElapsedTime(&t.postprocess_ms, s.events[ev_post], s.events[ev_post + 1]));
        HIP_CHECK(hipEventElapsedTime(&t.total_ms, s.events[0], s.events[ev_post + 1]));
        s.timing = t;
    }
#ifdef WF_EXIT_TIMES
    {   // diagnostic build: when did the wavefronts of every k_wf_trace launch run out of queue / exit? (us after launch start)
        std::unique_ptr<DevCounters> full(new DevCounters);
        HIP_CHECK(hipMemcpy(full.get(), s.counter_buf.p, sizeof(DevCounters), hipMemcpyDeviceToHost));
        for (int b = 0; b < 8; ++b) {
            std::vector<double> ex, qd;
            for (int w = 0; w < 8192; ++w)
                if (full->wave_exit[b][w]) {
                    ex.push_back((double)(full->wave_exit[b][w] - full->launch_start[b]) / 100.0);
                    if (full->wave_queue_done[b][w]) qd.push_back((double)(full->wave_queue_done[b][w] - full->launch_start[b]) / 100.0);
                }
            if (ex.empty()) continue;
            std::sort(ex.begin(), ex.end());
            std::sort(qd.begin(), qd.end());
            auto q = [](const std::vector<double>& v, double f) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
            fprintf(stderr, "[pt] trace bounce %d: %zu waves | queue exhausted (us) p1 %.0f p50 %.0f p99 %.0f max %.0f | exit p1 %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f\n",
                    b, ex.size(), q(qd, 0.01), q(qd, 0.5), q(qd, 0.99), qd.empty() ? 0.0 : qd.back(), q(ex, 0.01), q(ex, 0.1), q(ex, 0.5), q(ex, 0.9), q(ex, 0.99), q(ex, 0.999), ex.back());
        }
    }
#endif
    if (counting) {
        DevCounters c;
        HIP_CHECK(hipMemcpy(&c, s.counter_buf.p, sizeof c, hipMemcpyDeviceToHost));
        s.counters = pt_counters{c.samples, c.segments, c.shadow_rays, c.nodes_visited, c.tris_tested, c.shaded_hits,
                                 c.rng_draws, c.restarts, c.max_nodes_per_cast, c.casts_over_1k_nodes,
                                 c.trace_nodes, c.trace_tris, c.shadow_skipped, c.bounce0_hits, c.bounce0_shadow_rays,
                                 c.bounce0_tris, c.grid_tris, c.bounce0_cam_tris, c.deferred_casts, c.exact_casts, c.masked_casts, c.bounce0_masked};
        if (getenv("PT_DEBUG_HIST")) {   // casts of k_wf_trace by length (bins of 64 node visits; bin 0 not counted)
            fprintf(stderr, "[pt] cast length histogram (x64 nodes):");
            for (int b = 1; b < 16; ++b) fprintf(stderr, " %llu", c.cast_hist[b]);
            fprintf(stderr, "\n[pt] wide casts %llu: rounds total %llu max %llu | node steps %llu | time per cast (us): mean %.1f max %.1f\n",
                    c.deferred_casts, c.stamps[0], c.stamps[1], c.stamps[3],
                    c.deferred_casts ? (double)c.stamps[5] / 100.0 / (double)c.deferred_casts : 0.0, (double)c.stamps[4] / 100.0);
        }
        if (getenv("PT_DEBUG_STAMPS"))
            fprintf(stderr, "[pt] trace stamps: refill %llu walk %llu leaf %llu complete %llu cycles | walk lanes/step %.1f (%llu steps) | leaf lanes/run %.1f (%llu runs)\n",
                    c.stamps[0], c.stamps[1], c.stamps[2], c.stamps[3], c.stamps[5] ? (double)c.stamps[4] / c.stamps[5] : 0.0,
                    c.stamps[5], c.stamps[7] ? (double)c.stamps[6] / c.stamps[7] : 0.0, c.stamps[7]);
    }
}

template <class T>
struct Staged {  // host -> device copy of a test-hook input, freed on scope exit
    T* d = nullptr;
    Staged(const T* host, size_t count) {
        HIP_CHECK(hipMalloc((void**)&d, std::max<size_t>(16, count * sizeof(T))));
        if (host && count) HIP_CHECK(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    }
    ~Staged() {
        if (d) (void)hipFree(d);
    }
    void fetch(T* host, size_t count) { HIP_CHECK(hipMemcpy(host, d, count * sizeof(T), hipMemcpyDeviceToHost)); }
};

}  // namespace

// ==================================================================== C ABI
extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }
const char* pt_version(void) { return "path-tracer_amd 0.1 (gfx950)"; }

int pt_scene_create(const pt_scene_desc* desc, int device, pt_scene** out) {
    return guarded([&] {
        if (!desc || !out) fail(PT_ERR_INVALID, "pt_scene_create: null argument");
        pt_prep prep;
        auto s = std::make_unique<pt_scene>();
        prep_create(*desc, prep, s.get(), device);
        scene_upload(prep, device, *s);
        *out = s.release();
    });
}

int pt_prep_create(const pt_scene_desc* desc, pt_prep** out) {
    return guarded([&] {
        if (!desc || !out) fail(PT_ERR_INVALID, "pt_prep_create: null argument");
        auto p = std::make_unique<pt_prep>();
        prep_create(*desc, *p);
        *out = p.release();
    });
}

void pt_prep_destroy(pt_prep* prep) { delete prep; }

int pt_scene_create_from_prep(const pt_prep* prep, int device, pt_scene** out) {
    return guarded([&] {
        if (!prep || !out) fail(PT_ERR_INVALID, "pt_scene_create_from_prep: null argument");
        auto s = std::make_unique<pt_scene>();
        scene_upload(*prep, device, *s);
        *out = s.release();
    });
}

void pt_scene_destroy(pt_scene* scene) { delete scene; }

uint64_t pt_local_pixel_count(const pt_profile* profile, const pt_opts* opts) {
    uint64_t n = 0;
    guarded([&] {
        if (!profile) fail(PT_ERR_INVALID, "pt_local_pixel_count: null profile");
        pt_opts o;
        normalise_opts(*profile, opts, o);
        n = make_tile_map(*profile, o, o.shard_rank).n_local;
    });
    return n;
}

int pt_local_pixel_map(const pt_profile* profile, const pt_opts* opts, uint32_t* out) {
    return guarded([&] {
        if (!profile || !out) fail(PT_ERR_INVALID, "pt_local_pixel_map: null argument");
        pt_opts o;
        normalise_opts(*profile, opts, o);
        const pt_profile& p = *profile;
        if (o.shard_count <= 1) {
            for (uint64_t i = 0; i < (uint64_t)p.width * p.height; ++i) out[i] = (uint32_t)i;
            return;
        }
        TileMap tm = make_tile_map(p, o, o.shard_rank);
        for (uint32_t lt = 0; lt < tm.n_local_tiles; ++lt) {
            uint32_t k = tm.tiles[lt];
            uint32_t tx = k % tm.tiles_x, ty = k / tm.tiles_x;
            uint32_t cw = std::min(o.tile_w, p.width - tx * o.tile_w), ch = std::min(o.tile_h, p.height - ty * o.tile_h);
            for (uint32_t y = 0; y < ch; ++y)
                for (uint32_t x = 0; x < cw; ++x)
                    out[tm.offsets[lt] + y * cw + x] = (tx * o.tile_w + x) + (ty * o.tile_h + y) * p.width;
        }
    });
}

int pt_render_device(const pt_scene* scene, const pt_profile* profile, const pt_opts* opts, void* d_rgb8,
                     void* d_accum, void* hip_stream) {
    return guarded([&] {
        if (!scene || !profile) fail(PT_ERR_INVALID, "pt_render_device: null argument");
        render_device(*scene, *profile, opts, d_rgb8, d_accum, (hipStream_t)hip_stream);
    });
}

int pt_render(const pt_scene* scene, const pt_profile* profile, const pt_opts* opts, uint8_t* rgb8, float* accum) {
    return guarded([&] {
        if (!scene || !profile) fail(PT_ERR_INVALID, "pt_render: null argument");
        HIP_CHECK(hipSetDevice(scene->device));
        pt_opts o;
        normalise_opts(*profile, opts, o);
        uint64_t n = make_tile_map(*profile, o, o.shard_rank).n_local;
        Staged<uint8_t> d_rgb(nullptr, n * 3);
        Staged<float> d_acc(nullptr, n * 3);
        render_device(*scene, *profile, opts, d_rgb.d, d_acc.d, nullptr, true);
        HIP_CHECK(hipDeviceSynchronize());
        if (rgb8) d_rgb.fetch(rgb8, n * 3);
        if (accum) d_acc.fetch(accum, n * 3);
    });
}

int pt_debug_render(const pt_scene* scene, uint32_t width, uint32_t height, uint8_t* planes, int* any_hit) {
    return guarded([&] {
        if (!scene || !planes || !any_hit) fail(PT_ERR_INVALID, "pt_debug_render: null argument");
        if (!width || !height || (uint64_t)width * height >= (1ull << 31)) fail(PT_ERR_INVALID, "pt_debug_render: bad resolution");
        HIP_CHECK(hipSetDevice(scene->device));
        size_t bytes = (size_t)width * height * 3 * PT_DEBUG_PLANES;
        Staged<uint8_t> d_planes(nullptr, bytes);
        Staged<int> d_flag(nullptr, 1);
        HIP_CHECK(hipMemset(d_planes.d, 0, bytes));
        HIP_CHECK(hipMemset(d_flag.d, 0, sizeof(int)));
        hipLaunchKernelGGL(k_debug, dim3((width * height + 255u) / 256u), dim3(256), 0, 0, scene->dev, width, height,
                           d_planes.d, d_flag.d);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        d_planes.fetch(planes, bytes);
        d_flag.fetch(any_hit, 1);
    });
}

int pt_assemble_tiles(const pt_profile* profile, uint32_t shard_count, uint32_t tile_w, uint32_t tile_h,
                      uint64_t slice_pixels, uint32_t elem_bytes, const void* d_gathered, void* d_image,
                      void* hip_stream) {
    return guarded([&] {
        if (!profile || !d_gathered || !d_image) fail(PT_ERR_INVALID, "pt_assemble_tiles: null argument");
        pt_opts o{};
        o.shard_count = shard_count;
        o.tile_w = tile_w;
        o.tile_h = tile_h;
        pt_opts on;
        normalise_opts(*profile, &o, on);
        if (on.shard_count <= 1) {
            HIP_CHECK(hipMemcpyAsync(d_image, d_gathered, (size_t)profile->width * profile->height * elem_bytes,
                                     hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
            return;
        }
        // rank/tile offset table: built once per (resolution, shard count, tile size, device) and kept on the
        // device, so the per-frame call is a single kernel launch with no allocation and no host sync
        struct Cached {
            uint32_t* d = nullptr;
            uint32_t max_local = 0, tiles_x = 0;
        };
        static std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int>, Cached> cache;
        static std::mutex cache_mutex;
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        Cached c;
        {
            std::lock_guard<std::mutex> lock(cache_mutex);
            auto key = std::make_tuple(profile->width, profile->height, on.shard_count, on.tile_w, on.tile_h, dev);
            auto it = cache.find(key);
            if (it == cache.end()) {
                std::vector<uint32_t> table;
                for (uint32_t r = 0; r < on.shard_count; ++r) {
                    TileMap tm = make_tile_map(*profile, on, r);
                    c.tiles_x = tm.tiles_x;
                    c.max_local = std::max(c.max_local, tm.n_local_tiles);
                    table.resize(2 * (size_t)tm.tiles_x * tm.tiles_y, 0u);
                    for (uint32_t lt = 0; lt < tm.n_local_tiles; ++lt) {
                        table[2 * (size_t)tm.tiles[lt]] = r;
                        table[2 * (size_t)tm.tiles[lt] + 1] = tm.offsets[lt];
                    }
                }
                HIP_CHECK(hipMalloc((void**)&c.d, table.size() * 4));
                HIP_CHECK(hipMemcpy(c.d, table.data(), table.size() * 4, hipMemcpyHostToDevice));
                it = cache.emplace(key, c).first;
            }
            c = it->second;
        }
        for (uint32_t r = 0; r < on.shard_count; ++r)
            if (make_tile_map(*profile, on, r).n_local > slice_pixels) fail(PT_ERR_INVALID, "slice_pixels too small for rank %u", r);
        uint32_t npix = profile->width * profile->height;
        hipLaunchKernelGGL(k_assemble, dim3((npix + 255u) / 256u), dim3(256), 0, (hipStream_t)hip_stream,
                           (const uint8_t*)d_gathered, (uint8_t*)d_image, c.d, profile->width,
                           profile->height, on.tile_w, on.tile_h, c.tiles_x, slice_pixels, elem_bytes);
        HIP_CHECK(hipGetLastError());
    });
}

}  // extern "C"

// ------------------------------------------------------------------ RCCL gather
// librccl.so is loaded on first use: a single-GPU host never needs it.
namespace {
struct Rccl {
    typedef struct { char internal[128]; } UniqueId;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
    std::string error;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            r.error = std::string("cannot load librccl.so: ") + dlerror();
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl.so lacks ") + n;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    if (!r.error.empty()) fail(PT_ERR_DEVICE, "%s", r.error.c_str());
    return r;
}
#define RCCL_CHECK(expr)                                                                                           \
    do {                                                                                                           \
        int _e = (expr);                                                                                           \
        if (_e != 0) fail(PT_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
}  // namespace

struct pt_comm {
    void* comm = nullptr;
    int rank = 0, size = 1, device = 0;
};

extern "C" {

int pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES]) {
    return guarded([&] {
        if (!id) fail(PT_ERR_INVALID, "pt_comm_unique_id: null argument");
        static_assert(sizeof(Rccl::UniqueId) == PT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
        Rccl::UniqueId u;
        RCCL_CHECK(rccl().GetUniqueId(&u));
        memcpy(id, &u, sizeof u);
    });
}

int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], int rank, int size, int device, pt_comm** out) {
    return guarded([&] {
        if (!id || !out || size < 1 || rank < 0 || rank >= size) fail(PT_ERR_INVALID, "pt_comm_create: bad argument");
        select_device(device);
        auto c = std::make_unique<pt_comm>();
        HIP_CHECK(hipGetDevice(&c->device));
        c->rank = rank;
        c->size = size;
        Rccl::UniqueId u;
        memcpy(&u, id, sizeof u);
        RCCL_CHECK(rccl().CommInitRank(&c->comm, size, u, rank));
        *out = c.release();
    });
}

int pt_comm_create_all(const int* devices, int n, pt_comm** out) {
    return guarded([&] {
        if (!devices || !out || n < 1) fail(PT_ERR_INVALID, "pt_comm_create_all: bad argument");
        std::vector<void*> comms(n, nullptr);
        RCCL_CHECK(rccl().CommInitAll(comms.data(), n, devices));
        for (int i = 0; i < n; ++i) {
            out[i] = new pt_comm;
            out[i]->comm = comms[i];
            out[i]->rank = i;
            out[i]->size = n;
            out[i]->device = devices[i];
        }
    });
}

void pt_comm_destroy(pt_comm* c) {
    if (!c) return;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        (void)rccl().CommDestroy(c->comm);
    }
    delete c;
}

int pt_gather_tiles(pt_comm* comm, const pt_profile* profile, uint32_t tile_w, uint32_t tile_h, uint64_t slice_pixels,
                    uint32_t elem_bytes, const void* d_local, void* d_gathered, void* d_image, void* hip_stream) {
    return guarded([&] {
        if (!comm || !profile || !d_local || !d_gathered || !d_image) fail(PT_ERR_INVALID, "pt_gather_tiles: null argument");
        if (elem_bytes != 3 && elem_bytes != 12) fail(PT_ERR_INVALID, "pt_gather_tiles: elem_bytes must be 3 or 12");
        HIP_CHECK(hipSetDevice(comm->device));
        // one exchange step: every rank contributes its packed, zero-padded slice (ncclChar = 0)
        RCCL_CHECK(rccl().AllGather(d_local, d_gathered, (size_t)slice_pixels * elem_bytes, 0 /* ncclChar */, comm->comm,
                                    (hipStream_t)hip_stream));
        int rc = pt_assemble_tiles(profile, (uint32_t)comm->size, tile_w, tile_h, slice_pixels, elem_bytes, d_gathered, d_image,
                                   hip_stream);
        if (rc != PT_OK) throw GpuError{rc, g_err};
    });
}

int pt_render_gathered(const pt_scene* scene, pt_comm* comm, const pt_profile* profile, const pt_opts* opts,
                       uint64_t slice_pixels, uint8_t* rgb8_frame) {
    return guarded([&] {
        if (!scene || !comm || !profile || !opts) fail(PT_ERR_INVALID, "pt_render_gathered: null argument");
        if (opts->shard_count != (uint32_t)comm->size || opts->shard_rank != (uint32_t)comm->rank)
            fail(PT_ERR_INVALID, "pt_render_gathered: opts shard %u/%u does not match communicator rank %d/%d", opts->shard_rank,
                 opts->shard_count, comm->rank, comm->size);
        HIP_CHECK(hipSetDevice(scene->device));
        pt_opts o;
        normalise_opts(*profile, opts, o);
        const uint64_t n_local = make_tile_map(*profile, o, o.shard_rank).n_local;
        if (n_local > slice_pixels) fail(PT_ERR_INVALID, "pt_render_gathered: slice_pixels too small");
        const size_t npix = (size_t)profile->width * profile->height;
        Staged<uint8_t> d_local(nullptr, slice_pixels * 3), d_gathered(nullptr, slice_pixels * 3 * comm->size), d_image(nullptr, npix * 3);
        HIP_CHECK(hipMemset(d_local.d, 0, slice_pixels * 3));   // the padding of the slice
        render_device(*scene, *profile, opts, d_local.d, nullptr, nullptr);
        int rc = pt_gather_tiles(comm, profile, o.tile_w, o.tile_h, slice_pixels, 3, d_local.d, d_gathered.d, d_image.d, nullptr);
        if (rc != PT_OK) throw GpuError{rc, g_err};
        HIP_CHECK(hipDeviceSynchronize());
        if (rgb8_frame) d_image.fetch(rgb8_frame, npix * 3);
    });
}

int pt_get_timing(const pt_scene* scene, pt_timing* out) {
    if (!scene || !out) return PT_ERR_INVALID;
    *out = scene->timing;
    return PT_OK;
}
int pt_get_counters(const pt_scene* scene, pt_counters* out) {
    if (!scene || !out) return PT_ERR_INVALID;
    *out = scene->counters;
    return PT_OK;
}
int pt_get_cull_stats(const pt_scene* scene, uint32_t* n_blocks, uint32_t* n_empty) {
    return guarded([&] {
        if (!scene || !n_blocks || !n_empty) fail(PT_ERR_INVALID, "pt_get_cull_stats: null argument");
        *n_blocks = scene->last_mask_blocks;
        *n_empty = 0;
        if (scene->last_mask_blocks) {
            HIP_CHECK(hipSetDevice(scene->device));
            HIP_CHECK(hipDeviceSynchronize());
            std::vector<uint32_t> mask(scene->last_mask_blocks);
            HIP_CHECK(hipMemcpy(mask.data(), scene->pipe.block_mask.p, mask.size() * 4, hipMemcpyDeviceToHost));
            for (uint32_t m : mask) *n_empty += m != 0u;
        }
    });
}
int pt_scene_set_cu_mask(pt_scene* scene, const uint32_t* mask, uint32_t n_words) {
    return guarded([&] {
        if (!scene || (n_words && !mask)) fail(PT_ERR_INVALID, "pt_scene_set_cu_mask: null argument");
        if (scene->pipe.side || scene->n_cu) fail(PT_ERR_INVALID, "pt_scene_set_cu_mask: the scene has rendered already");
        scene->cu_mask.assign(mask, mask + n_words);
    });
}

int pt_stream_create_cu_mask(int device, const uint32_t* mask, uint32_t n_words, void** out) {
    return guarded([&] {
        if (!mask || !n_words || !out) fail(PT_ERR_INVALID, "pt_stream_create_cu_mask: null argument");
        HIP_CHECK(hipSetDevice(device));
        hipStream_t st = nullptr;
        HIP_CHECK(hipExtStreamCreateWithCUMask(&st, n_words, mask));
        *out = (void*)st;
    });
}

int pt_stream_destroy(void* stream) {
    return guarded([&] {
        if (stream) HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    });
}

int pt_scene_get_info(const pt_scene* scene, pt_scene_info* out) {
    if (!scene || !out) return PT_ERR_INVALID;
    *out = scene->info;
    out->queue_bytes = scene->queue_bytes_last;
    out->queue_chunk_items = scene->queue_chunk_last;
    out->frame_planned = scene->frame_planned_last;
    return PT_OK;
}

int pt_trace_rays(const pt_scene* scene, const float* rays, uint64_t n, pt_hit* out) {
    return guarded([&] {
        if (!scene || !rays || !out) fail(PT_ERR_INVALID, "pt_trace_rays: null argument");
        if (n == 0) return;
        HIP_CHECK(hipSetDevice(scene->device));
        Staged<float> d_rays(rays, n * 6);
        Staged<pt_hit> d_out(nullptr, n);
        hipLaunchKernelGGL(k_trace, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, scene->dev, d_rays.d, n, d_out.d);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        d_out.fetch(out, n);
    });
}

// The same through the WAVEFRONT integrator's own cast kernel (k_wf_trace: persistent lanes, resumable walk with the
// LDS stack and tree top, optional hand-over to k_wf_trace_wide) - the kernel the rays of bounces >= 1 of every frame
// go through.  mode bit 0: entry lists (trav_enter, the primitive each ray starts on in start_prims); bit 1: every
// cast is handed to k_wf_trace_wide as early as possible; bit 2: WITHOUT the hand-over of the rays the walker's slack does
// not cover to k_wf_trace_exact (study switch: the capped walk of round 3).
int pt_trace_rays_wavefront(const pt_scene* scene, const float* rays, const uint32_t* start_prims, uint64_t n, uint32_t mode, pt_hit* out) {
    return guarded([&] {
        if (!scene || !rays || !out) fail(PT_ERR_INVALID, "pt_trace_rays_wavefront: null argument");
        if ((mode & 1u) && !start_prims) fail(PT_ERR_INVALID, "pt_trace_rays_wavefront: mode 1 needs start_prims");
        if (n == 0) return;
        if (n >= (1ull << 28)) fail(PT_ERR_INVALID, "pt_trace_rays_wavefront: too many rays");
        HIP_CHECK(hipSetDevice(scene->device));
        const uint32_t cap = (uint32_t)((n + 63) & ~63ull);
        std::vector<float4> q((size_t)cap * 4 + (cap + 3) / 4, make_float4(0, 0, 0, 0));
        uint32_t* entry = (uint32_t*)(q.data() + (size_t)cap * 4);
        for (uint64_t i = 0; i < n; ++i) {
            const float* r = rays + 6 * i;
            uint32_t item = (uint32_t)i, draw = 1u << 16;
            float fi, fd;
            memcpy(&fi, &item, 4);
            memcpy(&fd, &draw, 4);
            q[2 * i] = make_float4(r[0], r[1], r[2], r[3]);
            q[2 * i + 1] = make_float4(r[4], r[5], fi, fd);
            if (mode & 1u) {
                if (start_prims[i] >= scene->dev.n_prims) fail(PT_ERR_INVALID, "pt_trace_rays_wavefront: start primitive out of range");
                entry[i] = scene->host_prim_entry[start_prims[i]];
            }
        }
        Staged<float4> d_q(q.data(), q.size());
        std::vector<WfCounters> ctr(3);
        memset(ctr.data(), 0, sizeof(WfCounters) * 3);
        ctr[1].queue_count = (uint32_t)n;
        Staged<WfCounters> d_ctr(ctr.data(), 3);
        Staged<uint4> d_hits(nullptr, (size_t)cap * 2);   // 4 B + 16 B per entry
        int n_cu = 0;
        HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, scene->device));
        const uint32_t blocks = (uint32_t)std::max(1, n_cu) * 4u;
        Staged<uint32_t> d_def(nullptr, (size_t)blocks * WF_THREADS * 7);   // (index | carried hit | progress: wf_list_*)
        WfParams W{};
        W.n_items = (uint32_t)n;
        W.cap = W.qcap_in = W.qcap_out = W.hcap = W.scap = W.ecap = cap;
        W.bounce = 1;
        W.refill_min = 16;
        W.walk_steps = 12;
        W.defer_age = (mode & 2u) ? 1u : 0u;
        W.list_cap = blocks * WF_THREADS;
        W.use_entry = mode & 1u;
        W.exact_handover = (mode & 4u) ? 0u : 1u;   // (as in every frame: the rays the walker's slack does not cover go to k_wf_trace_exact)
        Staged<uint32_t> d_exact(nullptr, (size_t)cap * 2);
        hipLaunchKernelGGL((k_wf_trace<false, false, false>), dim3(blocks), dim3(WF_THREADS), 0, 0, scene->dev, W, (const uint32_t*)nullptr,
                           d_q.d, d_hits.d, (const uint4*)nullptr, (uint32_t*)nullptr, d_def.d, d_exact.d, d_ctr.d, (DevCounters*)nullptr);
        HIP_CHECK(hipGetLastError());
        if (W.exact_handover) {
            hipLaunchKernelGGL((k_wf_trace_exact<false, false, false>), dim3((uint32_t)std::max(1, n_cu) * 16u), dim3(WF_EXACT_THREADS), 0, 0, scene->dev, W,
                               (const uint32_t*)nullptr, (const float4*)d_q.d, d_hits.d, (const uint4*)nullptr, (uint32_t*)nullptr, d_exact.d,
                               (const WfCounters*)d_ctr.d, (DevCounters*)nullptr);
            HIP_CHECK(hipGetLastError());
        }
        if (W.defer_age) {
            hipLaunchKernelGGL((k_wf_trace_wide<false, false>), dim3((uint32_t)std::max(1, n_cu) * 4u * (WF_WIDE_LANES / 16u > 0u ? WF_WIDE_LANES / 16u : 1u)), dim3(WF_THREADS), 0, 0, scene->dev, W,
                               (const uint32_t*)nullptr, (const float4*)d_q.d, d_hits.d, (const uint4*)nullptr, (uint32_t*)nullptr,
                               (const uint32_t*)d_def.d, (const WfCounters*)d_ctr.d, (DevCounters*)nullptr);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipDeviceSynchronize());
        std::vector<uint4> h((size_t)cap * 2);
        d_hits.fetch(h.data(), h.size());
        const uint32_t* word = (const uint32_t*)h.data();
        const uint4* rest = (const uint4*)(word + cap);
        for (uint64_t i = 0; i < n; ++i) {
            pt_hit& o = out[i];
            if (word[i] == 0xffffffffu) {
                o.prim = -1;
                o.flags = 0;
                o.dist = o.u = o.v = 0.f;
                continue;
            }
            const bool sphere = (word[i] & PT_PRIM_SPHERE) != 0;
            o.prim = (int32_t)(word[i] & 0x0fffffffu);
            o.flags = (int32_t)(((word[i] >> 30) & 1u) | (sphere ? 2u : 0u) | (((word[i] >> 29) & 1u) << 2));
            memcpy(&o.dist, &rest[i].x, 4);
            memcpy(&o.u, &rest[i].y, 4);
            memcpy(&o.v, &rest[i].z, 4);
            if (sphere) o.u = o.v = 0.f;
        }
    });
}

int pt_scene_escape_copy(const pt_scene* scene, void* out, uint64_t bytes) {
    return guarded([&] {
        if (!scene || !out) fail(PT_ERR_INVALID, "pt_scene_escape_copy: null argument");
        if (!scene->dev.escape && scene->escape_wanted && !scene->escape_tried) escape_masks_build(const_cast<pt_scene&>(*scene));
        if (!scene->dev.escape) fail(PT_ERR_INVALID, "pt_scene_escape_copy: the scene has no escape masks");
        if (bytes != (uint64_t)scene->dev.n_prims * 80u) fail(PT_ERR_INVALID, "pt_scene_escape_copy: %llu bytes expected", (unsigned long long)scene->dev.n_prims * 80ull);
        HIP_CHECK(hipSetDevice(scene->device));
        HIP_CHECK(hipMemcpy(out, scene->dev.escape, bytes, hipMemcpyDeviceToHost));
    });
}

int pt_scene_grid_header(const pt_scene* scene, uint32_t which, pth_origin_grid* out) {
    return guarded([&] {
        if (!scene || !out) fail(PT_ERR_INVALID, "pt_scene_grid_header: null argument");
        memset(out, 0, sizeof *out);
        if (which < scene->grid_headers.size()) *out = scene->grid_headers[which];
        out->cell_off = nullptr;
        out->refs = nullptr;
    });
}

int pt_scene_grid_copy(const pt_scene* scene, uint32_t which, uint32_t* cell_off, pth_grid_ref* refs) {
    return guarded([&] {
        if (!scene || !cell_off || !refs) fail(PT_ERR_INVALID, "pt_scene_grid_copy: null argument");
        if (which >= scene->grid_headers.size() || !scene->grid_headers[which].enabled) fail(PT_ERR_INVALID, "pt_scene_grid_copy: no such grid");
        const pth_origin_grid& h = scene->grid_headers[which];
        const DevGrid& g = which == 0 ? scene->dev.cam_grid : scene->host_light_grids[which - 1];
        HIP_CHECK(hipSetDevice(scene->device));
        HIP_CHECK(hipMemcpy(cell_off, g.cell_off, (h.n_cells + 1) * 4, hipMemcpyDeviceToHost));
        if (h.n_refs) HIP_CHECK(hipMemcpy(refs, g.refs, h.n_refs * 8, hipMemcpyDeviceToHost));
    });
}

int pt_trace_rays_all(const pt_scene* scene, const float* rays, uint64_t n, uint32_t max_hits, pt_hit* out,
                      uint32_t* counts) {
    return guarded([&] {
        if (!scene || !rays || !out || !counts || !max_hits) fail(PT_ERR_INVALID, "pt_trace_rays_all: bad argument");
        if (n == 0) return;
        HIP_CHECK(hipSetDevice(scene->device));
        Staged<float> d_rays(rays, n * 6);
        Staged<pt_hit> d_out(nullptr, n * max_hits);
        Staged<uint32_t> d_cnt(nullptr, n);
        hipLaunchKernelGGL(k_trace_all, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, scene->dev, d_rays.d, n,
                           max_hits, d_out.d, d_cnt.d);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        d_out.fetch(out, n * max_hits);
        d_cnt.fetch(counts, n);
    });
}

int pt_intersect_triangles(int device, const float* rays, const float* tris, uint64_t n, pt_hit* out) {
    return guarded([&] {
        if (!rays || !tris || !out) fail(PT_ERR_INVALID, "pt_intersect_triangles: null argument");
        if (n == 0) return;
        select_device(device);
        Staged<float> d_rays(rays, n * 6), d_tris(tris, n * 9);
        Staged<pt_hit> d_out(nullptr, n);
        hipLaunchKernelGGL(k_isect, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, d_rays.d, d_tris.d, n, d_out.d);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        d_out.fetch(out, n);
    });
}

int pt_rng_words(int device, const uint64_t* seeds, uint64_t n_seeds, uint32_t n_words, uint32_t* out) {
    return guarded([&] {
        if (!seeds || !out) fail(PT_ERR_INVALID, "pt_rng_words: null argument");
        if (n_seeds == 0 || n_words == 0) return;
        select_device(device);
        Staged<uint64_t> d_seeds(seeds, n_seeds);
        Staged<uint32_t> d_out(nullptr, n_seeds * n_words);
        hipLaunchKernelGGL(k_rng, dim3((uint32_t)((n_seeds + 255) / 256)), dim3(256), 0, 0, d_seeds.d, n_seeds, n_words, d_out.d);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        d_out.fetch(out, n_seeds * n_words);
    });
}

int pt_eval_math(int device, int fn, const float* x, uint64_t n, float* out) {
    return guarded([&] {
        if (!x || !out) fail(PT_ERR_INVALID, "pt_eval_math: null argument");
        if (fn < 0 || fn > 3) fail(PT_ERR_INVALID, "pt_eval_math: unknown function %d", fn);
        if (n == 0) return;
        select_device(device);
        Staged<float> d_x(x, n), d_out(nullptr, n);
        hipLaunchKernelGGL(k_math, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, fn, d_x.d, n, d_out.d);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        d_out.fetch(out, n);
    });
}

int pt_measure_copy_bandwidth(int device, uint64_t bytes, uint32_t reps, double* gb_per_s) {
    return guarded([&] {
        if (!gb_per_s) fail(PT_ERR_INVALID, "pt_measure_copy_bandwidth: null argument");
        if (bytes < 16 || reps == 0) fail(PT_ERR_INVALID, "pt_measure_copy_bandwidth: bytes >= 16 and reps >= 1 required");
        select_device(device);
        uint64_t n = bytes / 16;
        Staged<float4> src(nullptr, n), dst(nullptr, n);
        HIP_CHECK(hipMemset(src.d, 1, n * 16));
        if (n > (1ull << 40)) fail(PT_ERR_INVALID, "pt_measure_copy_bandwidth: at most 16 TiB");
        uint32_t grid = (uint32_t)((n + 1023) / 1024);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        float best = INFINITY;
        for (uint32_t r = 0; r <= reps; ++r) {  // the first pass only touches the pages
            HIP_CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_stream_copy, dim3(grid), dim3(256), 0, 0, src.d, dst.d, n);
            HIP_CHECK(hipEventRecord(e1, 0));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        HIP_CHECK(hipGetLastError());
        *gb_per_s = 2.0 * (double)(n * 16) / ((double)best * 1e-3) / 1e9;
    });
}

int pt_measure_gather_rate(int device, uint64_t table_bytes, uint32_t bytes_per_load, uint32_t loads_per_lane,
                           double* giga_loads_per_s) {
    return guarded([&] {
        if (!giga_loads_per_s) fail(PT_ERR_INVALID, "pt_measure_gather_rate: null argument");
        if (bytes_per_load != 8 && bytes_per_load != 16) fail(PT_ERR_INVALID, "pt_measure_gather_rate: 8 or 16 bytes per load");
        if (table_bytes < 4096 || loads_per_lane < 8) fail(PT_ERR_INVALID, "pt_measure_gather_rate: table >= 4 KiB, loads >= 8");
        select_device(device);
        uint64_t slots = 1;
        while (slots * 2 * 16 <= table_bytes) slots *= 2;   // power of two 16-byte slots
        Staged<uint4> table(nullptr, slots);
        Staged<uint32_t> sink(nullptr, 4);
        HIP_CHECK(hipMemset(table.d, 0x5a, slots * 16));
        int n_cu = 0;
        HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
        const uint32_t grid = (uint32_t)n_cu * 8u, rounds = loads_per_lane / 8u;   // 8 workgroups of 256 per CU
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        float best = INFINITY;
        for (int r = 0; r < 4; ++r) {
            HIP_CHECK(hipEventRecord(e0, 0));
            if (bytes_per_load == 16)
                hipLaunchKernelGGL(k_gather<16>, dim3(grid), dim3(256), 0, 0, table.d, (uint32_t)(slots - 1), rounds, sink.d);
            else
                hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, table.d, (uint32_t)(slots - 1), rounds, sink.d);
            HIP_CHECK(hipEventRecord(e1, 0));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        HIP_CHECK(hipGetLastError());
        *giga_loads_per_s = (double)grid * 256.0 * (double)rounds * 8.0 / ((double)best * 1e-3) / 1e9;
    });
}

}  // extern "C"

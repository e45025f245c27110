// Origin grids built ON THE DEVICE (the host builder, host/origin_grid.cpp, stays as the reference implementation and
// for the tests; both share the geometry of host/og_raster.h, so they produce the same lists byte for byte).
//
// A grid is a conservative rasterisation: every primitive's footprint (a polygon + a margin in cells + a distance
// bound, f64) is projected onto the cells of a cube map around a point / an orthographic grid along a direction, and
// every cell keeps the list of the primitives that may cover it, ascending by the distance bound.  On the GPU:
//   k_og_raster<0>   one WAVEFRONT per primitive (the rows of its bounding box dealt out to the 64 lanes: with one lane
//                    per primitive the few primitives that cover a million cells - the ground under the camera - were
//                    0.28 of the 0.33 s): footprint, one atomic per covered cell -> counts
//   k_og_block_sum / k_og_block_scan   exclusive scan of the 10^8 counts (1024-cell blocks; their sums are prefixed on
//                    the host in 64 bits: a grid must not exceed 2^32 list entries)
//   k_og_raster<1>   the same walk again, now with the cells' start words as fill cursors -> list entries
//   k_og_sort        one lane per cell: insertion sort of its (short) list by (distance bound, primitive)
// The counts live at index cell + 1 of an array of n_cells + 2 words: after the scan a word is its cell's START, the
// fill advances it to the cell's END = the next cell's start, so read from index 0 the array is the offset table.
// 0.5 s of host work per scene (two grids) becomes a few ten milliseconds, and the lists never cross PCIe.
#pragma once
#include "../host/og_raster.h"

namespace ogb {

using pth::og::Footprint;
using pth::og::GridParams;

__device__ inline Footprint og_footprint(const GridParams& P, const float* __restrict__ geom, uint32_t word, uint32_t p) {
    const float* g = geom + (size_t)p * 9;
    if (word & 0x80000000u) {   // sphere: (centre, radius)
        pt_model mo;
        mo.kind = PT_MODEL_SPHERE;
        mo.center[0] = g[0];
        mo.center[1] = g[1];
        mo.center[2] = g[2];
        mo.radius = g[3];
        return P.ortho ? pth::og::sphere_footprint_ortho(P, mo) : pth::og::sphere_footprint(P, mo);
    }
    float v[24];   // the ISF vertex stride the footprint functions expect (position first)
    v[0] = g[0]; v[1] = g[1]; v[2] = g[2];
    v[8] = g[3]; v[9] = g[4]; v[10] = g[5];
    v[16] = g[6]; v[17] = g[7]; v[18] = g[8];
    return P.ortho ? pth::og::triangle_footprint_ortho(P, v) : pth::og::triangle_footprint(P, v);
}

// PASS 0: count (cnt[cell] += 1); global primitives are appended to global_list (at most global_cap are kept, the
// counter keeps counting).  PASS 1: fill (slot = cnt[cell]++, refs[slot] = (word, bound)).
template <int PASS>
__global__ __launch_bounds__(256) void k_og_raster(GridParams P, const float* __restrict__ geom, const uint32_t* __restrict__ words,
                                                   uint32_t n_prims, uint32_t* __restrict__ cnt, uint2* __restrict__ refs,
                                                   uint32_t* __restrict__ global_list, uint32_t* __restrict__ n_global,
                                                   uint32_t global_cap) {
    const uint32_t p = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (p >= n_prims) return;
    const uint32_t word = words[p];
    const Footprint fp = og_footprint(P, geom, word, p);
    if (fp.skip) return;
    if (fp.global) {
        if (PASS == 0 && lane == 0u) {
            const uint32_t slot = atomicAdd(n_global, 1u);
            if (slot < global_cap) global_list[slot] = p;
        }
        return;
    }
    const uint2 entry = make_uint2(word, __float_as_uint(fp.mindist));
    pth::og::rasterize(P, fp, [&](size_t cell) {
        const uint32_t slot = atomicAdd(&cnt[cell], 1u);
        if (PASS == 1) refs[slot] = entry;
    }, lane, 64u);
}

#define OG_SCAN_BLOCK 1024u   // cells per block (256 threads x 4)

// (summed in 64 bits and SATURATED: 1024 cells that every one of several million large primitives overlaps hold more than
// 2^32 references between them - a wrapped sum would make the host's total too small and pass 1 write past the list array;
// the host rejects a grid with a saturated block - round-3 advisory)
__global__ __launch_bounds__(256) void k_og_block_sum(const uint32_t* __restrict__ cnt, uint64_t n_cells, uint32_t* __restrict__ block_sum) {
    __shared__ unsigned long long sh[256];
    const uint64_t base = (uint64_t)blockIdx.x * OG_SCAN_BLOCK + threadIdx.x * 4u;
    unsigned long long s = 0;
    for (uint32_t k = 0; k < 4u; ++k)
        if (base + k < n_cells) s += cnt[base + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 128u; d >= 1u; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sum[blockIdx.x] = sh[0] > 0xffffffffull ? 0xffffffffu : (uint32_t)sh[0];
}

// cnt[c] = block_off[block] + exclusive prefix inside the block
__global__ __launch_bounds__(256) void k_og_block_scan(uint32_t* __restrict__ cnt, uint64_t n_cells, const uint32_t* __restrict__ block_off) {
    __shared__ uint32_t sh[256];
    const uint64_t base = (uint64_t)blockIdx.x * OG_SCAN_BLOCK + threadIdx.x * 4u;
    uint32_t v[4], s = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        v[k] = base + k < n_cells ? cnt[base + k] : 0u;
        s += v[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t d = 1u; d < 256u; d <<= 1) {   // Hillis-Steele inclusive scan of the 256 thread sums
        const uint32_t add = threadIdx.x >= d ? sh[threadIdx.x - d] : 0u;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = block_off[blockIdx.x] + sh[threadIdx.x] - s;
    for (uint32_t k = 0; k < 4u; ++k)
        if (base + k < n_cells) {
            cnt[base + k] = run;
            run += v[k];
        }
}

// every list in ascending (distance bound, primitive) order - the host's comparator
__global__ __launch_bounds__(256) void k_og_sort(const uint32_t* __restrict__ off, uint64_t n_cells, uint2* __restrict__ refs,
                                                 uint32_t* __restrict__ longest) {
    const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    uint32_t len = 0;
    if (c < n_cells) {
        const uint32_t b = off[c], e = off[c + 1];
        len = e - b;
        for (uint32_t i = b + 1; i < e; ++i) {
            const uint2 x = refs[i];
            const float xk = __uint_as_float(x.y);
            uint32_t j = i;
            while (j > b) {
                const uint2 y = refs[j - 1];
                const float yk = __uint_as_float(y.y);
                if (!(xk < yk || (xk == yk && x.x < y.x))) break;
                refs[j] = y;
                --j;
            }
            refs[j] = x;
        }
    }
    // one atomic per wavefront
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(len, d);
        len = o > len ? o : len;
    }
    if ((threadIdx.x & 63u) == 0 && len) atomicMax(longest, len);
}

}  // namespace ogb

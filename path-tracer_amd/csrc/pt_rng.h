// rand 0.8.5 StdRng on the device: ChaCha12 keyed through PCG32.
//
// Call site in the reference: StdRng::seed_from_u64(sample + i*samples)
// (src/renderer/mod.rs:110-112); every rng.gen::<f32>() is
// (next_u32() >> 8) * 2^-24 (mod.rs:114,118,201; brdf/cook_torrance.rs:123-124;
// utils.rs:30).  Spec: SURVEY §8-a0 (rand_core 0.6 seed_from_u64 = PCG32 XSH-RR
// key expansion; rand_chacha 0.3 = 12 rounds, 64-bit block counter in words
// 12-13, stream id 0, output words consumed in order).
//
// A 64-byte block serves 16 draws; most path samples need one block
// (2.5-4.4 draws on average), so blocks are generated lazily.
// The key is not kept in registers: it is re-derived from the seed for the
// (rare) second block, which saves 8 VGPRs per lane for the whole path.
// The 16 output words live in LDS ([word][lane] layout: lane-private column,
// bank = lane, conflict-free for any per-lane word index).
#pragma once
#include "pt_math.h"

#define PT_RNG_BLOCK 256  // threads per workgroup sharing one LDS RNG slab

PT_D uint32_t pt_rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

struct PtRng {
    uint64_t seed;
    uint32_t block;   // next block counter
    uint32_t index;   // next word in the current block (16 = empty)
    uint32_t draws;   // statistics only
};

// Generate ChaCha12 block `counter` for `seed` into out[0..15] (register form).
PT_D void pt_chacha12_block(uint64_t seed, uint32_t counter, uint32_t out[16]) {
    uint32_t key[8];
    uint64_t state = seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    uint32_t x[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      counter, 0u, 0u, 0u};
#define PT_QR(a, b, c, d)                                      \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = pt_rotl(x[d], 16);      \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = pt_rotl(x[b], 12);      \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = pt_rotl(x[d], 8);       \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = pt_rotl(x[b], 7);
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        PT_QR(0, 4, 8, 12) PT_QR(1, 5, 9, 13) PT_QR(2, 6, 10, 14) PT_QR(3, 7, 11, 15)
        PT_QR(0, 5, 10, 15) PT_QR(1, 6, 11, 12) PT_QR(2, 7, 8, 13) PT_QR(3, 4, 9, 14)
    }
#undef PT_QR
    out[0] = x[0] + 0x61707865u;
    out[1] = x[1] + 0x3320646eu;
    out[2] = x[2] + 0x79622d32u;
    out[3] = x[3] + 0x6b206574u;
#pragma unroll
    for (int i = 0; i < 8; ++i) out[4 + i] = x[4 + i] + key[i];
    out[12] = x[12] + counter;
    out[13] = x[13];
    out[14] = x[14];
    out[15] = x[15];
}

PT_D void pt_rng_seed(PtRng& r, uint64_t seed) {
    r.seed = seed;
    r.block = 0;
    r.index = 16;
    r.draws = 0;
}

// slab: __shared__ uint32_t[16 * PT_RNG_BLOCK], tid = thread index in the workgroup
PT_D uint32_t pt_rng_next_u32(PtRng& r, uint32_t* slab, uint32_t tid) {
    if (r.index >= 16) {
        uint32_t w[16];
        pt_chacha12_block(r.seed, r.block, w);
#pragma unroll
        for (int i = 0; i < 16; ++i) slab[i * PT_RNG_BLOCK + tid] = w[i];
        r.block++;
        r.index = 0;
    }
    uint32_t v = slab[r.index * PT_RNG_BLOCK + tid];
    r.index++;
    return v;
}

PT_D float pt_rng_f32(PtRng& r, uint32_t* slab, uint32_t tid) {
    r.draws++;
    return (float)(pt_rng_next_u32(r, slab, tid) >> 8) * (1.0f / 16777216.0f);
}

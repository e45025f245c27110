// Wavefront integrator: the bounce loop of the reference (renderer/mod.rs:172-278) cut into
// stage kernels that exchange compacted queues in HBM, so that every lane of a wavefront runs
// the SAME stage (all traversing, or all shading) instead of serialising them against each
// other inside one fused kernel (measured: 8 of 64 lanes active per VALU instruction there).
//
//   per chunk of work items (one item = one path sample, pt_gpu.hip decode_item):
//     k_wf_rng                      ChaCha12 block 0 of every item, words 0-7 staged in two 16-byte planes (the
//                                   bounce-0 trace and shade kernels derive the camera ray in place from the
//                                   staged screen position - template parameter PRIMARY - there is no queue[0])
//     for bounce = 0 .. bounces:
//        k_wf_trace   (persistent)  ray_cast + alpha walk                   -> hit[i]
//        k_wf_shade                 material, BRDF, next ray, termination   -> queue[b+1], shadow queue
//        k_wf_shadow  (persistent)  get_light_info per light, adds direct light, retires paths
//   k_accumulate (pt_gpu.hip)       adds the staged samples per pixel in sample order
//
// Queue records are 16-byte vectors read and written by consecutive lanes (coalesced):
//   PathRec  64 B  q0 = (o.xyz, d.x)  q1 = (d.yz, thr.xy)  q2 = (thr.z, color.xyz)
//                  q3 = (item, draw_idx | bounce << 16, out_slot, 0)   item = work item of the chunk (-> RNG words)
//   HitRec   16 B  (pid | flags << 28 ... see pack_hit, key, u, v)
//   ShadowRec 64 B s0 = (pos.xyz, gn.x) s1 = (gn.yz, uv.xy) s2 = (color.xyz, bits(next_index))
//                  s3 = (bits(out_slot), bits(flags), 0, 0)     + contrib[light][k] float4
// Survivors are compacted with one atomic per wavefront (ballot -> popcount -> mbcnt prefix).
// The trace/shadow kernels are persistent: a lane whose ray is finished fetches the next queue
// entry (same ballot/popcount scheme), so long traversals do not hold 63 idle lanes.
#pragma once
#include "pt_integrator.h"
#include "pt_grid.h"
#include "pt_escape.h"

// Work item -> pixel / sample (see the item numbering in pt_gpu.hip).
struct ItemRef {
    uint32_t x, y, global_index, out_index, sample;  // sample is 1-based (mod.rs:105)
    uint32_t block;                                  // the 8x8 pixel block (a wavefront's 64 items), numbered over the rank's tiles
    bool valid;
};

__device__ __forceinline__ ItemRef decode_item(const RenderParams& P, const uint32_t* __restrict__ tile_offsets,
                                               uint32_t item) {
    ItemRef r;
    // (the four divisors are fixed for the launch: multiply-high by host-made magic numbers, ~6 instructions
    // per quotient + remainder instead of ~40 for a 32-bit division)
    uint32_t batch = P.sample_end - P.sample_begin;
    uint32_t lane = item & 63u;
    uint32_t g = item >> 6;   // < 2^26
    uint32_t b64 = pt_fastdiv(g, P.div_batch), s_rel = g - b64 * batch;
    uint32_t blocks_per_tile = (P.tile_w >> 3) * (P.tile_h >> 3);
    uint32_t lt = pt_fastdiv(b64, P.div_tile_blocks), sub = b64 - lt * blocks_per_tile;
    if (P.tile_order_base) lt = tile_offsets[P.tile_order_base + lt];   // (visiting order of the tiles, e.g. a Z curve)
    uint32_t waves_x = P.tile_w >> 3;
    uint32_t sub_y = pt_fastdiv(sub, P.div_tile_cols), sub_x = sub - sub_y * waves_x;
    uint32_t tx = sub_x * 8u + (lane & 7u);
    uint32_t ty = sub_y * 8u + (lane >> 3);
    uint32_t k = P.tile_k_base ? tile_offsets[P.tile_k_base + lt] : lt;   // (the rank's tiles, ascending; unsharded: all)
    uint32_t tile_y = pt_fastdiv(k, P.div_tiles_x), tile_x = k - tile_y * P.tiles_x;
    r.x = tile_x * P.tile_w + tx;
    r.y = tile_y * P.tile_h + ty;
    r.valid = tile_y < P.tiles_y && r.x < P.width && r.y < P.height;
    r.global_index = r.x + r.y * P.width;
    r.sample = P.sample_begin + 1u + s_rel;
    r.block = b64;
    if (P.shard_count <= 1) {
        r.out_index = r.global_index;
    } else {
        uint32_t cw = min(P.tile_w, P.width - tile_x * P.tile_w);
        r.out_index = tile_offsets[lt] + ty * cw + tx;
    }
    return r;
}

struct WfParams {
    RenderParams P;
    uint32_t item_base;   // first absolute work item of this chunk
    uint32_t n_items;     // items in this chunk
    uint32_t cap;         // work items per chunk: the stride of the RNG planes (and, when nothing smaller is known, of everything)
    // capacities of what is indexed by a QUEUE position instead of a work item (the host sizes them by the records that can
    // exist - a frame's counts are known exactly once it has been rendered once, pt_gpu.hip FramePlan):
    uint32_t qcap_in;     //   the queue this bounce reads (ray plane, path plane at 2 x qcap_in, entry plane at 4 x qcap_in)
    uint32_t qcap_out;    //   the queue this bounce writes (the next bounce's)
    uint32_t hcap;        //   hit records (word plane, then the (key, u, v) plane) and the alpha walk's draw counts
    uint32_t scap;        //   shadow records, the contrib planes' stride, the off-grid list
    uint32_t ecap;        //   the exact lists (queue indices, then the words plane)
    uint32_t rng_first_plane;   // the RNG buffer starts with this plane (1: the fused bounce-0 kernel keeps words 0-3 in registers)
    uint32_t bounce;      // current bounce (shade / shadow)
    uint32_t refill_min;  // idle lanes that trigger a queue refill in the persistent kernels
    uint32_t walk_steps;  // node steps per walking phase
    uint32_t sort_octants;  // k_wf_shade: bit 0 - survivors of a workgroup step bucketed by direction octant; bit 1 - hits shaded in material order
    uint32_t defer_age;     // k_wf_trace, queue exhausted: casts older than this many loop iterations go to k_wf_trace_wide (0: never)
    uint32_t use_entry;     // the queues carry entry words (trav_enter): casts of bounces >= 1 start at their primitive's home node
    uint32_t n_mask_blocks;    // k_cam_block_mask's table: one word per 8x8 pixel block of the rank, then "any of them empty"
    uint32_t list_cap;         // capacity of the hand-over list (queue index | carried hit | progress, wf_list_*)
    uint32_t split_deferred;   // k_wf_trace marks the casts it hands over WF_HIT_PENDING, k_wf_trace_wide stores THEIR hits by
                               // list position in the list's own plane, and k_wf_shade's pass over the queue leaves them to a
                               // second launch over that list
    uint32_t exact_handover;   // k_wf_trace / k_wf_shadow: a cast whose ray would need more slack than PT_SLACK_MAX (slack_is_capped,
                               // pt_integrator.h) is not walked here but left to k_wf_trace_exact / k_og_shadow_offgrid.  1: k_wf_trace
                               // lists it when it fetches it (camera rays of the KD-tree pipeline, test hook); 2: k_wf_shade listed it
                               // when it wrote the ray (bounces >= 1: k_wf_trace_exact then runs BESIDE k_wf_trace, not behind it),
                               // k_wf_trace only leaves it alone.  0: the capped walk of round 3 (A/B measurements only)
    uint32_t exact_shade_lists;   // k_wf_shade lists the survivors whose new ray k_wf_trace will not take (exact_handover 2 at the next bounce)
};

// A path queue of capacity `cap` records is two planes of 32 bytes per record (the casts read the first only, the misses
// of k_wf_shade the second only - that kernel runs at HBM speed):
//   ray plane   q[2 i]     = (o.xyz, d.x)        q[2 i + 1]           = (d.y, d.z, item, draw | bounce << 16)
//   path plane  q[2 cap + 2 i] = (thr.xyz, out_slot)  q[2 cap + 2 i + 1] = (colour.xyz, -)    <- patched by the shadow kernels
PT_D const float4* wf_ray_rec(const float4* q, uint32_t i) { return q + (size_t)i * 2; }
PT_D float4* wf_ray_rec(float4* q, uint32_t i) { return q + (size_t)i * 2; }
// A third plane of 4 bytes per record: the entry word of the primitive the ray starts on (trav_enter; 0 = the root)
PT_D const uint32_t* wf_entry_plane(const float4* q, uint32_t cap) { return (const uint32_t*)(q + (size_t)cap * 4); }
PT_D uint32_t* wf_entry_plane(float4* q, uint32_t cap) { return (uint32_t*)(q + (size_t)cap * 4); }
PT_D const float4* wf_path_rec(const float4* q, uint32_t cap, uint32_t i) { return q + ((size_t)cap + i) * 2; }
PT_D float4* wf_path_rec(float4* q, uint32_t cap, uint32_t i) { return q + ((size_t)cap + i) * 2; }

#ifndef WF_CURSORS
#define WF_CURSORS 8u   // fetch cursors per queue (a power of two)
#endif
#define WF_CURSOR_STRIDE 32u   // words between two cursors (128 bytes)
struct WfCounters {  // one set per bounce level, zeroed once per chunk
    uint32_t queue_count;    // records in queue[b]
    uint32_t shadow_count;   // records in the shadow queue of bounce b
    // dynamic fetch cursors (see wave_fetch): WF_CURSORS words per queue, each on a 128-byte line of its own (the
    // memory-side atomic unit serialises per line, not per word: eight cursors in one line behaved like one)
    uint32_t trace_work[WF_CURSORS * WF_CURSOR_STRIDE];
    uint32_t shadow_work[WF_CURSORS * WF_CURSOR_STRIDE];
    uint32_t overflow;       // a kernel found a list full (a capacity of the plan was wrong): the frame is not to be trusted
    uint32_t offgrid_count;  // shadow records k_og_shadow left to k_og_shadow_offgrid (pt_grid_kernels.h)
    uint32_t deferred_count; // casts k_wf_trace left to k_wf_trace_wide in its drain phase
    uint32_t exact_count;    // casts k_wf_trace left to k_wf_trace_exact when it fetched them (slack_is_capped)
};

#define WF_FLAG_TERMINATED 1u
#define WF_FLAG_SPHERE 2u

// Lane state of the persistent kernels, kept in ONE vector register: bool flags live in scalar lane
// masks, and every divergent update of such a mask costs an andn2 / and / or triple on the scalar
// unit that all waves of the CU share (the walk loop spent ~35 of its ~130 instructions there).
#define WF_LANE_WALK 0u   // walking the tree (trav_step returns these three)
#define WF_LANE_LEAF 1u   // parked at a non-empty leaf
#define WF_LANE_DONE 2u   // the cast has no further segment
#define WF_LANE_IDLE 3u   // no cast in progress
PT_D bool wf_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

PT_D uint32_t wf_lane_rank(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Reserve `popcount(ballot(want))` consecutive slots with one atomic per wavefront.
PT_D uint32_t wf_reserve(uint32_t* counter, bool want) {
    unsigned long long m = __ballot(want);
    if (!m) return 0;
    int leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if ((int)__lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + wf_lane_rank(m);
}

// Queue consumption for the persistent kernels.  One returning atomic on a single word saturates at ~88 per
// microsecond on MI355X (MI355X_MICROARCH.md, "dequeue"), so (1) a wavefront reserves a whole chunk of entries at a
// time and hands them to its idle lanes from wave-uniform registers (ballot + mbcnt prefix), and (2) the queue has
// WF_CURSORS cursor words that deal out INTERLEAVED chunks: reservation j of cursor k covers entries
// [(j * WF_CURSORS + k) * chunk, + chunk), a wavefront starts on cursor (its number mod WF_CURSORS) and moves on to
// the next one when its own is exhausted.  All wavefronts still work in one moving window of the queue (one region
// of the scene: the cache locality a partition into eight contiguous parts lost, +17 % when tried), but the
// reservations may be small without the cursor becoming the limit: with ONE word a tile shard's bounce-1 launch
// (10 M rays, 64-entry reservations) ran at exactly 88 x 64 = 5.6 G rays/s, 2/3 of the full frame's rate.
// The reservation shrinks with the queue (>= ~16 reservations per wavefront, 64 ... WF_CHUNK entries): a wavefront
// holds at most one unfinished reservation when the queue runs dry, and that is what the drain phase costs.
#ifndef WF_CHUNK
#define WF_CHUNK 512u
#endif
struct WaveFetch {
    uint32_t cur, end;
    bool done;
    uint32_t chunk;
    uint32_t cursor;   // the cursor word this wavefront draws from
};
PT_D uint32_t wf_chunk_for(uint32_t n) {
    const uint32_t per_wave = n / (gridDim.x * (blockDim.x / 64u) * 16u);
    return per_wave >= WF_CHUNK ? WF_CHUNK : per_wave >= 256u ? 256u : per_wave >= 128u ? 128u : 64u;
}
PT_D WaveFetch wf_fetch_init(uint32_t n) {
    const uint32_t wave = blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
    return WaveFetch{0u, 0u, false, wf_chunk_for(n), wave & (WF_CURSORS - 1u)};
}

PT_D uint32_t wave_fetch(WaveFetch& wf, uint32_t* cursors, uint32_t n, bool need, bool& got, bool& exhausted) {
    got = false;
    unsigned long long m = __ballot(need);
    if (!m) return 0;
    while (wf.cur >= wf.end && !wf.done) {   // (wave-uniform: at most WF_CURSORS turns, normally one)
        int leader = __ffsll((long long)m) - 1;
        uint32_t j = 0;
        if ((int)__lane_id() == leader) j = atomicAdd(&cursors[wf.cursor * WF_CURSOR_STRIDE], 1u);
        j = __builtin_amdgcn_readfirstlane(__shfl(j, leader));
        const unsigned long long base = ((unsigned long long)j * WF_CURSORS + wf.cursor) * wf.chunk;
        if (base >= n) {
            // this cursor is exhausted: LOOK at the others (plain L2 reads - 5120 wavefronts probing eight words with
            // returning atomics cost 0.4 ms per launch) and move to one that still has a reservation to give
            uint32_t next = WF_CURSORS;
            for (uint32_t t = 1; t < WF_CURSORS; ++t) {
                const uint32_t k = (wf.cursor + t) & (WF_CURSORS - 1u);
                const uint32_t seen = __hip_atomic_load(&cursors[k * WF_CURSOR_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (((unsigned long long)seen * WF_CURSORS + k) * wf.chunk < n && next == WF_CURSORS) next = k;
            }
            if (next == WF_CURSORS) wf.done = true;
            else wf.cursor = next;
        } else {
            wf.cur = (uint32_t)base;
            wf.end = base + wf.chunk < n ? (uint32_t)base + wf.chunk : n;
        }
    }
    uint32_t w = 0;
    if (wf.cur < wf.end) {
        uint32_t rank = wf_lane_rank(m), avail = wf.end - wf.cur, want = (uint32_t)__popcll(m);
        if (need && rank < avail) {
            got = true;
            w = wf.cur + rank;
        }
        wf.cur += want < avail ? want : avail;
    } else if (wf.done && need) {
        exhausted = true;
    }
    return w;
}

// ---------------------------------------------------------------------------
// Resumable KD traversal (same decisions as kd_traverse in pt_integrator.h), one node per
// call so that the kernels can run it in phases.
//
// Stack: the first WF_LDS_STACK entries of every lane live in LDS ([entry][thread], 8 B each,
// conflict-free because the entry stride is a multiple of the bank row), deeper entries
// overflow into a per-lane scratch array.  Almost all pushes stay within the LDS part, which
// takes the stack traffic off the vector-memory path (it was 2 GB of HBM writes per 33 M rays).
// ---------------------------------------------------------------------------
#ifndef WF_LDS_STACK
// 14 entries (28 KB per workgroup; with the 2 KB tree top five workgroups still share a CU's 160 KB).  With 8, the
// overflow into scratch was 12 % of the trace kernel's vector-L1 accesses (PMC: 9.69 G at 8, 8.54 G at 12, 15.1 G at 4
// entries), and those accesses are what the kernel is bound by: frame 36.40 -> 35.47 ms at 11 entries.
#define WF_LDS_STACK 14
#endif
#ifndef WF_THREADS
#define WF_THREADS 256
#endif
#ifndef WF_MIN_WAVES
#define WF_MIN_WAVES 1   // minimum waves per SIMD the trace/shadow kernels are compiled for
#endif
#ifndef WF_ALPHA_WAVES
#define WF_ALPHA_WAVES WF_MIN_WAVES   // the translucent trace of bounces >= 1: 113 registers, 4 waves (5 by launch bounds: measured below)
#endif
#ifndef WF_PRIMARY_WAVES
#define WF_PRIMARY_WAVES 5   // the bounce-0 trace and the opaque trace are held to 96 registers (they fit without spilling)
#endif

#ifndef WF_LEAN_TRAV
#define WF_LEAN_TRAV 0   // 1: the walk keeps neither 1/d nor the key scale in registers (re-derived where used): -4 registers
#endif
struct Trav {
    f3 o, d;
#if !WF_LEAN_TRAV
    f3 inv;
    float key_scale;
    float rel;   // the ray's relative slack (exit_rel, pt_integrator.h)
#endif
    float tmin, tmax;
    uint32_t node;
    uint32_t dneg;   // bit a: d[a] <= 0 (the tie rule of the child order)
    int sp;
    uint2 leaf;   // the non-empty leaf the lane is holding (valid when trav_step returned 1)
};

// The LDS column is typed with its address space: a generic pointer that may come from either LDS
// or scratch makes the compiler merge the two loads of a pop into flat (generic) loads.
typedef __attribute__((address_space(3))) unsigned long long wf_lds_u64;
struct TravStack {
    wf_lds_u64* lds;       // this thread's column: entry e at lds[e * WF_THREADS] = node | bits(tmax) << 32
    uint32_t* ov_node;     // overflow (scratch)
    float* ov_tmax;
    const wf_lds_u64* top; // the first WF_LDS_NODES node slots (the top treelets of the tree), one copy per workgroup
};

// Top of the KD-tree staged in LDS.  A walk step is one 8-byte node fetch per lane at ~44 scattered
// addresses per wavefront; the vector memory path serves those at a few cycles per lane whether they
// hit L1 or not (profiles/r01_f_trace_stamps.txt), and every ray passes through the top levels, so the
// first node slots - the device layout is breadth-first over 4-level treelets - are read from LDS.
#ifndef WF_LDS_NODES
#define WF_LDS_NODES 256   // (the top eight levels; 1024 nodes with an 11-entry stack measured 0.7 % slower than 256 with 14)
#endif
PT_D void wf_load_tree_top(const DevScene& S, unsigned long long* lds_top) {
    if (WF_LDS_NODES == 0) return;
    const uint32_t n = S.n_node_slots < (uint32_t)WF_LDS_NODES ? S.n_node_slots : (uint32_t)WF_LDS_NODES;
    for (uint32_t i = threadIdx.x; i < n; i += WF_THREADS) {
        uint2 g = S.kd_nodes[i];
        lds_top[i] = (unsigned long long)g.x | ((unsigned long long)g.y << 32);
    }
    __syncthreads();
}

// v_cndmask on a lane mask.  Written as an instruction because the optimiser turns a `?:` chain over
// the three axes into a scratch array indexed by the axis (scratch loads on the hottest path).
PT_D float wf_select(unsigned long long mask, float if_set, float if_clear) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(mask));
    return r;
}

// The LDS part is always addressed as LDS.  The (rare) overflow into scratch sits behind a
// wave-uniform branch: a per-lane "pointer select" makes the compiler fall back to flat (generic)
// loads on the hot path, and a per-lane branch costs an exec-mask save / restore on every step.
PT_D void stack_push(const TravStack& st, int sp, uint32_t node, float tmax) {
    const unsigned long long e = (unsigned long long)node | ((unsigned long long)__float_as_uint(tmax) << 32);
    if (__builtin_expect(wf_any(sp >= WF_LDS_STACK), 0)) {
        if (sp >= WF_LDS_STACK) {
            st.ov_node[sp - WF_LDS_STACK] = node;
            st.ov_tmax[sp - WF_LDS_STACK] = tmax;
        } else {
            st.lds[sp * WF_THREADS] = e;
        }
    } else {
        st.lds[sp * WF_THREADS] = e;
    }
}
PT_D void stack_get(const TravStack& st, int sp, uint32_t& node, float& tmax) {
    if (__builtin_expect(wf_any(sp >= WF_LDS_STACK), 0)) {
        if (sp >= WF_LDS_STACK) {
            node = st.ov_node[sp - WF_LDS_STACK];
            tmax = st.ov_tmax[sp - WF_LDS_STACK];
        } else {
            const unsigned long long e = st.lds[sp * WF_THREADS];
            node = (uint32_t)e;
            tmax = __uint_as_float((uint32_t)(e >> 32));
        }
    } else {
        const unsigned long long e = st.lds[sp * WF_THREADS];
        node = (uint32_t)e;
        tmax = __uint_as_float((uint32_t)(e >> 32));
    }
}

PT_D bool trav_start(const DevScene& S, Trav& T, f3 o, f3 d, float t_start) {
    T.o = o;
    T.d = d;
    // v_rcp_f32 (1 ulp) instead of three IEEE divisions (~10 instructions each): the reciprocals only
    // place the split planes along the ray, and every plane test carries a 1e-5 relative slack
    const f3 inv3 = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
#if !WF_LEAN_TRAV
    T.inv = inv3;
    float dlen = mag3(d);
    // (min(1, |d|) converts a ray parameter to key units; the early exit's relative slack is folded in: see trav_key_scale)
    T.rel = exit_rel(inv3.x, inv3.y, inv3.z);
    T.key_scale = (dlen < 1.0f ? dlen : 1.0f) / (1.0f + T.rel);
#endif
    T.dneg = (d.x <= 0.f ? 1u : 0u) | (d.y <= 0.f ? 2u : 0u) | (d.z <= 0.f ? 4u : 0u);
    float tmin = t_start, tmax = INFINITY;
    const float oa[3] = {o.x, o.y, o.z}, ia[3] = {inv3.x, inv3.y, inv3.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float tn = (S.bounds_min[a] - oa[a]) * ia[a];
        float tf = (S.bounds_max[a] - oa[a]) * ia[a];
        if (tn > tf) {
            float tmp = tn;
            tn = tf;
            tf = tmp;
        }
        tmin = tn > tmin ? tn : tmin;
        tmax = tf < tmax ? tf : tmax;
    }
    T.tmin = tmin;
    T.tmax = tmax;
    T.node = 0;
    T.sp = 0;
    return !(tmin > tmax);
}

// Move to the next stacked segment; false when the walk is over (stack empty or the segment
// starts beyond `limit`, the best hit key so far).
// key >= t * min(1, |d|) (next_hit), divided by 1 + the ray's exit slack (exit_rel): a segment that starts at tmin is
// beyond a hit with key `limit` when tmin * trav_key_scale > limit + PT_EXIT_ABS
PT_D float trav_key_scale(const Trav& T) {
#if WF_LEAN_TRAV
    const float dlen = mag3(T.d);
    return (dlen < 1.0f ? dlen : 1.0f) / (1.0f + exit_rel(__builtin_amdgcn_rcpf(T.d.x), __builtin_amdgcn_rcpf(T.d.y), __builtin_amdgcn_rcpf(T.d.z)));
#else
    return T.key_scale;
#endif
}
PT_D float trav_rel(const Trav& T) {
#if WF_LEAN_TRAV
    return exit_rel(__builtin_amdgcn_rcpf(T.d.x), __builtin_amdgcn_rcpf(T.d.y), __builtin_amdgcn_rcpf(T.d.z));
#else
    return T.rel;
#endif
}
PT_D float trav_inv_axis(const Trav& T, unsigned long long ax0, unsigned long long ax1) {   // 1 / d[axis] (v_rcp_f32, as trav_start)
#if WF_LEAN_TRAV
    return __builtin_amdgcn_rcpf(wf_select(ax0, T.d.x, wf_select(ax1, T.d.y, T.d.z)));
#else
    return wf_select(ax0, T.inv.x, wf_select(ax1, T.inv.y, T.inv.z));
#endif
}
PT_D bool trav_pop(Trav& T, const TravStack& st, float limit) {
    if (T.sp == 0) return false;
    --T.sp;
    T.tmin = T.tmax;
    stack_get(st, T.sp, T.node, T.tmax);
    return !(T.tmin * trav_key_scale(T) > limit + PT_EXIT_ABS);
}

// Start a cast whose origin lies on a primitive (every ray after the camera ray: origin = hit point + normal * 1e-5,
// mod.rs:266-268) at that primitive's HOME NODE instead of the root (the lists: prep_create, pt_gpu.hip).  `word` =
// list offset << 6 | entries; an entry = (split, far child << 3 | near is below << 2 | axis) of one ancestor of the home
// node, root first.  What the walk from the root does at an ancestor whose near child (the one towards the home node)
// holds the origin is decided by the plane parameter alone: the far child is pushed iff the ray reaches the plane
// inside the current interval.  So the ~15 dependent node fetches of the descent become one contiguous read with every
// load in flight at once, ~12 instructions per level instead of ~45, and the walk starts where the decisions start.
// The leaves visited are a superset of the root walk's (same slack on the far side; an origin exactly ON a plane takes
// both children where the root walk takes the one the ray heads into), and the accepted hit is the minimum of a total
// order over all intersected primitives: the same bits.  The origin's side of every plane is CHECKED; a cast that fails
// the check - a region estimate of the host that was too tight - walks from the root.
#ifndef WF_ENTRY_BATCH
#define WF_ENTRY_BATCH 8   // list entries loaded together (two registers each)
#endif
PT_D bool trav_enter(const DevScene& S, Trav& T, const TravStack& st, f3 o, f3 d, uint32_t word) {
    if (!trav_start(S, T, o, d, 0.f)) return false;
    const uint32_t count = word & 63u;
    if (count == 0u) return true;
    const uint4* list = (const uint4*)(S.entry_lists + (word >> 6));   // (16-byte aligned, padded: prep_create)
    const float tmax0 = T.tmax;
    float tmax = T.tmax;
    int sp = 0;
    uint32_t last_far = 0u, bad = 0u;
    for (uint32_t base = 0; base < count; base += WF_ENTRY_BATCH) {
        uint4 r[WF_ENTRY_BATCH / 2];
#pragma unroll
        for (int j = 0; j < WF_ENTRY_BATCH / 2; ++j) r[j] = list[(base >> 1) + j];   // (reads past the list stay inside the array)
#pragma unroll
        for (int j = 0; j < WF_ENTRY_BATCH; ++j) {
            if (base + j < count) {
                const uint32_t w0 = (j & 1) ? r[j >> 1].z : r[j >> 1].x, w1 = (j & 1) ? r[j >> 1].w : r[j >> 1].y;
                const float split = __uint_as_float(w0);
                const uint32_t axis = w1 & 3u;
                const unsigned long long ax0 = __builtin_amdgcn_uicmp(axis, 0u, 32), ax1 = __builtin_amdgcn_uicmp(axis, 1u, 32);
                const float o_a = wf_select(ax0, o.x, wf_select(ax1, o.y, o.z));
                const float i_a = trav_inv_axis(T, ax0, ax1);
                const float tplane = (split - o_a) * i_a;
                // the origin must be on the near side of the plane (or in it)
                bad |= (w1 & 4u) ? (o_a > split ? 1u : 0u) : (o_a < split ? 1u : 0u);
                const float rel = trav_rel(T);
                const bool push = !(tplane < 0.f) & !(tplane > __builtin_fmaf(tmax, rel, tmax + PT_EXIT_ABS));   // (a NaN parameter: both)
                last_far = w1 >> 3;
                if (push) {
                    stack_push(st, sp, last_far, tmax);
                    ++sp;
                    tmax = fmaxf(fminf(tplane, tmax), T.tmin);
                }
            }
        }
    }
    if (bad) {   // start at the root after all
        T.tmax = tmax0;
        return true;
    }
    T.node = last_far ^ 1u;   // sibling pairs are adjacent: (pair, pair + 1), pair even
    T.sp = sp;
    T.tmax = tmax;
    return true;
}

// One node.  Returns 0 = still walking, 1 = holding a non-empty leaf (T.leaf), 2 = walk over.
//
// Written without short-circuit operators and if/else ladders: every `||`, `&&` and ladder rung
// costs an s_and_saveexec / s_cbranch pair, and the scalar unit is shared by the 20 waves of a CU —
// the branchy form of this step was ~110 instructions, more than half of them scalar mask juggling,
// and took ~4400 cycles per wave-step whatever the scene size (profiles/r01_f_trace_stamps.txt).
PT_D uint2 trav_node(const DevScene& S, uint32_t node, const TravStack& st) {
#if WF_LDS_NODES > 0
    // Deep lanes fetch from memory, all lanes read the (clamped) LDS slot, and the two results are merged
    // with explicit selects: they must sit in different registers, or the compiler serialises the two
    // fetches (one waits for the other's destination registers).
    const bool is_deep = node >= (uint32_t)WF_LDS_NODES;
    const unsigned long long deep = __builtin_amdgcn_ballot_w64(is_deep);
    uint2 g;
    asm volatile("" : "=v"(g.x), "=v"(g.y));   // (lanes that do not load keep whatever is there: they select the LDS word)
    if (is_deep) g = S.kd_nodes[node];
    const uint32_t top_slot = node < (uint32_t)WF_LDS_NODES ? node : (uint32_t)WF_LDS_NODES - 1u;
    const unsigned long long e = st.top[top_slot];
    uint2 nd;
    nd.x = __float_as_uint(wf_select(deep, __uint_as_float(g.x), __uint_as_float((uint32_t)e)));
    nd.y = __float_as_uint(wf_select(deep, __uint_as_float(g.y), __uint_as_float((uint32_t)(e >> 32))));
    return nd;
#else
    return S.kd_nodes[node];
#endif
}
template <bool COUNT>
PT_D uint32_t trav_step(const DevScene& S, Trav& T, const TravStack& st, float limit, LocalCtr& lc) {
    const uint2 nd = trav_node(S, T.node, st);
    if (COUNT) lc.nodes++;
    const uint32_t axis = nd.y & 3u;
    if (axis != 3u) {
        const float split = __uint_as_float(nd.x);
        const unsigned long long ax0 = __builtin_amdgcn_uicmp(axis, 0u, 32), ax1 = __builtin_amdgcn_uicmp(axis, 1u, 32);  // EQ
        const float o_a = wf_select(ax0, T.o.x, wf_select(ax1, T.o.y, T.o.z));
        const float i_a = trav_inv_axis(T, ax0, ax1);
        const float tplane = (split - o_a) * i_a;
        // below child first iff o < split, or o == split and d <= 0
        const uint32_t dn = (T.dneg >> axis) & 1u;
        const uint32_t bf = (o_a < split ? 1u : 0u) | (o_a == split ? dn : 0u);
        const uint32_t pair = nd.y >> 2;  // children = pair (below), pair + 1 (above)
        const uint32_t second = pair + bf, first = pair + (bf ^ 1u);
        // the slack of the exit tests is the traversal's own (not reference arithmetic): fused
        // (the ray's relative slack, exit_rel: it grows with the largest 1 / |d_axis|)
        const float rel = trav_rel(T);
        const bool only_first = (tplane > __builtin_fmaf(T.tmax, rel, T.tmax + PT_EXIT_ABS)) | (tplane <= 0.f);
        const bool only_second = !only_first & (tplane < __builtin_fmaf(-T.tmin, rel, T.tmin - PT_EXIT_ABS));
        const bool both = !(only_first | only_second);  // also for a NaN plane parameter: conservative
        if (both) {
            stack_push(st, T.sp, second, T.tmax);
            ++T.sp;
#ifndef PT_NO_LOW_CLAMP
            T.tmax = fmaxf(fminf(tplane, T.tmax), T.tmin);   // (never beyond the node's own interval, nor before its start: see kd_traverse)
#else
            T.tmax = fminf(tplane, T.tmax);
#endif
        }
        T.node = only_second ? second : first;
        return WF_LANE_WALK;
    }
    if (nd.y >> 2) {
        T.leaf = nd;
        return WF_LANE_LEAF;
    }
    return trav_pop(T, st, limit) ? WF_LANE_WALK : WF_LANE_DONE;  // empty leaf: straight on to the next segment
}

#ifndef WF_MAILBOX
#define WF_MAILBOX 0   // one-entry mailbox in leaf_closest (measured: see DESIGN.md section 4)
#endif
#ifndef WF_LDS_LEAVES
#define WF_LDS_LEAVES 0   // k_wf_trace: the leaf records of a wavefront's parked lanes staged through LDS (measured, ditto)
#endif
#define WF_LEAF_SLOTS 32u  // leaf records per wavefront and staging round (1536 B: five workgroups per CU still fit)

// Inclusive prefix sum over the 64 lanes of a wavefront.
PT_D uint32_t wf_scan_inclusive(uint32_t v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(v, d);
        if ((int)__lane_id() >= d) v += up;
    }
    return v;
}
// (The 48-byte leaf records are read with plain loads: a primitive is referenced from ~7 leaves, and
// marking these loads non-temporal cost 19 % of the trace stage.)
// Closest-hit candidate update for one leaf (same acceptance rule as next_hit()).
#ifndef WF_LEAN_HIT
#define WF_LEAN_HIT 0   // 1: the opaque cast keeps only (key, order, id) of its best hit across the walk; u, v and the flags are
#endif                  //    re-derived once, at the end, by testing the winning primitive again (same arithmetic, same bits)
template <bool COUNT, bool LEAN = false>
PT_D void leaf_closest(const DevScene& S, const Trav& T, uint2 leaf, float t_prev, uint32_t ord_prev, RawHit& best,
                       LocalCtr& lc, uint32_t& mailbox) {
    const float4* lp = S.leaf_prims + (size_t)leaf.x * 3;
    uint32_t n = leaf.y >> 2;
    for (uint32_t i = 0; i < n; ++i) {
        float4 q0, q1, q2;
        load_prim_record(lp + 3 * i, q0, q1, q2);
        uint32_t pid = __float_as_uint(q0.w);
#if WF_MAILBOX
        // a primitive is referenced from ~7 leaves: the one tested last need not be tested again (the acceptance rule
        // is idempotent: the same (distance, order) never replaces itself)
        if (pid == mailbox) continue;
        mailbox = pid;
#endif
        if (COUNT) lc.tris++;
        if (!(pid & PT_PRIM_SPHERE)) {
            float dist, u, v;
            bool bf;
            if (!isect_triangle(T.o, T.d, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q1.w, q2.x, q2.y), dist, u,
                                v, bf))
                continue;
            uint32_t ord = PT_PRIM_INDEX(pid) * 2u;
            if (key_less(t_prev, ord_prev, dist, ord) && key_less(dist, ord, best.key, best.ord)) {
                best.key = dist;
                best.ord = ord;
                best.pid = pid;
                if (!LEAN) {
                    best.u = u;
                    best.v = v;
                    best.flags = bf ? 1u : 0u;
                }
            }
        } else {
            float t[2], key[2];
            bool ex[2];
            int nh = isect_sphere(T.o, T.d, mk3(q0.x, q0.y, q0.z), q1.x, t, key, ex);
            for (int k = 0; k < nh; ++k) {
                uint32_t ord = PT_PRIM_INDEX(pid) * 2u + (ex[k] ? 1u : 0u);
                if (key[k] == key[k] && key_less(t_prev, ord_prev, key[k], ord) && key_less(key[k], ord, best.key, best.ord)) {
                    best.key = key[k];
                    best.ord = ord;
                    best.pid = pid;
                    if (!LEAN) {
                        best.u = t[k];
                        best.v = 0.f;
                        best.flags = 2u | (ex[k] ? 4u : 0u);
                    }
                }
            }
        }
    }
}
// LEAN: u, v and the flags of the winning hit (key, ord, pid), by testing that primitive once more
PT_D void rederive_hit(const DevScene& S, const Trav& T, RawHit& best) {
    const float4* pp = S.prim_pos + (size_t)PT_PRIM_INDEX(best.pid) * 3;
    float4 q0, q1, q2;
    load_prim_record(pp, q0, q1, q2);
    if (!(best.pid & PT_PRIM_SPHERE)) {
        float dist, u, v;
        bool bf;
        (void)isect_triangle(T.o, T.d, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q1.w, q2.x, q2.y), dist, u, v, bf);
        best.u = u;
        best.v = v;
        best.flags = bf ? 1u : 0u;
    } else {
        float t[2], key[2];
        bool ex[2];
        (void)isect_sphere(T.o, T.d, mk3(q0.x, q0.y, q0.z), q1.x, t, key, ex);
        const uint32_t k = best.ord & 1u;   // (entry hit first: order bit 0 = the exit hit)
        const bool second = k != 0u && !ex[0];
        best.u = second ? t[1] : t[0];
        best.v = 0.f;
        best.flags = 2u | (k ? 4u : 0u);
    }
}

PT_D float next_start(float t_prev, f3 d) {  // where a continuation cast may start (see next_hit)
    float dlen = mag3(d);
    return t_prev > 0.f ? restart_param(t_prev * (dlen > 1.0f ? 1.0f / dlen : 1.0f), 1.0f / d.x, 1.0f / d.y, 1.0f / d.z) : 0.f;
}

// HitRec packing: x = pid (bit 31 sphere), y = bits(key), z = bits(u), w = bits(v) with the
// flags in a separate word is wasteful; backface / exit flags ride in the low bits of a 5th
// value folded into `ord`: ord = prim*2 + exit is recomputable, backface is bit 30 of x.
PT_D uint4 pack_hit(const RawHit& h, bool hit) {
    if (!hit) return make_uint4(0xffffffffu, 0u, 0u, 0u);
    uint32_t x = (h.pid & 0x8fffffffu) | ((h.flags & 1u) << 30) | ((h.flags & 4u) << 27);  // bit31 sphere, bit30 backface, bit29 exit
    return make_uint4(x, __float_as_uint(h.key), __float_as_uint(h.u), __float_as_uint(h.v));
}
PT_D bool unpack_hit(uint4 r, RawHit& h) {
    if (r.x == 0xffffffffu) return false;
    bool sphere = (r.x & PT_PRIM_SPHERE) != 0;
    uint32_t prim = r.x & 0x0fffffffu;
    h.pid = prim | (sphere ? PT_PRIM_SPHERE : 0u);
    h.flags = ((r.x >> 30) & 1u) | (sphere ? 2u : 0u) | (((r.x >> 29) & 1u) << 2);
    h.key = __uint_as_float(r.y);
    h.u = __uint_as_float(r.z);
    h.v = __uint_as_float(r.w);
    h.ord = prim * 2u + ((h.flags >> 2) & 1u);
    return true;
}

// The hit records of a chunk, two planes (k_wf_shade's pass over the queue reads the first only; a miss writes the first
// only): hits[i] as a 4-byte word - primitive | flags, 0xffffffff = no hit - for i < cap, then (key, u, v, -) per entry.
#define WF_HIT_PENDING 0xfffffffeu   // (bit 28 of a hit's word is always clear) the cast is still with k_wf_trace_wide
PT_D uint32_t wf_hit_word(const uint4* hits, uint32_t i) { return ((const uint32_t*)hits)[i]; }
PT_D void wf_store_hit(uint4* hits, uint32_t cap, uint32_t i, const RawHit& h, bool hit) {
    const uint4 r = pack_hit(h, hit);
    ((uint32_t*)hits)[i] = r.x;
    if (hit) ((uint4*)((uint32_t*)hits + cap))[i] = make_uint4(r.y, r.z, r.w, 0u);
}
PT_D bool wf_load_hit(const uint4* hits, uint32_t cap, uint32_t i, RawHit& h) {
    const uint32_t x = ((const uint32_t*)hits)[i];
    if (x == 0xffffffffu) return false;
    const uint4 k = ((const uint4*)((const uint32_t*)hits + cap))[i];
    return unpack_hit(make_uint4(x, k.x, k.y, k.z), h);
}

// The hand-over list (k_wf_trace -> k_wf_trace_wide), list_cap entries: the queue index (4 B), a hit plane like the
// chunk's (20 B: the best hit of the leaves walked so far, later the cast's result if the shade pass is split) and the
// distance up to which the tree has been walked (4 B).
PT_D uint4* wf_list_hits(uint32_t* list, uint32_t list_cap) { return (uint4*)(list + list_cap); }
PT_D const uint4* wf_list_hits(const uint32_t* list, uint32_t list_cap) { return (const uint4*)(list + list_cap); }
PT_D float* wf_list_progress(uint32_t* list, uint32_t list_cap) { return (float*)(list + (size_t)list_cap * 6); }
PT_D const float* wf_list_progress(const uint32_t* list, uint32_t list_cap) { return (const float*)(list + (size_t)list_cap * 6); }
// Translucent scenes (28 -> 60 B): the state of the alpha walk (mod.rs:188-205) - the last skipped hit as the lower bound of the
// search (t_prev, ord_prev), the path's rng.gen() count, the last skipped surface ("kept", shaded if every hit is skipped).
#define WF_LIST_WORDS_OPAQUE 7u
#define WF_LIST_WORDS_ALPHA 15u
PT_D void wf_store_alpha_state(uint32_t* list, uint32_t cap, uint32_t slot, float t_prev, uint32_t ord_prev, uint32_t draw,
                               const RawHit& kept, bool have_kept) {
    list[(size_t)cap * 7 + slot] = __float_as_uint(t_prev);
    list[(size_t)cap * 8 + slot] = ord_prev;
    list[(size_t)cap * 9 + slot] = draw;
    const uint4 r = pack_hit(kept, have_kept);
    list[(size_t)cap * 10 + slot] = r.x;
    if (have_kept) ((uint4*)(list + (size_t)cap * 11))[slot] = make_uint4(r.y, r.z, r.w, 0u);
}
PT_D bool wf_load_alpha_state(const uint32_t* list, uint32_t cap, uint32_t slot, float& t_prev, uint32_t& ord_prev, uint32_t& draw,
                              RawHit& kept) {
    t_prev = __uint_as_float(list[(size_t)cap * 7 + slot]);
    ord_prev = list[(size_t)cap * 8 + slot];
    draw = list[(size_t)cap * 9 + slot];
    const uint32_t x = list[(size_t)cap * 10 + slot];
    if (x == 0xffffffffu) return false;
    const uint4 k = ((const uint4*)(list + (size_t)cap * 11))[slot];
    unpack_hit(make_uint4(x, k.x, k.y, k.z), kept);
    return true;
}
PT_D void wf_store_carry(uint32_t* list, uint32_t list_cap, uint32_t slot, const RawHit& h) {
    const bool hit = h.pid != 0xffffffffu;
    const uint4 r = pack_hit(h, hit);
    list[list_cap + slot] = r.x;
    if (hit) ((uint4*)(list + (size_t)list_cap * 2))[slot] = make_uint4(r.y, r.z, r.w, 0u);
}
PT_D bool wf_load_carry(const uint32_t* list, uint32_t list_cap, uint32_t slot, RawHit& h) {
    const uint32_t x = list[list_cap + slot];
    if (x == 0xffffffffu) return false;
    const uint4 k = ((const uint4*)(list + (size_t)list_cap * 2))[slot];
    unpack_hit(make_uint4(x, k.x, k.y, k.z), h);
    return true;
}

// The exact list (k_wf_trace -> k_wf_trace_exact), `cap` entries (a queue entry is listed at most once): the queue index
// (4 B), then - split shade pass - the hit's WORD by list position (4 B); the rest of the hit (key, u, v) goes to the chunk's
// own hit plane at the queue index, whose word stays WF_HIT_PENDING while the pass of k_wf_shade over the queue is running.
PT_D uint32_t* wf_exact_words(uint32_t* list, uint32_t cap) { return list + cap; }
PT_D const uint32_t* wf_exact_words(const uint32_t* list, uint32_t cap) { return list + cap; }

PT_D float wf_rng_float(uint32_t word) { return (float)(word >> 8) * (1.0f / 16777216.0f); }  // rng.gen::<f32>()

// ---------------------------------------------------------------------------
// Camera-grid cull (bounce 0, fused kernel): which 8x8 pixel blocks can no camera ray hit anything in?  One thread per
// pixel, every frame (a few ten microseconds).  A pixel's jittered rays (mod.rs:110-120: x + r1, y + r2 with r in [0, 1))
// are positive combinations of its four corner rays (r = 0 and r = 1.0f: primary_screen is monotone in r), so on ONE face
// of the cube map - a convex cone - they stay on that face and their cell coordinates, a projective image of the pixel
// rectangle, stay inside the bounding box of the corners' cells.  If every cell of that box, one more ring of cells for
// the rounding of og_cell, has an empty list (and the grid lists no primitive globally) the lists - conservative by
// construction (host/og_raster.h) - prove that every such ray misses: the sample's colour is the background whatever
// its random numbers are, and its StdRng feeds nothing else (one generator per sample).  Boxes that touch a face's
// edge, corners on different faces: not empty.  A block is empty if its 64 pixels are (pixels outside the image count).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cam_block_mask(DevScene S, RenderParams P, const uint32_t* __restrict__ tile_offsets,
                                                        uint32_t n_blocks, uint32_t* __restrict__ block_empty,
                                                        uint8_t* __restrict__ pixel_empty) {
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;   // block * 64 + pixel of the block (a batch of ONE sample)
    const bool in_range = (item >> 6) < n_blocks;
    bool empty = true;
    uint32_t out_index = 0xffffffffu;
    if (in_range) {
        const ItemRef it = decode_item(P, tile_offsets, item);
        if (it.valid) {
            out_index = it.out_index;
            const DevGrid& G = S.cam_grid;
            empty = G.n_global == 0u;
            uint32_t face0 = 0, cu0 = 0xffffffffu, cu1 = 0, cv0 = 0xffffffffu, cv1 = 0;
            for (uint32_t c = 0; c < 4u && empty; ++c) {
                float sx, sy;
                f3 o, d;
                primary_screen(S, it.x, it.y, P.width, P.height, (c & 1u) ? 1.0f : 0.0f, (c & 2u) ? 1.0f : 0.0f, sx, sy);
                primary_from_screen(S, sx, sy, o, d);
                if (!(d.x == d.x && d.y == d.y && d.z == d.z)) empty = false;
                uint32_t face, cu, cv;
                og_cell_coords(G, d, face, cu, cv);   // (og_cell's own arithmetic)
                if (c == 0u) face0 = face;
                else if (face != face0) empty = false;
                cu0 = cu < cu0 ? cu : cu0;
                cu1 = cu > cu1 ? cu : cu1;
                cv0 = cv < cv0 ? cv : cv0;
                cv1 = cv > cv1 ? cv : cv1;
            }
            // one ring of cells more; a box at the face's edge could have neighbours on another face
            if (empty && (cu0 == 0u || cv0 == 0u || cu1 + 1u >= G.res || cv1 + 1u >= G.res)) empty = false;
            if (empty && (cu1 - cu0 > 16u || cv1 - cv0 > 16u)) empty = false;   // (a grid much finer than the pixels: not worth the loop)
            if (empty) {
                for (uint32_t cv = cv0 - 1u; cv <= cv1 + 1u && empty; ++cv) {
                    const uint32_t row = (face0 * G.res + cv) * G.res;
                    // (the offsets are cumulative: a run of cells is empty iff its ends' offsets agree)
                    if (G.cell_off[row + cu1 + 2u] != G.cell_off[row + cu0 - 1u]) empty = false;
                }
            }
        }
    }
    const bool all_empty = __all(empty);
    // (per pixel for k_accumulate, which adds the background itself: an empty block's samples are never staged)
    if (out_index != 0xffffffffu) pixel_empty[out_index] = all_empty ? 1 : 0;
    if ((threadIdx.x & 63u) == 0u && in_range) {
        block_empty[item >> 6] = all_empty ? 1u : 0u;
        // word n_blocks, zeroed by the host: is ANY block empty?  (a plain store of the same value from many wavefronts - an
        // atomic count here serialised 17 000 additions on one address: 0.15 of the kernel's 0.2 ms)
        if (all_empty) block_empty[n_blocks] = 1u;
    }
}

// ---------------------------------------------------------------------------
// rng: ChaCha12 block 0 of every item of the chunk, once, with every lane busy.  The block costs ~700
// integer instructions; deriving it where it is consumed (in the refill of the bounce-0 trace at ~40 %
// lane occupancy, again in the bounce-0 shade, again at every later bounce) was 15-20 % of all vector
// instructions of a frame, and the trace / shade / shadow kernels are bound by instruction issue.
// A path uses words 0,1 for the pixel jitter and two per bounce: words 0-7 (bounces 0-2) are staged
// as two planes of 16 bytes per item (plane p of item i at planes[p * cap + i]: the bounce-0 readers
// take consecutive items, so their loads are dense); later draws re-derive the block (WfRng).
// Words 0,1 are only ever used for the pixel jitter, so their slots hold the jittered screen position
// (primary_screen) instead: the refill of the bounce-0 trace then needs no item decoding (five integer
// divisions) at all.  Items outside the image carry WF_ITEM_INVALID there.
// ---------------------------------------------------------------------------
#define WF_ITEM_INVALID 0x7fc0deadu   // a NaN pattern no arithmetic produces
__global__ __launch_bounds__(256) void k_wf_rng(DevScene S, WfParams W, const uint32_t* __restrict__ tile_offsets,
                                                uint4* __restrict__ rng_planes) {
    uint32_t rel = blockIdx.x * 256u + threadIdx.x;
    if (rel >= W.n_items) return;
    ItemRef it = decode_item(W.P, tile_offsets, W.item_base + rel);
    if (!it.valid) {
        rng_planes[rel] = make_uint4(WF_ITEM_INVALID, WF_ITEM_INVALID, 0u, 0u);
        return;
    }
    uint32_t w[16];
    pt_chacha12_block((uint64_t)it.sample + (uint64_t)it.global_index * (uint64_t)W.P.samples, 0u, w);
    float sx, sy;
    primary_screen(S, it.x, it.y, W.P.width, W.P.height, wf_rng_float(w[0]), wf_rng_float(w[1]), sx, sy);
    rng_planes[rel] = make_uint4(__float_as_uint(sx), __float_as_uint(sy), w[2], w[3]);
    rng_planes[(size_t)W.cap + rel] = make_uint4(w[4], w[5], w[6], w[7]);
}

// rng.gen::<f32>() number idx of the path that started as work item `item` of the chunk: words 0-7 come
// from the staged planes, later ones from the block re-derived on first use (kept in `fb`).
#define WF_RNG_STAGED 8u
struct WfRng {
    uint32_t block;      // index of the block held in w (0xffffffff = none)
    uint32_t w[16];
};
PT_D uint32_t wf_rng_staged(const uint4* __restrict__ planes, uint32_t cap, uint32_t first_plane, uint32_t item, uint32_t idx) {
    const uint32_t* p = (const uint32_t*)(planes + (size_t)((idx >> 2) - first_plane) * cap + item);
    return p[idx & 3u];
}
PT_D float wf_rng_draw(WfRng& fb, const WfParams& W, const uint32_t* __restrict__ tile_offsets,
                       const uint4* __restrict__ planes, uint32_t item, uint32_t idx) {
    if (idx < WF_RNG_STAGED) return wf_rng_float(wf_rng_staged(planes, W.cap, W.rng_first_plane, item, idx));
    if ((idx >> 4) != fb.block) {
        ItemRef it = decode_item(W.P, tile_offsets, W.item_base + item);
        fb.block = idx >> 4;
        pt_chacha12_block((uint64_t)it.sample + (uint64_t)it.global_index * (uint64_t)W.P.samples, fb.block, fb.w);
    }
    uint32_t word = fb.w[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) word = (idx & 15u) == (uint32_t)i ? fb.w[i] : word;
    return wf_rng_float(word);
}

// ---------------------------------------------------------------------------
// trace: persistent closest-hit traversal with the alpha walk (mod.rs:182-205)
//
// Loop of every wavefront:
//   refill   idle lanes take queue entries (once enough lanes are idle)
//   phase A  up to walk_steps node steps: interior nodes and EMPTY leaves (85 % of the leaf
//            visits) are handled here by all walking lanes together; a lane that reaches a
//            non-empty leaf parks
//   phase B  the parked lanes run Möller–Trumbore over their leaves together, then pop
// ---------------------------------------------------------------------------
// PRIMARY (bounce 0): there is no queue - entry i IS work item i of the chunk and the camera ray is derived in
// place from the screen position k_wf_rng staged (no 64 B/item ray record written and read back).  Translucent
// scenes keep the number of rng.gen() calls of a path in draws[entry]: the alpha walk below may draw.
template <bool ALPHA, bool COUNT, bool PRIMARY>
__global__ __launch_bounds__(WF_THREADS, (!COUNT && (PRIMARY || !ALPHA)) ? WF_PRIMARY_WAVES : (!COUNT ? WF_ALPHA_WAVES : WF_MIN_WAVES)) void k_wf_trace(DevScene S, WfParams W,
                                                         const uint32_t* __restrict__ tile_offsets,
                                                         float4* __restrict__ queue, uint4* __restrict__ hits,
                                                         const uint4* __restrict__ rng_planes,
                                                         uint32_t* __restrict__ draws, uint32_t* __restrict__ deferred,
                                                         uint32_t* __restrict__ exact_list,
                                                         WfCounters* __restrict__ ctr, DevCounters* __restrict__ gctr) {
    __shared__ unsigned long long lds_stack[WF_LDS_STACK * WF_THREADS];
    __shared__ unsigned long long lds_top[WF_LDS_NODES ? WF_LDS_NODES : 1];
#if WF_LDS_LEAVES
    __shared__ float4 lds_leaf_rec[WF_THREADS / 64][WF_LEAF_SLOTS * 3];
    __shared__ uint32_t lds_leaf_addr[WF_THREADS / 64][WF_LEAF_SLOTS];
#endif
    wf_load_tree_top(S, lds_top);
#ifdef WF_EXIT_TIMES
    if (gctr && threadIdx.x == 0 && W.bounce < 8) atomicMin(&gctr->launch_start[W.bounce], __builtin_amdgcn_s_memrealtime());
    unsigned long long t_queue_done = 0;
#endif
    const uint32_t n = PRIMARY ? W.n_items : min(ctr[W.bounce].queue_count, W.qcap_in);   // (beyond the capacity: an overflowed, flagged frame)
    uint32_t* cursor = ctr[W.bounce].trace_work;
    uint32_t ov_node[PT_KD_STACK - WF_LDS_STACK];
    float ov_tmax[PT_KD_STACK - WF_LDS_STACK];
    const TravStack st = {(wf_lds_u64*)(lds_stack + threadIdx.x), ov_node, ov_tmax, (const wf_lds_u64*)lds_top};
    Trav T;
    RawHit best, kept;       // kept: the last surface of the alpha walk when every hit was skipped
    bool have_kept = false;
    float t_prev = -INFINITY;
    uint32_t ord_prev = 0, idx = 0, item = 0, draw = 0;
    bool active = false, exhausted = false;
    uint32_t lstate = WF_LANE_IDLE;
    uint32_t mailbox = 0xffffffffu;
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    uint32_t cast_nodes0 = 0;
    WaveFetch wf = wf_fetch_init(n);

    // the current cast has no further segment: best = next entry of the sorted hit list (or none)
    // (called from ONE place per loop iteration to keep the kernel's register footprint small)
    auto complete = [&]() {
        lstate = WF_LANE_WALK;
        bool hit = best.pid != 0xffffffffu;
        if (hit && !hit_passes_slab(S, T.o, T.d)) {   // kdtree-ray's box test: no hits at all
            hit = false;
            have_kept = false;
        }
        bool finished = true;
        if (ALPHA && hit) {
            float opacity = hit_opacity(S, T.o, T.d, best);
            if (COUNT) lc.shaded++;
            bool stop = opacity >= 1.f;
            if (!stop && opacity > 0.001f) {
                WfRng fb;    // (draws past the staged words re-derive the block, per draw: rare)
                fb.block = 0xffffffffu;
                float r = wf_rng_draw(fb, W, tile_offsets, rng_planes, item, draw++);
                stop = r < opacity;
                if (COUNT) lc.shadow_rays++;  // (re-used as the alpha-draw counter in this kernel)
            }
            if (!stop) {  // skipped: remember it and look for the next entry
                kept = best;
                have_kept = true;
                t_prev = best.key;
                ord_prev = best.ord;
                best.key = INFINITY;
                best.ord = 0xffffffffu;
                best.pid = 0xffffffffu;
                mailbox = 0xffffffffu;
                finished = !trav_start(S, T, T.o, T.d, next_start(t_prev, T.d));
                if (COUNT) lc.restarts++;
                if (finished) hit = false;
            }
        }
        if (finished) {
            if (COUNT) {
                uint32_t nn = lc.nodes - cast_nodes0;
                atomicMax(&gctr->max_nodes_per_cast, (unsigned long long)nn);
                if (nn > 1000u) atomicAdd(&gctr->casts_over_1k_nodes, 1ull);
                if (nn >= 64u) atomicAdd(&gctr->cast_hist[nn / 64u < 15u ? nn / 64u : 15u], 1ull);
            }
            if (ALPHA && !hit && have_kept) {  // every hit skipped: the last one is shaded
                best = kept;
                hit = true;
            }
            if (WF_LEAN_HIT && !ALPHA && hit) rederive_hit(S, T, best);
            wf_store_hit(hits, W.hcap, idx, best, hit);
            if (ALPHA) draws[idx] = draw;   // rng.gen() calls of the path so far (the alpha walk may have drawn)
            active = false;
            lstate = WF_LANE_IDLE;
        }
    };

#ifdef WF_STAMPS
    // Diagnostic build only: shader-clock cycles of every loop phase, summed per wavefront (lane 0
    // adds them to spare counter slots at the end).  Never enabled in the shipped library.
    unsigned long long st_refill = 0, st_walk = 0, st_leaf = 0, st_done = 0, st_t0, st_walk_lanes = 0, st_walk_steps = 0,
                       st_leaf_lanes = 0, st_leaf_runs = 0;
#define WF_STAMP(acc)                                  \
    do {                                               \
        unsigned long long _t = __builtin_readcyclecounter(); \
        acc += _t - st_t0;                             \
        st_t0 = _t;                                    \
    } while (0)
    st_t0 = __builtin_readcyclecounter();
#else
#define WF_STAMP(acc) do {} while (0)
#endif
    while (true) {
        // ---- refill
        bool need = !active && !exhausted;
        unsigned long long m_need = __ballot(need);
        if (m_need && ((uint32_t)__popcll(m_need) >= W.refill_min || !__any(active))) {
            bool got, handover = false;
            uint32_t w = wave_fetch(wf, cursor, n, need, got, exhausted);
            if (got) {
                idx = w;
                f3 o, d;
                bool valid_item;
                uint32_t entry_word = 0u;
                if (PRIMARY) {
                    const uint2 sc = *(const uint2*)(rng_planes + idx);  // jittered screen position (k_wf_rng)
                    valid_item = sc.x != WF_ITEM_INVALID;
                    o = d = mk3(0.f, 0.f, 0.f);
                    if (valid_item) primary_from_screen(S, __uint_as_float(sc.x), __uint_as_float(sc.y), o, d);
                    if (ALPHA) {
                        item = idx;
                        draw = 2;   // the pixel jitter
                    }
                } else {
                    // (two 16-byte words of the record: these kernels run at the vector L1's request rate, ~0.97
                    // accesses per CU-cycle, so every load a ray does not need counts; queues past bounce 0 hold
                    // live paths only - items outside the image never leave the bounce-0 kernels)
                    const float4* q = wf_ray_rec(queue, idx);
                    float4 q0 = q[0], q1 = q[1];
                    entry_word = W.use_entry ? wf_entry_plane(queue, W.qcap_in)[idx] : 0u;
                    o = mk3(q0.x, q0.y, q0.z);
                    d = mk3(q0.w, q1.x, q1.y);
                    valid_item = true;
                    if (ALPHA) {
                        item = __float_as_uint(q1.z);
                        draw = __float_as_uint(q1.w) & 0xffffu;
                    }
                }
                if (COUNT && valid_item) lc.segments++;
                if (COUNT) cast_nodes0 = lc.nodes;
                t_prev = -INFINITY;
                ord_prev = 0;
                have_kept = false;
                mailbox = 0xffffffffu;
                best.key = INFINITY;
                best.ord = 0xffffffffu;
                best.pid = 0xffffffffu;
                active = true;
                // outside the image / misses the scene box: answered as "no hit" below
                lstate = (!valid_item || !(PRIMARY ? trav_start(S, T, o, d, 0.f) : trav_enter(S, T, st, o, d, entry_word))) ? WF_LANE_DONE : WF_LANE_WALK;
                // A ray that would need more slack than PT_SLACK_MAX is not walked by this walker at all (slop model,
                // pt_integrator.h): it is listed for k_wf_trace_exact and the lane takes another entry at the next refill.
                // (the reciprocals are trav_start's own: both sides of the switch-over see the same numbers)
                handover = W.exact_handover && lstate == WF_LANE_WALK &&
                           slack_is_capped(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
            }
            if (wf_any(handover)) {   // (0.24 % of random directions)
                uint32_t slot = 0u;
                if (W.exact_handover == 1u) slot = wf_reserve(&ctr[W.bounce].exact_count, handover);   // (2: k_wf_shade listed it)
                if (handover) {
                    if (W.exact_handover == 1u && slot < W.ecap) exact_list[slot] = idx;
                    else if (W.exact_handover == 1u) ctr[0].overflow = 1u;
                    if (W.split_deferred) ((uint32_t*)hits)[idx] = WF_HIT_PENDING;
                    if (COUNT) lc.segments--;   // counted by k_wf_trace_exact
                    active = false;
                    lstate = WF_LANE_IDLE;
                }
            }
        }
        WF_STAMP(st_refill);
#ifdef WF_EXIT_TIMES
        if (wf.done && !t_queue_done) t_queue_done = __builtin_amdgcn_s_memrealtime();
#endif
        if (!__any(active)) {
            if (__all(exhausted)) break;
            continue;
        }
        // ---- phase A: walk
        for (uint32_t k = 0; k < W.walk_steps; ++k) {
            const bool walking = lstate == WF_LANE_WALK;
            // (kept to a compare + branch on vcc: a popcount threshold here - leave once only a few lanes
            // still walk - put a VALU->SALU dependency on the critical path of every step and cost 18 %)
            if (!wf_any(walking)) break;
#ifdef WF_STAMPS
            st_walk_lanes += __popcll(__ballot(walking));
            st_walk_steps++;
#endif
            if (walking) lstate = trav_step<COUNT>(S, T, st, best.key, lc);
        }
        WF_STAMP(st_walk);
        // ---- phase B: primitives of the parked leaves
#ifdef WF_STAMPS
        if (__any(lstate == WF_LANE_LEAF)) {
            st_leaf_lanes += __popcll(__ballot(lstate == WF_LANE_LEAF));
            st_leaf_runs++;
        }
#endif
#if WF_LDS_LEAVES
        // Triangle vertices through LDS: the parked lanes of the wavefront need sum(n) 48-byte leaf records at
        // scattered places.  Instead of every lane loading its records one after the other (n dependent round trips
        // to memory), the wavefront lists the record numbers in LDS, loads ALL their 16-byte thirds together - lane
        // j takes third j, 64 at a time, whoever owns the record - into a staging area, and every parked lane tests
        // its records from there.
        {
            const bool parked = lstate == WF_LANE_LEAF;
            if (wf_any(parked)) {
                const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
                const uint32_t n_rec = parked ? T.leaf.y >> 2 : 0u;
                uint32_t tested = 0;
                while (true) {   // (wave-uniform: one round unless more than WF_LEAF_SLOTS records are waiting)
                    const uint32_t want = n_rec - tested;
                    const uint32_t incl = wf_scan_inclusive(want), excl = incl - want;
                    const uint32_t all = __builtin_amdgcn_readlane(incl, 63);
                    if (all == 0u) break;
                    const uint32_t total = all < WF_LEAF_SLOTS ? all : WF_LEAF_SLOTS;
                    const uint32_t grant = excl >= WF_LEAF_SLOTS ? 0u : (want < WF_LEAF_SLOTS - excl ? want : WF_LEAF_SLOTS - excl);
                    for (uint32_t k = 0; k < grant; ++k) lds_leaf_addr[wv][excl + k] = T.leaf.x + tested + k;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t j = lane; j < 3u * total; j += 64u) {
                        const uint32_t r = j / 3u;
                        lds_leaf_rec[wv][j] = S.leaf_prims[(size_t)lds_leaf_addr[wv][r] * 3 + (j - 3u * r)];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t k = 0; k < grant; ++k) {
                        const float4* rec = &lds_leaf_rec[wv][(excl + k) * 3u];
                        og_test_closest<COUNT>(T.o, T.d, rec[0], rec[1], rec[2], t_prev, ord_prev, best, lc);
                    }
                    tested += grant;
                    __builtin_amdgcn_wave_barrier();
                }
                if (parked) lstate = trav_pop(T, st, best.key) ? WF_LANE_WALK : WF_LANE_DONE;
            }
        }
#else
        if (lstate == WF_LANE_LEAF) {
            leaf_closest<COUNT, WF_LEAN_HIT && !ALPHA>(S, T, T.leaf, t_prev, ord_prev, best, lc, mailbox);
            lstate = trav_pop(T, st, best.key) ? WF_LANE_WALK : WF_LANE_DONE;
        }
#endif
        WF_STAMP(st_leaf);
        if (lstate == WF_LANE_DONE) complete();
        WF_STAMP(st_done);
        // ---- drain phase: once the queue has nothing left for this wavefront, a cast that has been going for
        // defer_age iterations is handed to k_wf_trace_wide (WF_WIDE_LANES = 32 lanes per cast) instead of keeping the wavefront - and,
        // at the end, the whole launch - waiting for one lane: the last 1 % of the wavefronts of a launch used to leave
        // 0.2 ... 0.5 ms after the median one, a fifth of the launch for an eighth of a 1080p frame.
        if (!PRIMARY && W.defer_age != 0u) {
            T.dneg += 8u;   // (bits 3 and up: the age of the cast in loop iterations; trav_start sets the register afresh)
            if (wf.done && wf.cur >= wf.end) {
                const bool defer = active && T.dneg >= W.defer_age * 8u;
                if (wf_any(defer)) {
                    const uint32_t slot = wf_reserve(&ctr[W.bounce].deferred_count, defer);
                    if (defer) {
                        // (the walk so far is not lost: the group starts at the current segment - everything nearer has
                        // been visited, front to back - with the best hit of the leaves behind it)
                        deferred[slot] = idx;
                        wf_store_carry(deferred, W.list_cap, slot, best);
                        wf_list_progress(deferred, W.list_cap)[slot] = T.tmin;
                        if (ALPHA) wf_store_alpha_state(deferred, W.list_cap, slot, t_prev, ord_prev, draw, kept, have_kept);
                        if (W.split_deferred) ((uint32_t*)hits)[idx] = WF_HIT_PENDING;
                        active = false;
                        lstate = WF_LANE_IDLE;
                    }
                }
            }
        }
    }
#ifdef WF_STAMPS
    if (COUNT && (threadIdx.x & 63u) == 0) {
        atomicAdd(&gctr->stamps[0], st_refill);
        atomicAdd(&gctr->stamps[1], st_walk);
        atomicAdd(&gctr->stamps[2], st_leaf);
        atomicAdd(&gctr->stamps[3], st_done);
        atomicAdd(&gctr->stamps[4], st_walk_lanes);
        atomicAdd(&gctr->stamps[5], st_walk_steps);
        atomicAdd(&gctr->stamps[6], st_leaf_lanes);
        atomicAdd(&gctr->stamps[7], st_leaf_runs);
    }
#endif
#ifdef WF_EXIT_TIMES
    if (gctr && (threadIdx.x & 63u) == 0 && W.bounce < 8) {
        const uint32_t w = blockIdx.x * (WF_THREADS / 64u) + (threadIdx.x >> 6);
        if (w < 8192u) {
            gctr->wave_exit[W.bounce][w] = __builtin_amdgcn_s_memrealtime();
            gctr->wave_queue_done[W.bounce][w] = t_queue_done;
        }
    }
#endif
    if (COUNT) {
        atomicAdd(&gctr->segments, (unsigned long long)lc.segments);
        atomicAdd(&gctr->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
        atomicAdd(&gctr->trace_nodes, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->trace_tris, (unsigned long long)lc.tris);
        if (ALPHA) atomicAdd(&gctr->shaded_hits, (unsigned long long)lc.shaded);
        if (ALPHA) atomicAdd(&gctr->rng_draws, (unsigned long long)lc.shadow_rays);
    }
}

// ---------------------------------------------------------------------------
// The casts k_wf_trace handed over in its drain phase, WF_WIDE_LANES lanes per cast.  The group walks ONE tree
// traversal together, in rounds (the wavefront runs in lock step: ballots and lane ranks, no atomics):
//   * a lane without a segment takes the next one from its own stack; lanes that still have none are fed by lanes
//     that have pending segments on theirs (one each, through a pool in LDS);
//   * every lane with a segment walks up to WF_WIDE_STEPS nodes, stopping at a non-empty leaf;
//   * the leaves reached in the round are tested by the whole group, lane p taking primitive p of each (the ray is
//     the group's, so any lane can test any primitive; all record loads of a leaf are in flight together).
// The cast's hit is the smallest (distance, order) any lane found.  The same answer as the one-lane walk: the accepted
// hit is the minimum of a total order over ALL intersected primitives, whichever lane tests them, and the group visits
// every leaf the one-lane walk visits, except those beyond a hit already found (segments are culled against the
// group's best distance with the traversal's usual slack; a handed-over segment starts no later than recorded).
// ---------------------------------------------------------------------------
#ifndef WF_WIDE_LANES
// lanes per cast.  One shard of eight (MI355X, config 3), the k_wf_trace_wide launches of bounces 1 / 2 / 3:
// 8: 480 / 335 / 95 us   16: 266 / 231 / 83 us   32: 207 / 200 / 69 us (profiles/r03_experiments.txt item 6)
#define WF_WIDE_LANES 32u
#endif
#ifndef WF_WIDE_STEPS
#define WF_WIDE_STEPS 2u
#endif
// Experiment (PT_WF_ALLWIDE=1, profiles/r03_experiments.txt item 8): EVERY cast of a bounce >= 1 through the cooperative
// kernel - the hand-over list becomes the identity over the queue, nothing carried.
__global__ __launch_bounds__(256) void k_wf_list_identity(uint32_t* __restrict__ list, uint32_t list_cap, WfCounters* __restrict__ ctr,
                                                          uint32_t bounce) {
    const uint32_t n = ctr[bounce].queue_count;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n; e += gridDim.x * 256u) {
        list[e] = e;
        list[list_cap + e] = 0xffffffffu;
        wf_list_progress(list, list_cap)[e] = 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr[bounce].deferred_count = n;
}

#define WF_WIDE_LIST (4u * WF_WIDE_LANES)   // leaf records a group lists per round
template <bool COUNT, bool ALPHA>
__global__ __launch_bounds__(WF_THREADS, WF_MIN_WAVES) void k_wf_trace_wide(DevScene S, WfParams W, const uint32_t* __restrict__ tile_offsets,
                                                                            const float4* __restrict__ queue,
                                                                            uint4* __restrict__ hits,
                                                                            const uint4* __restrict__ rng_planes,
                                                                            uint32_t* __restrict__ draws,
                                                                            const uint32_t* __restrict__ deferred,
                                                                            const WfCounters* __restrict__ ctr,
                                                                            DevCounters* __restrict__ gctr) {
    // (W.split_deferred: `hits` is the hand-over list's own plane, W.list_cap records, indexed by list position -
    // the pass of k_wf_shade over the queue, which runs meanwhile, must keep seeing WF_HIT_PENDING at the queue index)
    // ALPHA: the group also finishes the alpha walk of its cast (mod.rs:188-205, as k_wf_trace's complete()): every lane
    // holds the group's hit and evaluates opacity and draw alike, a skipped hit restarts the group behind it.
    constexpr uint32_t L = WF_WIDE_LANES, GROUPS = WF_THREADS / L;
    __shared__ unsigned long long lds_stack[WF_LDS_STACK * WF_THREADS];
    __shared__ unsigned long long lds_top[WF_LDS_NODES ? WF_LDS_NODES : 1];
    __shared__ uint32_t pool_node[GROUPS][L];          // segments handed over in this round ...
    __shared__ float pool_tmin[GROUPS][L], pool_tmax[GROUPS][L];
    __shared__ uint32_t leaf_list[GROUPS][WF_WIDE_LIST];   // ... and the leaf records reached in it
    const uint32_t n = ctr[W.bounce].deferred_count;
    if (n == 0u) return;   // (the usual case for a launch with a short queue)
    wf_load_tree_top(S, lds_top);
    uint32_t ov_node[PT_KD_STACK - WF_LDS_STACK];
    float ov_tmax[PT_KD_STACK - WF_LDS_STACK];
    const TravStack st = {(wf_lds_u64*)(lds_stack + threadIdx.x), ov_node, ov_tmax, (const wf_lds_u64*)lds_top};
    const uint32_t group = threadIdx.x / L, part = threadIdx.x & (L - 1u), lane = threadIdx.x & 63u;
    const unsigned long long group_mask = L == 64u ? ~0ull : ((1ull << (L & 63u)) - 1ull) << (lane & ~(L - 1u));
    const unsigned long long below = (1ull << lane) - 1ull;
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    // (whole wavefronts stay in the loops together: ballots and shuffles are wave-wide)
    const uint32_t passes = (n + gridDim.x * GROUPS - 1u) / (gridDim.x * GROUPS);
    for (uint32_t r = 0; r < passes; ++r) {
        const uint32_t e = (r * gridDim.x + blockIdx.x) * GROUPS + group;
        const uint32_t idx = e < n ? deferred[e] : 0u;
        RawHit best;
        best.key = INFINITY;
        best.ord = 0xffffffffu;
        best.pid = 0xffffffffu;
        best.u = best.v = 0.f;
        best.flags = 0u;
        Trav T;
        bool in_scene = false;
        // the alpha walk's state (group-uniform)
        float t_prev = -INFINITY;
        uint32_t ord_prev = 0u, draw = 0u, item = 0u;
        RawHit kept = best;
        bool have_kept = false;
        if (e < n) {
            const float4* q = wf_ray_rec(queue, idx);
            const float4 q0 = q[0], q1 = q[1];
            // from where k_wf_trace left the walk (less the slack a restarted walk backs off by, kd_traverse)
            const float walked = wf_list_progress(deferred, W.list_cap)[e];
            const f3 qd = mk3(q0.w, q1.x, q1.y);
            const float t_start = walked > 0.f ? restart_param(walked, 1.0f / qd.x, 1.0f / qd.y, 1.0f / qd.z) : 0.f;
            in_scene = trav_start(S, T, mk3(q0.x, q0.y, q0.z), qd, t_start);
            if (part == 0u) {   // ... with the best hit of the leaves behind that point
                RawHit carried;
                if (wf_load_carry(deferred, W.list_cap, e, carried)) best = carried;
            }
            if (ALPHA) {
                item = __float_as_uint(q1.z);
                have_kept = wf_load_alpha_state(deferred, W.list_cap, e, t_prev, ord_prev, draw, kept);
            }
        }
        bool live = e < n;   // the group's cast is not answered yet
        uint32_t dbg_rounds = 0, dbg_busy = 0;
        const unsigned long long dbg_t0 = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
        bool again;
        do {
            again = false;
            bool busy = in_scene && part == 0u;   // the root segment; the other lanes live off what is handed over
            while (true) {
                if (COUNT) dbg_rounds++;
                // the group's best distance so far: what every lane culls against
                float gkey = best.key;
#pragma unroll
                for (uint32_t m = L / 2u; m >= 1u; m >>= 1) {
                    const float k2 = __shfl_xor(gkey, (int)m);
                    gkey = k2 < gkey ? k2 : gkey;
                }
                // a lane without a segment goes on with its own stack (nothing left within gkey: the rest is dropped) ...
                if (in_scene && !busy && T.sp > 0) {
                    busy = trav_pop(T, st, gkey);
                    if (!busy) T.sp = 0;
                }
                // ... or is handed the nearest pending segment of a lane that has one
                const bool need = in_scene && !busy;
                if (wf_any(need)) {
                    const bool can_give = busy && T.sp > 0;
                    const unsigned long long m_need = __ballot(need) & group_mask, m_give = __ballot(can_give) & group_mask;
                    const uint32_t n_need = (uint32_t)__popcll(m_need), n_can = (uint32_t)__popcll(m_give);
                    const uint32_t n_give = n_need < n_can ? n_need : n_can;
                    if (can_give && (uint32_t)__popcll(m_give & below) < n_give) {
                        const uint32_t slot = (uint32_t)__popcll(m_give & below);
                        uint32_t node;
                        float tmax;
                        --T.sp;
                        stack_get(st, T.sp, node, tmax);
                        pool_node[group][slot] = node;
                        pool_tmin[group][slot] = T.tmax;   // (its true start if nothing nearer was handed over before: never later)
                        pool_tmax[group][slot] = tmax;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (need && (uint32_t)__popcll(m_need & below) < n_give) {
                        const uint32_t slot = (uint32_t)__popcll(m_need & below);
                        T.node = pool_node[group][slot];
                        T.tmin = pool_tmin[group][slot];
                        T.tmax = pool_tmax[group][slot];
                        T.sp = 0;
                        busy = !(T.tmin * trav_key_scale(T) > gkey + PT_EXIT_ABS);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (!wf_any(busy)) break;   // (a lane without a segment has an empty stack: the casts of this wavefront are done)
                // up to WF_WIDE_STEPS nodes
                bool at_leaf = false;
                for (uint32_t k = 0; k < WF_WIDE_STEPS; ++k) {
                    if (busy && !at_leaf) {
                        if (COUNT) dbg_busy++;
                        const uint32_t state = trav_step<COUNT>(S, T, st, gkey, lc);
                        at_leaf = state == WF_LANE_LEAF;
                        busy = state != WF_LANE_DONE;
                        if (!busy) T.sp = 0;
                    }
                }
                // the leaves reached: tested by the group together - the records of ALL of them are listed in LDS and lane p
                // takes entries p, p + L, ... of the list, so one round costs one trip to memory for its leaves
                if (wf_any(at_leaf)) {
                    const uint32_t my_count = at_leaf ? T.leaf.y >> 2 : 0u;
                    uint32_t incl = my_count;
#pragma unroll
                    for (uint32_t m = 1u; m < L; m <<= 1) {
                        const uint32_t up = __shfl_up(incl, (int)m);
                        if (part >= m) incl += up;
                    }
                    const uint32_t total = __shfl(incl, (int)((lane & ~(L - 1u)) + L - 1u));
                    if (at_leaf) {
                        const uint32_t off = incl - my_count;
                        for (uint32_t j = 0; j < my_count; ++j)
                            if (off + j < WF_WIDE_LIST) leaf_list[group][off + j] = T.leaf.x + j;
                        busy = false;   // (its next segment: from the stack, in the next round)
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t listed = total < WF_WIDE_LIST ? total : WF_WIDE_LIST;
                    for (uint32_t i = part; i < listed; i += L) {
                        const float4* rec = S.leaf_prims + (size_t)leaf_list[group][i] * 3;
                        float4 q0, q1, q2;
                        load_prim_record(rec, q0, q1, q2);
                        og_test_closest<COUNT>(T.o, T.d, q0, q1, q2, t_prev, ord_prev, best, lc);
                    }
                    if (__builtin_expect(wf_any(total > WF_WIDE_LIST), 0)) {   // (more records than the list holds: the owners test the rest)
                        if (at_leaf) {
                            const uint32_t off = incl - my_count;
                            for (uint32_t j = 0; j < my_count; ++j)
                                if (off + j >= WF_WIDE_LIST) {
                                    const float4* rec = S.leaf_prims + (size_t)(T.leaf.x + j) * 3;
                                    float4 q0, q1, q2;
                        load_prim_record(rec, q0, q1, q2);
                        og_test_closest<COUNT>(T.o, T.d, q0, q1, q2, t_prev, ord_prev, best, lc);
                                }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // the group's minimum of (key, ord)
            float kmin = best.key;
            uint32_t omin = best.ord;
#pragma unroll
            for (uint32_t m = L / 2u; m >= 1u; m >>= 1) {
                const float k2 = __shfl_xor(kmin, (int)m);
                const uint32_t o2 = __shfl_xor(omin, (int)m);
                if (key_less(k2, o2, kmin, omin)) {
                    kmin = k2;
                    omin = o2;
                }
            }
            const bool found = best.pid != 0xffffffffu && best.key == kmin && best.ord == omin;
            const unsigned long long winners = __ballot(found) & group_mask;
            const uint32_t h_cap = W.split_deferred ? W.list_cap : W.hcap, h_idx = W.split_deferred ? e : idx;
            if (!ALPHA) {
                if (live) {
                    if (winners) {
                        if (lane == (uint32_t)__ffsll((long long)winners) - 1u)
                            wf_store_hit(hits, h_cap, h_idx, best, hit_passes_slab(S, T.o, T.d));
                    } else if (part == 0u) {
                        wf_store_hit(hits, h_cap, h_idx, best, false);
                    }
                }
            } else {
                // every lane of the group takes the winner's record
                const int src = winners ? __ffsll((long long)winners) - 1 : (int)lane;
                RawHit win;
                win.key = __shfl(best.key, src);
                win.ord = __shfl(best.ord, src);
                win.pid = __shfl(best.pid, src);
                win.u = __shfl(best.u, src);
                win.v = __shfl(best.v, src);
                win.flags = __shfl(best.flags, src);
                if (live) {
                    bool hit = winners != 0ull;
                    if (hit && !hit_passes_slab(S, T.o, T.d)) {   // kdtree-ray's box test: no hits at all
                        hit = false;
                        have_kept = false;
                    }
                    bool finished = true;
                    if (hit) {
                        const float opacity = hit_opacity(S, T.o, T.d, win);
                        if (COUNT && part == 0u) lc.shaded++;
                        bool stop = opacity >= 1.f;
                        if (!stop && opacity > 0.001f) {
                            WfRng fb;
                            fb.block = 0xffffffffu;
                            stop = wf_rng_draw(fb, W, tile_offsets, rng_planes, item, draw++) < opacity;
                            if (COUNT && part == 0u) lc.shadow_rays++;   // (the alpha-draw counter, as in k_wf_trace)
                        }
                        if (!stop) {   // skipped: remember it, the group looks for the next entry behind it
                            kept = win;
                            have_kept = true;
                            t_prev = win.key;
                            ord_prev = win.ord;
                            best.key = INFINITY;
                            best.ord = 0xffffffffu;
                            best.pid = 0xffffffffu;
                            in_scene = trav_start(S, T, T.o, T.d, next_start(t_prev, T.d));
                            if (COUNT && part == 0u) lc.restarts++;
                            finished = !in_scene;
                            if (finished) hit = false;
                        }
                    }
                    if (finished) {
                        if (!hit && have_kept) {   // every hit skipped: the last one is shaded
                            win = kept;
                            hit = true;
                        }
                        if (part == 0u) {
                            wf_store_hit(hits, h_cap, h_idx, win, hit);
                            draws[idx] = draw;
                        }
                        live = false;
                        in_scene = false;
                    } else {
                        again = true;
                    }
                }
            }
        } while (ALPHA && wf_any(again));
#ifndef WF_STAMPS   // (a -DWF_STAMPS build keeps its phase cycles in the same slots)
        if (COUNT && e < n) {   // diagnostics (PT_DEBUG_HIST): rounds per cast, node steps
            if (part == 0u) atomicAdd(&gctr->stamps[0], (unsigned long long)dbg_rounds);
            if (part == 0u) atomicMax(&gctr->stamps[1], (unsigned long long)dbg_rounds);
            atomicAdd(&gctr->stamps[3], (unsigned long long)dbg_busy);
            if (part == 0u) atomicMax(&gctr->stamps[4], __builtin_amdgcn_s_memrealtime() - dbg_t0);
            if (part == 0u) atomicAdd(&gctr->stamps[5], __builtin_amdgcn_s_memrealtime() - dbg_t0);
        }
#endif
    }
    if (COUNT) {
        atomicAdd(&gctr->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->trace_nodes, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->trace_tris, (unsigned long long)lc.tris);
        if (ALPHA) {
            atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
            atomicAdd(&gctr->shaded_hits, (unsigned long long)lc.shaded);
            atomicAdd(&gctr->rng_draws, (unsigned long long)lc.shadow_rays);
        }
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&gctr->deferred_casts, (unsigned long long)n);
    }
}

// ---------------------------------------------------------------------------
// The casts k_wf_trace did not take: rays that would need more slack than PT_SLACK_MAX in the wavefront walker (a direction
// component below 8e-4; slop model in pt_integrator.h), on the fat-ray walker (FatRay, pt_integrator.h), whose candidate set
// does not depend on the ray's direction.  WF_EXACT_LANES lanes per cast: the pending segments (node, interval) of a cast sit
// on ONE stack in LDS; every round each lane takes one from the top - up to L nodes of the tree at once, their fetches in
// flight together -, an interior node puts back the children the fat ray touches, a leaf is tested by the lane that took it.
// The visiting order is free (the cast's hit is the minimum of a total order over all accepted hits), so nothing needs
// sorting; a segment that starts beyond the group's best hit is dropped.  With one lane per cast (first form) a launch took
// as long as its longest cast - ~1000 nodes one after the other: 2.1 ms for the 130 000 casts of config 3's bounce 1.
// The alpha walk (mod.rs:188-205) as in k_wf_trace's complete(), group-uniform as in k_wf_trace_wide.
// PRIMARY: the camera rays of the KD-tree-only pipeline (entry = work item of the chunk).  W.split_deferred: the pass of
// k_wf_shade over the queue is running meanwhile - the hit's word goes to the exact list's own plane (by list position), the
// rest to the chunk's plane at the queue index, whose word stays WF_HIT_PENDING.
// ---------------------------------------------------------------------------
#ifndef WF_EXACT_LANES
#define WF_EXACT_LANES 16u
#endif
#ifndef WF_EXACT_COOP
#define WF_EXACT_COOP 0   // 1: the lane-cooperative form below; 0: one lane per cast with its stack in LDS (k_wf_trace_exact1)
#endif
#define WF_EXACT_SOFT 192u   // above this many pending segments a group goes depth first, one segment per round ...
#define WF_EXACT_CAP 272u    // ... which adds at most one entry per level of the tree (PT_KD_STACK = 64) before it shrinks
template <bool COUNT, bool ALPHA, bool PRIMARY>
__global__ __launch_bounds__(256) void k_wf_trace_exact_coop(DevScene S, WfParams W, const uint32_t* __restrict__ tile_offsets,
                                                        const float4* __restrict__ queue, uint4* __restrict__ hits,
                                                        const uint4* __restrict__ rng_planes, uint32_t* __restrict__ draws,
                                                        uint32_t* __restrict__ exact_list, const WfCounters* __restrict__ ctr,
                                                        DevCounters* __restrict__ gctr) {
    constexpr uint32_t L = WF_EXACT_LANES, GROUPS = 256u / L;
    static_assert(WF_EXACT_SOFT + PT_KD_STACK + L <= WF_EXACT_CAP, "stack of the exact walker");
    __shared__ uint32_t st_node[GROUPS][WF_EXACT_CAP];
    __shared__ float st_t0[GROUPS][WF_EXACT_CAP], st_t1[GROUPS][WF_EXACT_CAP];
    const uint32_t n = min(ctr[W.bounce].exact_count, W.ecap);
    if (n == 0u) return;
    const uint32_t group = threadIdx.x / L, part = threadIdx.x & (L - 1u), lane = threadIdx.x & 63u;
    const unsigned long long group_mask = ((1ull << L) - 1ull) << (lane & ~(L - 1u));
    const unsigned long long below = (1ull << lane) - 1ull;
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    uint32_t n_alpha_draws = 0;
    const uint32_t passes = (n + gridDim.x * GROUPS - 1u) / (gridDim.x * GROUPS);   // (whole wavefronts stay in the loops together)
    for (uint32_t r = 0; r < passes; ++r) {
        const uint32_t e = (r * gridDim.x + blockIdx.x) * GROUPS + group;
        const bool have = e < n;
        const uint32_t idx = have ? exact_list[e] : 0u;
        f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 0.f);
        uint32_t item = idx, draw = 2u;   // (PRIMARY: the pixel jitter)
        if (have) {
            if (PRIMARY) {
                const uint2 sc = *(const uint2*)(rng_planes + idx);   // jittered screen position (k_wf_rng); valid: k_wf_trace checked
                primary_from_screen(S, __uint_as_float(sc.x), __uint_as_float(sc.y), o, d);
            } else {
                const float4* q = wf_ray_rec(queue, idx);
                const float4 q0 = q[0], q1 = q[1];
                o = mk3(q0.x, q0.y, q0.z);
                d = mk3(q0.w, q1.x, q1.y);
                item = __float_as_uint(q1.z);
                draw = __float_as_uint(q1.w) & 0xffffu;
            }
        }
        if (COUNT && have && part == 0u) lc.segments++;
        const float dlen = mag3(d);
        const float key_scale = dlen < 1.0f ? dlen : 1.0f;   // key >= t * min(1, |d|)
        FatRay F;
        float r0 = 0.f, r1 = 0.f;
        const bool in_scene = have && F.init(S, o, d, 0.f, r0, r1);
        float t_prev = -INFINITY;
        uint32_t ord_prev = 0u;
        RawHit kept, win;
        kept.pid = win.pid = 0xffffffffu;
        bool have_kept = false, hit = false, live = have;
        do {   // one turn per entry of the sorted hit list the alpha walk looks at (opaque scenes: one)
            RawHit best;
            best.key = INFINITY;
            best.ord = 0xffffffffu;
            best.pid = 0xffffffffu;
            best.u = best.v = 0.f;
            best.flags = 0u;
            uint32_t count = (live && in_scene) ? 1u : 0u;   // pending segments of the group's cast (the same in its L lanes)
            if (count && part == 0u) {
                st_node[group][0] = 0u;
                st_t0[group][0] = r0;
                st_t1[group][0] = r1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            while (wf_any(count != 0u)) {
                float gkey = best.key;   // the group's best distance so far: what every lane culls against
#pragma unroll
                for (uint32_t m = L / 2u; m >= 1u; m >>= 1) {
                    const float k2 = __shfl_xor(gkey, (int)m);
                    gkey = k2 < gkey ? k2 : gkey;
                }
                const uint32_t pops = count > WF_EXACT_SOFT ? 1u : (count < L ? count : L);
                const bool take = part < pops;
                uint32_t node = 0u;
                float t0 = 0.f, t1 = 0.f;
                if (take) {
                    node = st_node[group][count - 1u - part];
                    t0 = st_t0[group][count - 1u - part];
                    t1 = st_t1[group][count - 1u - part];
                }
                count -= pops;
                __builtin_amdgcn_wave_barrier();
                // (the segment's hits have keys >= t0 * min(1, |d|): beyond the best one it holds nothing of interest)
                uint32_t n_push = 0u, c_node[2] = {0u, 0u};
                float c_t0[2] = {0.f, 0.f}, c_t1[2] = {0.f, 0.f};
                if (take && !(t0 * key_scale > gkey + PT_EXIT_ABS)) {
                    const uint2 nd = S.kd_nodes[node];
                    if (COUNT) lc.nodes++;
                    const uint32_t axis = nd.y & 3u;
                    if (axis != 3u) {
                        const uint32_t lo = nd.y >> 2, hi = lo + 1u;
                        float b0, b1, a0, a1;
                        bool vb, va;
                        F.children(axis, __uint_as_float(nd.x), t0, t1, vb, b0, b1, va, a0, a1);
                        const bool below_first = b0 <= a0;   // the nearer child goes on top
                        if (vb && va) {
                            c_node[0] = below_first ? hi : lo;
                            c_t0[0] = below_first ? a0 : b0;
                            c_t1[0] = below_first ? a1 : b1;
                            c_node[1] = below_first ? lo : hi;
                            c_t0[1] = below_first ? b0 : a0;
                            c_t1[1] = below_first ? b1 : a1;
                            n_push = 2u;
                        } else if (vb || va) {
                            c_node[0] = vb ? lo : hi;
                            c_t0[0] = vb ? b0 : a0;
                            c_t1[0] = vb ? b1 : a1;
                            n_push = 1u;
                        }
                    } else {
                        const uint32_t n_refs = nd.y >> 2;
                        const float4* lp = S.leaf_prims + (size_t)nd.x * 3;
                        for (uint32_t i = 0; i < n_refs; ++i) {
                            float4 q0, q1, q2;
                            load_prim_record(lp + 3 * i, q0, q1, q2);
                            og_test_closest<COUNT>(o, d, q0, q1, q2, t_prev, ord_prev, best, lc);
                        }
                    }
                }
                // put the children back: exclusive prefix of n_push over the group's lanes
                const unsigned long long m1 = __ballot(n_push >= 1u) & group_mask, m2 = __ballot(n_push == 2u) & group_mask;
                const uint32_t off = (uint32_t)__popcll(m1 & below) + (uint32_t)__popcll(m2 & below);
                if (n_push >= 1u) {
                    st_node[group][count + off] = c_node[0];
                    st_t0[group][count + off] = c_t0[0];
                    st_t1[group][count + off] = c_t1[0];
                }
                if (n_push == 2u) {
                    st_node[group][count + off + 1u] = c_node[1];
                    st_t0[group][count + off + 1u] = c_t0[1];
                    st_t1[group][count + off + 1u] = c_t1[1];
                }
                count += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // the group's minimum of (key, ord): every lane takes the winner's record
            float kmin = best.key;
            uint32_t omin = best.ord;
#pragma unroll
            for (uint32_t m = L / 2u; m >= 1u; m >>= 1) {
                const float k2 = __shfl_xor(kmin, (int)m);
                const uint32_t o2 = __shfl_xor(omin, (int)m);
                if (key_less(k2, o2, kmin, omin)) {
                    kmin = k2;
                    omin = o2;
                }
            }
            const bool found = best.pid != 0xffffffffu && best.key == kmin && best.ord == omin;
            const unsigned long long winners = __ballot(found) & group_mask;
            const int src = winners ? __ffsll((long long)winners) - 1 : (int)lane;
            win.key = __shfl(best.key, src);
            win.ord = __shfl(best.ord, src);
            win.pid = __shfl(best.pid, src);
            win.u = __shfl(best.u, src);
            win.v = __shfl(best.v, src);
            win.flags = __shfl(best.flags, src);
            bool again = false;
            if (live) {
                hit = winners != 0ull;
                if (hit && !hit_passes_slab(S, o, d)) {   // kdtree-ray's box test: no hits at all
                    hit = false;
                    have_kept = false;
                }
                if (ALPHA && hit) {
                    const float opacity = hit_opacity(S, o, d, win);
                    if (COUNT && part == 0u) lc.shaded++;
                    bool stop = opacity >= 1.f;
                    if (!stop && opacity > 0.001f) {
                        WfRng fb;
                        fb.block = 0xffffffffu;
                        stop = wf_rng_draw(fb, W, tile_offsets, rng_planes, item, draw++) < opacity;
                        if (COUNT && part == 0u) n_alpha_draws++;
                    }
                    if (!stop) {   // skipped: remember it, the group looks for the next entry of the sorted list
                        kept = win;
                        have_kept = true;
                        t_prev = win.key;
                        ord_prev = win.ord;
                        if (COUNT && part == 0u) lc.restarts++;
                        again = true;   // (from the origin again: these are the rare rays, and "behind t_prev" needs no slack argument
                                        //  this way - the acceptance rule of the leaf tests is the only filter)
                    }
                }
                if (!again) {
                    if (ALPHA && !hit && have_kept) {   // every hit skipped: the last one is shaded
                        win = kept;
                        hit = true;
                    }
                    if (part == 0u) {
                        if (W.split_deferred) {
                            const uint4 rec = pack_hit(win, hit);
                            wf_exact_words(exact_list, W.ecap)[e] = rec.x;
                            if (hit) ((uint4*)((uint32_t*)hits + W.hcap))[idx] = make_uint4(rec.y, rec.z, rec.w, 0u);
                        } else {
                            wf_store_hit(hits, W.hcap, idx, win, hit);
                        }
                        if (ALPHA) draws[idx] = draw;
                    }
                    live = false;
                }
            }
            if (!wf_any(again)) break;
        } while (true);
    }
    if (COUNT) {
        atomicAdd(&gctr->segments, (unsigned long long)lc.segments);
        atomicAdd(&gctr->exact_casts, (unsigned long long)lc.segments);
        atomicAdd(&gctr->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->trace_nodes, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->trace_tris, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
        if (ALPHA) atomicAdd(&gctr->shaded_hits, (unsigned long long)lc.shaded);
        if (ALPHA) atomicAdd(&gctr->rng_draws, (unsigned long long)n_alpha_draws);
    }
}

// One lane per cast: 64 walks per wavefront in flight (the cooperative form above has four, and most of a walk is the descent
// to its first leaf, one segment wide).  The stack of pending segments - (node, interval), 12 bytes - in LDS, [entry][thread]
// like k_wf_trace's, the rare deep part in scratch: with the whole stack in scratch (first form) every push and pop of a
// wavefront whose lanes are at different depths was 64 separate cache lines.
#define WF_EXACT_LDS_STACK 16
#define WF_EXACT_THREADS 128
template <bool COUNT, bool ALPHA, bool PRIMARY>
__global__ __launch_bounds__(WF_EXACT_THREADS) void k_wf_trace_exact(DevScene S, WfParams W, const uint32_t* __restrict__ tile_offsets,
                                                        const float4* __restrict__ queue, uint4* __restrict__ hits,
                                                        const uint4* __restrict__ rng_planes, uint32_t* __restrict__ draws,
                                                        uint32_t* __restrict__ exact_list, const WfCounters* __restrict__ ctr,
                                                        DevCounters* __restrict__ gctr) {
    __shared__ uint32_t s_node[WF_EXACT_LDS_STACK][WF_EXACT_THREADS];
    __shared__ float s_t0[WF_EXACT_LDS_STACK][WF_EXACT_THREADS], s_t1[WF_EXACT_LDS_STACK][WF_EXACT_THREADS];
    uint32_t ov_node[PT_KD_STACK - WF_EXACT_LDS_STACK];
    float ov_t0[PT_KD_STACK - WF_EXACT_LDS_STACK], ov_t1[PT_KD_STACK - WF_EXACT_LDS_STACK];
    const uint32_t n = min(ctr[W.bounce].exact_count, W.ecap), tid = threadIdx.x;
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    uint32_t n_alpha_draws = 0;
    for (uint32_t e = blockIdx.x * WF_EXACT_THREADS + tid; e < n; e += gridDim.x * WF_EXACT_THREADS) {
        const uint32_t idx = exact_list[e];
        f3 o, d;
        uint32_t item = idx, draw = 2u;   // (PRIMARY: the pixel jitter)
        if (PRIMARY) {
            const uint2 sc = *(const uint2*)(rng_planes + idx);   // jittered screen position (k_wf_rng); valid: k_wf_trace checked
            primary_from_screen(S, __uint_as_float(sc.x), __uint_as_float(sc.y), o, d);
        } else {
            const float4* q = wf_ray_rec(queue, idx);
            const float4 q0 = q[0], q1 = q[1];
            o = mk3(q0.x, q0.y, q0.z);
            d = mk3(q0.w, q1.x, q1.y);
            item = __float_as_uint(q1.z);
            draw = __float_as_uint(q1.w) & 0xffffu;
        }
        if (COUNT) lc.segments++;
        const float dlen = mag3(d);
        const float key_scale = dlen < 1.0f ? dlen : 1.0f;   // key >= t * min(1, |d|)
        FatRay F;
        float r0 = 0.f, r1 = 0.f;
        const bool in_scene = F.init(S, o, d, 0.f, r0, r1);
        // The alpha walk (mod.rs:188-205) looks at the sorted hit list entry by entry.  The FIRST walk is a closest-hit walk (it
        // stops at the first leaf with a hit: most first entries are opaque); a ray that skips its first entry walks again and
        // collects the KH nearest entries behind (t_prev, ord_prev) in one go - such a walk ends only where KH entries lie
        // before the segment - and the list is walked from those; a ray that skips all KH walks once more.  (A walk per
        // entry made a translucent cast of config 5 five walks long; KH entries from the first walk on made every cast a walk to
        // the far end of the scene.  Opaque scenes: KH = 1, the closest hit.)
        constexpr int KH = ALPHA ? 4 : 1;
        RawHit cand[KH], best, kept;
        bool have_kept = false, hit = false, first_walk = true;
        float t_prev = -INFINITY;
        uint32_t ord_prev = 0u;
        best.pid = 0xffffffffu;
        while (true) {
            int nc = 0;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                cand[k].key = INFINITY;
                cand[k].ord = 0xffffffffu;
                cand[k].pid = 0xffffffffu;
            }
            // a hit (key, ord) behind (t_prev, ord_prev): into the sorted candidates if it is among the KH nearest (a primitive is
            // referenced from several leaves: the same entry again is not a new one)
            auto offer = [&](float key, uint32_t ord, uint32_t pid, float u, float v, uint32_t flags) {
                if (!(key == key) || !key_less(t_prev, ord_prev, key, ord) || !key_less(key, ord, cand[KH - 1].key, cand[KH - 1].ord)) return;
#pragma unroll
                for (int k = 0; k < KH; ++k)
                    if (cand[k].key == key && cand[k].ord == ord) return;
                RawHit in;
                in.key = key;
                in.ord = ord;
                in.pid = pid;
                in.u = u;
                in.v = v;
                in.flags = flags;
#pragma unroll
                for (int k = 0; k < KH; ++k) {   // insertion: `in` moves down to its place, the last candidate falls out
                    if (key_less(in.key, in.ord, cand[k].key, cand[k].ord)) {
                        const RawHit t = cand[k];
                        cand[k] = in;
                        in = t;
                    }
                }
                if (nc < KH) ++nc;
            };
            if (in_scene) {
                int sp = 0;
                uint32_t node = 0u;
                float t0 = r0, t1 = r1;
                while (true) {
                    const uint2 nd = S.kd_nodes[node];
                    if (COUNT) lc.nodes++;
                    const uint32_t axis = nd.y & 3u;
                    bool descend = false;
                    if (axis != 3u) {
                        const uint32_t lo = nd.y >> 2, hi = lo + 1u;
                        float b0, b1, a0, a1;
                        bool vb, va;
                        F.children(axis, __uint_as_float(nd.x), t0, t1, vb, b0, b1, va, a0, a1);
                        const bool below_first = b0 <= a0;
                        if (vb && va) {   // the farther child waits
                            const uint32_t fn = below_first ? hi : lo;
                            const float f0 = below_first ? a0 : b0, f1 = below_first ? a1 : b1;
                            if (sp < WF_EXACT_LDS_STACK) {
                                s_node[sp][tid] = fn;
                                s_t0[sp][tid] = f0;
                                s_t1[sp][tid] = f1;
                            } else {
                                ov_node[sp - WF_EXACT_LDS_STACK] = fn;
                                ov_t0[sp - WF_EXACT_LDS_STACK] = f0;
                                ov_t1[sp - WF_EXACT_LDS_STACK] = f1;
                            }
                            ++sp;
                            node = below_first ? lo : hi;
                            t0 = below_first ? b0 : a0;
                            t1 = below_first ? b1 : a1;
                            descend = true;
                        } else if (vb || va) {
                            node = vb ? lo : hi;
                            t0 = vb ? b0 : a0;
                            t1 = vb ? b1 : a1;
                            descend = true;
                        }
                    } else {
                        const uint32_t n_refs = nd.y >> 2;
                        const float4* lp = S.leaf_prims + (size_t)nd.x * 3;
                        for (uint32_t i = 0; i < n_refs; ++i) {
                            float4 q0, q1, q2;
                            load_prim_record(lp + 3 * i, q0, q1, q2);
                            const uint32_t pid = __float_as_uint(q0.w);
                            if (COUNT) lc.tris++;
                            if (!(pid & PT_PRIM_SPHERE)) {
                                float dist, u, v;
                                bool bf;
                                if (isect_triangle(o, d, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q1.w, q2.x, q2.y), dist, u, v, bf))
                                    offer(dist, PT_PRIM_INDEX(pid) * 2u, pid, u, v, bf ? 1u : 0u);
                            } else {
                                float ts[2], key[2];
                                bool ex[2];
                                const int nh = isect_sphere(o, d, mk3(q0.x, q0.y, q0.z), q1.x, ts, key, ex);
                                for (int k = 0; k < nh; ++k) offer(key[k], PT_PRIM_INDEX(pid) * 2u + (ex[k] ? 1u : 0u), pid, ts[k], 0.f, 2u | (ex[k] ? 4u : 0u));
                            }
                        }
                    }
                    if (descend) continue;
                    bool more = false;
                    while (sp > 0) {
                        --sp;
                        if (sp < WF_EXACT_LDS_STACK) {
                            node = s_node[sp][tid];
                            t0 = s_t0[sp][tid];
                            t1 = s_t1[sp][tid];
                        } else {
                            node = ov_node[sp - WF_EXACT_LDS_STACK];
                            t0 = ov_t0[sp - WF_EXACT_LDS_STACK];
                            t1 = ov_t1[sp - WF_EXACT_LDS_STACK];
                        }
                        // (the segment's hits have keys >= t0 * min(1, |d|): beyond the last candidate it holds nothing of interest)
                        if (!(t0 * key_scale > (first_walk ? cand[0].key : cand[KH - 1].key) + PT_EXIT_ABS)) {
                            more = true;
                            break;
                        }
                    }
                    if (!more) break;
                }
            }
            // kdtree-ray's box test: a ray it rejects has no hits at all
            if (nc > 0 && !hit_passes_slab(S, o, d)) {
                nc = 0;
                have_kept = false;
            }
            hit = false;
            bool walk_again = false;
            // (the first walk stopped behind its nearest entry: the others it happened to see are not known to be the next ones)
            const int n_known = (first_walk && nc > 1) ? 1 : nc, last_known = first_walk ? 0 : KH - 1;
            first_walk = false;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                if (k < n_known && !hit) {
                    best = cand[k];
                    bool stop = true;
                    if (ALPHA) {
                        const float opacity = hit_opacity(S, o, d, best);
                        if (COUNT) lc.shaded++;
                        stop = opacity >= 1.f;
                        if (!stop && opacity > 0.001f) {
                            WfRng fb;
                            fb.block = 0xffffffffu;
                            stop = wf_rng_draw(fb, W, tile_offsets, rng_planes, item, draw++) < opacity;
                            if (COUNT) n_alpha_draws++;
                        }
                    }
                    if (stop) {
                        hit = true;
                    } else {   // skipped: remember it, look at the next entry of the sorted list
                        kept = best;
                        have_kept = true;
                        t_prev = best.key;
                        ord_prev = best.ord;
                        if (COUNT) lc.restarts++;
                        if (k == last_known) walk_again = true;   // (every known entry skipped: there may be more behind them)
                    }
                }
            }
            if (!walk_again) break;
        }
        if (ALPHA && !hit && have_kept) {   // every hit skipped: the last one is shaded
            best = kept;
            hit = true;
        }
        if (W.split_deferred) {
            const uint4 rec = pack_hit(best, hit);
            wf_exact_words(exact_list, W.ecap)[e] = rec.x;
            if (hit) ((uint4*)((uint32_t*)hits + W.hcap))[idx] = make_uint4(rec.y, rec.z, rec.w, 0u);
        } else {
            wf_store_hit(hits, W.hcap, idx, best, hit);
        }
        if (ALPHA) draws[idx] = draw;
    }
    if (COUNT) {
        atomicAdd(&gctr->segments, (unsigned long long)lc.segments);
        atomicAdd(&gctr->exact_casts, (unsigned long long)lc.segments);
        atomicAdd(&gctr->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->trace_nodes, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->trace_tris, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
        if (ALPHA) atomicAdd(&gctr->shaded_hits, (unsigned long long)lc.shaded);
        if (ALPHA) atomicAdd(&gctr->rng_draws, (unsigned long long)n_alpha_draws);
    }
}

// A light whose term (thr ⊙ eval_direct) is exactly (0,0,0) - it lies below the shading horizon and the
// surface is not emissive - adds 0 ⊙ radiance to the colour (mod.rs:255-261), i.e. nothing, whatever
// the shadow ray finds, as long as the radiance is finite: colour finite and, for a point light, the
// surface not within 1e-3 of it (colour / (4 pi d^2) < 1e30 / 1.2e-5).  Those shadow rays are not cast.
PT_D bool wf_light_is_moot(const DevLight& L, f3 term, f3 surface_pos) {
    if (!(term.x == 0.f && term.y == 0.f && term.z == 0.f) || !L.tame) return false;
    return L.kind != PT_LIGHT_POINT || mag3(surface_pos - ld3(L.vec)) > 1e-3f;
}

// ---------------------------------------------------------------------------
// shade: compute_radiance without the light visibility (mod.rs:230-278)
// ---------------------------------------------------------------------------
#ifndef WF_SHADE_THREADS
// Workgroup = compaction unit (one atomic per queue and workgroup) and barrier domain.  MI355X, config 3,
// shade stage per 64 spp: 1024 threads 8.45 ms (one workgroup per CU: its three barriers stall the whole
// CU on the slowest wave), 512: 7.33, 256: 6.9, 128: 10.2 (the queue counters saturate: ~88 returning
// atomics per microsecond and word).  Register budget WF_SHADE_WAVES in every case.
#define WF_SHADE_THREADS 256
#endif
#ifndef WF_SHADE_WAVES
#define WF_SHADE_WAVES 4      // waves per SIMD the kernel is compiled for (register budget 512 / 4)
#endif
#ifndef WF_SHADE_AGGREGATE
#define WF_SHADE_AGGREGATE 1  // bounces >= 1: retire the misses first, shade the hits 256 at a time (see the kernel)
#endif
#ifndef WF_SHADE_GRID_WAVES
// ... for the variants that cast through origin grids inline: 3 (170 registers).  At 4 the fused bounce-0 kernel keeps
// 132 B of its state per lane in scratch: 18.66 ms against 17.09 ms at 3 (and 21.3 ms at 2) on config 3.
#define WF_SHADE_GRID_WAVES 3
#endif
#ifndef WF_SHADE_GRID_WAVES_ALPHA
#define WF_SHADE_GRID_WAVES_ALPHA WF_SHADE_GRID_WAVES   // (translucent scenes, 88 B of scratch at 3: 2 waves measured worse, item 17)
#endif
// GRID (origin grids, pt_grid.h): 0 - none: direct light goes through the shadow queue and k_wf_shadow / k_og_shadow;
// 1 - every light is a point light with a grid: get_light_info (mod.rs:281-333) is evaluated HERE, light after
//     light, so a surface costs no shadow record, no contrib entries and no colour patch (190 B of queue traffic
//     per shaded hit); only surfaces whose normal is too long for the grids' margin still take the queue;
// 2 - (bounce 0) 1 + the camera ray is cast HERE through the camera grid: no hits[] round trip, and the 44 % of
//     the samples that leave into the background never reach a second kernel;
// 3 - (bounce 0) 2 + the item's ChaCha12 block is computed HERE instead of by k_wf_rng: the ~800 integer
//     instructions per item run underneath the memory latency of the casts, and only the paths that go on write
//     the words of bounces 1 and 2 (plane 1 of the RNG planes; a path has made at least four draws by then).
// GRIDX = GRID + 4 * DIRL; DIRL: some light is directional (orthographic grid branch of og_light_radiance compiled in).
template <bool ALPHA, bool COUNT, bool PRIMARY, int GRIDX>
__global__ __launch_bounds__(WF_SHADE_THREADS, GRIDX ? (ALPHA ? WF_SHADE_GRID_WAVES_ALPHA : WF_SHADE_GRID_WAVES) : WF_SHADE_WAVES) void k_wf_shade(DevScene S, WfParams W,
                                                  const uint32_t* __restrict__ tile_offsets,
                                                  const float4* __restrict__ queue_in, const uint4* __restrict__ hits,
                                                  const uint4* rng_planes,
                                                  const uint32_t* __restrict__ draws,
                                                  float4* __restrict__ queue_out, float4* __restrict__ shadow_q,
                                                  float4* __restrict__ contrib, float* __restrict__ staging,
                                                  const uint32_t* __restrict__ index_list,
                                                  const uint32_t* __restrict__ exact_list, const uint4* __restrict__ chunk_hits,
                                                  uint32_t* __restrict__ exact_next,
                                                  const uint32_t* __restrict__ block_empty,
                                                  WfCounters* __restrict__ ctr, DevCounters* __restrict__ gctr) {
    // index_list (bounces >= 1): null - the whole queue, entries marked WF_HIT_PENDING left out; else the entries to shade:
    // the hand-over list of k_wf_trace, whose casts k_wf_trace_wide has finished by now - `hits` is then that list's
    // own plane (W.list_cap records, by list position).  The pass over the queue runs WHILE k_wf_trace_wide walks
    // those few long casts (WF_WIDE_LANES = 32 lanes each, a launch bound by its longest cast).  Behind that list, in the same launch, the
    // exact list (k_wf_trace_exact, pt_wavefront.h wf_exact_words): word by list position, the rest of the hit in the chunk's
    // own plane (chunk_hits) at the queue index.  exact_next: the exact list of the NEXT bounce - a survivor whose new ray the
    // wavefront walker will not take (slack_is_capped) is listed here, where the ray is made, so that k_wf_trace_exact can
    // walk it while k_wf_trace is busy with the rest of the queue.
    constexpr int GRID = GRIDX & 3;
    constexpr bool DIRL = GRIDX >= 4;
    static_assert(GRIDX != 4, "DIRL needs a grid mode");
    uint4* rng_planes_out = const_cast<uint4*>(rng_planes);   // GRID == 3 writes plane 1 (nobody reads it before bounce 1)
    static_assert(GRID < 2 || PRIMARY, "the camera grid serves bounce 0");
    const bool list_pass = !PRIMARY && index_list != nullptr;
    const uint32_t n_def = list_pass ? ctr[W.bounce].deferred_count : 0u;
    const uint32_t n = PRIMARY ? W.n_items : list_pass ? n_def + (exact_list ? min(ctr[W.bounce].exact_count, W.ecap) : 0u) : min(ctr[W.bounce].queue_count, W.qcap_in);
    // entry e of this launch: its queue record, the word of its hit, the hit
    auto entry_index = [&](uint32_t e) -> uint32_t { return !list_pass ? e : e < n_def ? index_list[e] : exact_list[e - n_def]; };
    auto entry_word = [&](uint32_t e) -> uint32_t {
        return (!list_pass || e < n_def) ? wf_hit_word(hits, e) : wf_exact_words(exact_list, W.ecap)[e - n_def];
    };
    auto entry_hit = [&](uint32_t e, uint32_t i, RawHit& h) -> bool {
        if (!list_pass) return wf_load_hit(hits, W.hcap, e, h);
        if (e < n_def) return wf_load_hit(hits, W.list_cap, e, h);
        const uint32_t x = wf_exact_words(exact_list, W.ecap)[e - n_def];
        if (x == 0xffffffffu) return false;
        const uint4 k = ((const uint4*)((const uint32_t*)chunk_hits + W.hcap))[i];
        return unpack_hit(make_uint4(x, k.x, k.y, k.z), h);
    };
    uint32_t n_draws = 0, n_new = 0, n_moot = 0, n_hits = 0, n_cam_tris = 0, n_masked = 0;
    LocalCtr lc = {0, 0, 0, 0, 0, 0};   // (GRID: casts made here)
    __shared__ uint32_t sh_cnt[2][WF_SHADE_THREADS / 64];
    __shared__ uint32_t sh_base[2];
    __shared__ uint32_t sh_oct[8][WF_SHADE_THREADS / 64], sh_oct_off[8][WF_SHADE_THREADS / 64];
    const uint32_t wave = threadIdx.x >> 6;
    // (Bounce 0 needs no hit aggregation like the later bounces' below: a wavefront is one 8x8 pixel block of one sample,
    // so its camera rays hit or miss together - collecting the hits of 256 items in LDS and shading them 256 at a time left
    // the instruction count and the 52 active lanes per instruction unchanged, profiles/r02_experiments.txt item 13.)
    // One workgroup-wide step: thread t shades queue entry i (live = it has one).  Every thread of the workgroup
    // calls this together: the compaction at the end has barriers.
    auto shade_one = [&](const uint32_t e, bool live) {   // e: position in the queue / in the lists
    const uint32_t i = (list_pass && live) ? entry_index(e) : e;
    f3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), thr = mk3(0, 0, 0), color = mk3(0, 0, 0);
    uint32_t item = i, draw = 0, out_slot = 0;
    RawHit h;
    bool hit = false;
    WfRng rng;
    rng.block = 0xffffffffu;
    uint32_t word2 = 0, word3 = 0;                      // GRID == 3: words 2 and 3 of the item's ChaCha block
    uint4 later_words = make_uint4(0u, 0u, 0u, 0u);     //            words 4-7 (bounces 1 and 2; the alpha walk of bounce 0)
    // rng.gen::<f32>() number idx (>= 2) of the path at bounce 0, GRID == 3: from the block in registers; beyond
    // word 7 (more than three alpha draws) the block is derived again by wf_rng_draw
    auto draw_b0 = [&](uint32_t idx) -> float {
        if (idx >= WF_RNG_STAGED) return wf_rng_draw(rng, W, tile_offsets, rng_planes, item, idx);
        uint32_t word = word2;
        word = idx == 3u ? word3 : word;
        word = idx == 4u ? later_words.x : word;
        word = idx == 5u ? later_words.y : word;
        word = idx == 6u ? later_words.z : word;
        word = idx == 7u ? later_words.w : word;
        return wf_rng_float(word);
    };
    if (PRIMARY && live) {  // entry i is work item i: the initial path state, built in place
        ItemRef it = decode_item(W.P, tile_offsets, W.item_base + i);
        if (!it.valid) {
            live = false;
        } else {
            thr = mk3(1.f, 1.f, 1.f);
            color = mk3(0.f, 0.f, 0.f);
            out_slot = (it.sample - 1u - W.P.sample_begin) * W.P.n_local + it.out_index;
            if (GRID >= 2) {   // ray_cast + alpha walk of the camera ray (mod.rs:182-205) through the camera grid
                if (GRID == 3) {   // StdRng::seed_from_u64(sample + i * samples), jitter x then y (mod.rs:110-120)
                    uint32_t w[16];
                    pt_chacha12_block((uint64_t)it.sample + (uint64_t)it.global_index * (uint64_t)W.P.samples, 0u, w);
                    float sx, sy;
                    primary_screen(S, it.x, it.y, W.P.width, W.P.height, wf_rng_float(w[0]), wf_rng_float(w[1]), sx, sy);
                    primary_from_screen(S, sx, sy, o, d);
                    word2 = w[2];
                    word3 = w[3];
                    later_words = make_uint4(w[4], w[5], w[6], w[7]);
                } else {
                    const uint2 sc = *(const uint2*)(rng_planes + i);  // jittered screen position (k_wf_rng)
                    primary_from_screen(S, __uint_as_float(sc.x), __uint_as_float(sc.y), o, d);
                }
                const uint32_t cell = og_cell(S.cam_grid, d);
                const float dlen = mag3(d);
                const float kmax = (dlen > 1.0f ? dlen : 1.0f) * 1.00002f;
                draw = 2u;   // the pixel jitter
                if (COUNT) lc.segments++;
                const uint32_t tris_before = lc.tris;
                hit = og_next_hit<COUNT>(S, S.cam_grid, cell, o, d, kmax, INFINITY, -INFINITY, 0u, h, lc);
                if (ALPHA) {
                    RawHit kept = h;
                    bool have_kept = false;
                    while (hit) {
                        const float opacity = hit_opacity(S, o, d, h);
                        if (COUNT) lc.shaded++;
                        bool stop = opacity >= 1.f;
                        if (!stop && opacity > 0.001f)
                            stop = (GRID == 3 ? draw_b0(draw++) : wf_rng_draw(rng, W, tile_offsets, rng_planes, item, draw++)) < opacity;
                        if (stop) break;
                        kept = h;   // skipped: remember it, look for the next entry of the list
                        have_kept = true;
                        if (COUNT) lc.restarts++;
                        hit = og_next_hit<COUNT>(S, S.cam_grid, cell, o, d, kmax, INFINITY, kept.key, kept.ord, h, lc);
                    }
                    if (!hit && have_kept) {   // every hit skipped: the last one is shaded
                        h = kept;
                        hit = true;
                    }
                }
                if (COUNT) n_cam_tris += lc.tris - tris_before;
            } else {
                draw = ALPHA ? draws[i] : 2u;   // 2 = the pixel jitter (+ the draws of the alpha walk)
                hit = wf_load_hit(hits, W.hcap, i, h);
                if (hit) {  // (a missed cast only adds the background: no ray, no normalisation)
                    const uint2 sc = *(const uint2*)(rng_planes + i);  // jittered screen position (k_wf_rng)
                    primary_from_screen(S, __uint_as_float(sc.x), __uint_as_float(sc.y), o, d);
                }
            }
            if (COUNT) n_new++;
        }
    }
    if (!PRIMARY && live) {
        const float4* qr = wf_ray_rec(queue_in, i);
        const float4* qp = wf_path_rec(queue_in, W.qcap_in, i);
        float4 q0 = qr[0], q1 = qr[1], q2 = qp[0], q3 = qp[1];
        o = mk3(q0.x, q0.y, q0.z);
        d = mk3(q0.w, q1.x, q1.y);
        thr = mk3(q2.x, q2.y, q2.z);
        color = mk3(q3.x, q3.y, q3.z);
        item = __float_as_uint(q1.z);
        draw = ALPHA ? draws[i] : (__float_as_uint(q1.w) & 0xffffu);
        out_slot = __float_as_uint(q2.w);
        hit = entry_hit(e, i, h);
        if (out_slot == 0xffffffffu) live = false;  // (records of items outside the image; none since bounce 0 is fused)
    }
    const uint32_t bounce = PRIMARY ? 0u : W.bounce, bounces = W.P.bounces;   // (PRIMARY: no Russian roulette code at all)
    bool to_shadow = false, survive = false;
    f3 term0 = mk3(0.f, 0.f, 0.f);
    Surface surf;
    f3 next_o = o, next_d = d, next_thr = thr;
    if (live && !hit) {  // background (mod.rs:184-186): the path ends here
        color = color + mul_ew(thr, ld3(S.background));
        float* out = staging + (size_t)out_slot * 3;
        out[0] = color.x;
        out[1] = color.y;
        out[2] = color.z;
    }
    Brdf brdf;
    f3 normal = mk3(0, 0, 0), view = mk3(0, 0, 0);
    if (live && hit) {
        if (COUNT) n_hits++;
        make_surface(S, o, d, h, surf);
        MatSample ms;
        material_sample(S, surf.model, surf.sphere, surf.uv, ms);
        normal = shading_normal(S, surf);
        if (COUNT && !ALPHA) atomicAdd(&gctr->shaded_hits, 1ull);
        view = -1.f * d;
        ct_init(brdf, ms);
        color = color + mul_ew(thr, ms.emissive);
        // (thr ⊙ eval_direct) per light.  GRID: the light's visibility is looked up here and the light added at
        // once, in light order (mod.rs:248-262).  Otherwise (and for a normal too long for the grids' margin) the
        // visibility factor is applied by the shadow kernel: the first light's term stays in registers, the others
        // are recomputed when the record is written.
        const bool inline_lights = GRID != 0 && dot3(surf.normal, surf.normal) <= S.light_grid_max_normal2;
        for (uint32_t li = 0; li < S.n_lights; ++li) {
            const DevLight& L = S.lights[li];
            f3 ldir = L.kind == PT_LIGHT_POINT ? normalize3(surf.pos - ld3(L.vec)) : ld3(L.vec);
            f3 c = mul_ew(thr, ct_eval_direct(brdf, normal, view, -1.f * ldir));
            if (li == 0) term0 = c;
            const bool moot = wf_light_is_moot(L, c, surf.pos);
            if (GRID != 0 && inline_lights) {
                if (moot) {
                    if (COUNT) n_moot++;
                } else {
                    const f3 rad = og_light_radiance<ALPHA, COUNT, DIRL>(S, li, surf.pos, surf.normal, surf.uv, surf.sphere, lc);
                    if (!(rad.x == 0.f && rad.y == 0.f && rad.z == 0.f)) color = color + mul_ew(c, rad);
                }
            } else if (!moot) {
                to_shadow = true;
            }
        }
        if (COUNT && !inline_lights && !to_shadow) n_moot += S.n_lights;
        bool ended = false;
        if (bounce < bounces) {
            next_o = surf.pos + surf.normal * 0.00001f;
            float r1, r2;
            if (GRID == 3 && ALPHA) {
                r1 = draw_b0(draw++);
                r2 = draw_b0(draw++);
            } else if (GRID == 3) {
                r1 = wf_rng_float(word2);
                r2 = wf_rng_float(word3);
                draw += 2u;
            } else if (PRIMARY && !ALPHA) {   // draws 2 and 3 of the item: staged next to the screen position (k_wf_rng)
                const uint2 w23 = *(const uint2*)((const uint32_t*)(rng_planes + item) + 2);
                r1 = wf_rng_float(w23.x);
                r2 = wf_rng_float(w23.y);
                draw += 2u;
            } else {
                r1 = wf_rng_draw(rng, W, tile_offsets, rng_planes, item, draw++);
                r2 = wf_rng_draw(rng, W, tile_offsets, rng_planes, item, draw++);
            }
            next_d = ct_sample(brdf, normal, view, r1, r2);
            f3 wgt = ct_eval_indirect(brdf, normal, view, next_d) / 1.0f;
            next_thr = mul_ew(thr, wgt);
        }
        if (dot3(next_thr, next_thr) < 0.00001f) ended = true;
        if (!ended && bounce > 3) {
            float p = max_rs(max_rs(next_thr.x, next_thr.y), next_thr.z);
            next_thr = next_thr * (1.f / p);
            if (wf_rng_draw(rng, W, tile_offsets, rng_planes, item, draw++) > p) ended = true;
        }
        survive = !ended && bounce + 1 <= bounces;
        // Escape mask of the primitive the ray leaves (pt_escape.h): a clear bit proves that the next ray_cast is empty, i.e.
        // that the path ends with `colour + throughput x background` (mod.rs:184-186) - no record, no cast.  The background
        // term must be added AFTER this bounce's lights (mod.rs:248-262 come first): here if the lights are added here
        // (inline) or none can contribute; through the shadow queue only if the term is exactly zero (a black background).
        if (survive && S.escape != nullptr && !surf.sphere && escape_proves_miss(S, PT_PRIM_INDEX(h.pid), next_o, next_d)) {
            const f3 bg = mul_ew(next_thr, ld3(S.background));
            const bool zero = bg.x == 0.f && bg.y == 0.f && bg.z == 0.f;
            if (zero || !to_shadow) {
                if (!zero) color = color + bg;
                survive = false;
                if (COUNT) n_masked++;
            }
        }
    }
    // ---- compaction: survivors -> queue[b+1], surface hits -> shadow queue.  One atomic per
    // workgroup and queue (wave ballots -> LDS -> one lane), not one per wavefront: the counter
    // word would otherwise cap the kernel at ~88 M wave-atomics per second.
    // Coherence sorting (W.sort_octants): the survivors of a workgroup step - paths that started in the same few
    // 8x8 pixel blocks - are placed in the workgroup's slice of the queue by the OCTANT of their new direction
    // (ballot per octant, mbcnt rank, prefix over octants and waves in LDS), so that the 64 consecutive records a
    // wavefront of the next k_wf_trace launch fetches walk the tree in the same child order.  The position of a
    // record in a queue is free (results are keyed by out_slot), so no bit of the image changes.
    unsigned long long m_next = __ballot(survive), m_sh = __ballot(to_shadow);
    uint32_t oct = 0, oct_rank = 0;
    if (W.sort_octants & 1u) {
        oct = (next_d.x < 0.f ? 1u : 0u) | (next_d.y < 0.f ? 2u : 0u) | (next_d.z < 0.f ? 4u : 0u);
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) {
            const unsigned long long m = __ballot(survive && oct == k);
            if ((threadIdx.x & 63u) == 0) sh_oct[k][wave] = (uint32_t)__popcll(m);
            if (oct == k) oct_rank = wf_lane_rank(m);
        }
    }
    if ((threadIdx.x & 63u) == 0) {
        sh_cnt[0][wave] = (uint32_t)__popcll(m_next);
        sh_cnt[1][wave] = (uint32_t)__popcll(m_sh);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t total = 0;
        for (uint32_t k = 0; k < WF_SHADE_THREADS / 64; ++k) total += sh_cnt[threadIdx.x][k];
        uint32_t* counter = threadIdx.x == 0 ? &ctr[bounce + 1].queue_count : &ctr[bounce].shadow_count;
        sh_base[threadIdx.x] = total ? atomicAdd(counter, total) : 0u;
    } else if ((W.sort_octants & 1u) && threadIdx.x >= 64u && threadIdx.x < 64u + 8u * (WF_SHADE_THREADS / 64)) {
        // exclusive prefix of the (octant, wave) counts, octant-major
        const uint32_t e = threadIdx.x - 64u;
        uint32_t before = 0;
        for (uint32_t j = 0; j < e; ++j) before += sh_oct[j / (WF_SHADE_THREADS / 64)][j % (WF_SHADE_THREADS / 64)];
        sh_oct_off[e / (WF_SHADE_THREADS / 64)][e % (WF_SHADE_THREADS / 64)] = before;
    }
    __syncthreads();
    uint32_t next_idx = sh_base[0] + wf_lane_rank(m_next), sh_idx = sh_base[1] + wf_lane_rank(m_sh);
    for (uint32_t k = 0; k < wave; ++k) {
        next_idx += sh_cnt[0][k];
        sh_idx += sh_cnt[1][k];
    }
    if (W.sort_octants & 1u) next_idx = sh_base[0] + sh_oct_off[oct][wave] + oct_rank;
    __syncthreads();  // sh_cnt / sh_base are rewritten by the next step
    // (the queues are sized by the records this very frame is known to produce, pt_gpu.hip FramePlan: a record beyond its
    // queue's end means that knowledge was wrong - nothing is written, the frame is flagged and rendered again with room)
    if ((survive && next_idx >= W.qcap_out) || (to_shadow && sh_idx >= W.scap)) {
        ctr[0].overflow = 1u;
        survive = false;
        to_shadow = false;
    }
    if (GRID == 3 && survive) rng_planes_out[(size_t)(1u - W.rng_first_plane) * W.cap + item] = later_words;   // draws 4-7 of the path
    if (W.exact_shade_lists) {   // (wave-uniform; 0.24 % of random directions)
        const bool listed = survive && slack_is_capped(__builtin_amdgcn_rcpf(next_d.x), __builtin_amdgcn_rcpf(next_d.y), __builtin_amdgcn_rcpf(next_d.z));
        if (wf_any(listed)) {
            const uint32_t slot = wf_reserve(&ctr[bounce + 1].exact_count, listed);
            if (listed && slot < W.ecap) exact_next[slot] = next_idx;
            else if (listed) ctr[0].overflow = 1u;
        }
    }
    if (survive) {
        float4* qr = wf_ray_rec(queue_out, next_idx);
        float4* qp = wf_path_rec(queue_out, W.qcap_out, next_idx);
        qr[0] = make_float4(next_o.x, next_o.y, next_o.z, next_d.x);
        qr[1] = make_float4(next_d.y, next_d.z, __uint_as_float(item), __uint_as_float((draw & 0xffffu) | ((bounce + 1) << 16)));
        qp[0] = make_float4(next_thr.x, next_thr.y, next_thr.z, __uint_as_float(out_slot));
        qp[1] = make_float4(color.x, color.y, color.z, 0.f);  // colour is patched by the shadow kernels
        if (W.use_entry) wf_entry_plane(queue_out, W.qcap_out)[next_idx] = S.prim_entry[PT_PRIM_INDEX(h.pid)];
    }
    if (to_shadow) {
        float4* sq = shadow_q + (size_t)sh_idx * 4;
        sq[0] = make_float4(surf.pos.x, surf.pos.y, surf.pos.z, surf.normal.x);
        sq[1] = make_float4(surf.normal.y, surf.normal.z, surf.uv.x, surf.uv.y);
        sq[2] = make_float4(color.x, color.y, color.z, __uint_as_float(survive ? next_idx : 0xffffffffu));
        sq[3] = make_float4(__uint_as_float(out_slot), __uint_as_float(surf.sphere ? WF_FLAG_SPHERE : 0u), 0.f, 0.f);
        contrib[sh_idx] = make_float4(term0.x, term0.y, term0.z, 0.f);
        for (uint32_t li = 1; li < S.n_lights; ++li) {
            const DevLight& L = S.lights[li];
            f3 ldir = L.kind == PT_LIGHT_POINT ? normalize3(surf.pos - ld3(L.vec)) : ld3(L.vec);
            f3 c = mul_ew(thr, ct_eval_direct(brdf, normal, view, -1.f * ldir));
            contrib[(size_t)li * W.scap + sh_idx] = make_float4(c.x, c.y, c.z, 0.f);
        }
    } else if (live && hit && !survive) {  // no light can contribute and the path ends: the sample is complete
        float* out = staging + (size_t)out_slot * 3;
        out[0] = color.x;
        out[1] = color.y;
        out[2] = color.z;
    }
    // draws made HERE (the alpha walk counts its own): since the value this kernel started from, plus the jitter
    if (COUNT && live)
        n_draws += GRID >= 2 ? draw
                             : draw - (ALPHA ? draws[i] : PRIMARY ? 0u : (__float_as_uint(wf_ray_rec(queue_in, i)[1].w) & 0xffffu)) +
                                   ((ALPHA && PRIMARY) ? 2u : 0u);
    };  // shade_one
    if (PRIMARY || !WF_SHADE_AGGREGATE) {
        // grid-stride over the queue, one workgroup-wide step at a time (the loop bound is uniform in the workgroup)
        // Camera-grid cull (k_cam_block_mask): no camera ray of an EMPTY 8x8 pixel block can hit anything, so its samples
        // are the background - no ChaCha block, no cast (the instrumented variant counts them the long way).  A wavefront
        // is one block of one sample; chunks and queues are whole wavefronts.  Nothing is staged for them: k_accumulate
        // adds the background for the pixels of an empty block itself, once per sample.
        // (word W.n_mask_blocks of the table: non-zero if any block is empty - a frame without one pays nothing here)
        const bool cull = PRIMARY && GRID >= 2 && !COUNT && block_empty != nullptr && block_empty[W.n_mask_blocks] != 0u;
        for (uint32_t base = blockIdx.x * WF_SHADE_THREADS; base < n; base += gridDim.x * WF_SHADE_THREADS) {
            const uint32_t e = base + threadIdx.x;
            bool live = e < n && (PRIMARY || entry_word(e) != WF_HIT_PENDING);
            if (cull) {   // a step whose four wavefronts are all empty skips the compaction's barriers as well
                const uint32_t g0 = (W.item_base + base) >> 6;
                bool step_empty = true;
#pragma unroll
                for (uint32_t k = 0; k < WF_SHADE_THREADS / 64; ++k) {
                    const bool em = base + 64u * k >= n || block_empty[pt_fastdiv(g0 + k, W.P.div_batch)] != 0u;
                    step_empty = step_empty && em;
                    live = (k == wave && em) ? false : live;
                }
                if (step_empty) continue;   // (the same answer in every thread of the workgroup)
            }
            shade_one(e, live);
        }
    } else {
        // Hit aggregation (bounces >= 1).  Three of four secondary rays of an open scene leave into the background:
        // shaded in queue order, 23 % of the lanes would carry the material fetch, the BRDF and the GGX sample while
        // the others wait (27.7 of 64 lanes active per vector instruction, profiles/r02_a_pmc.json).  So a first
        // pass over a step's 256 entries retires the misses (background term, mod.rs:184-186: 48 of the 64 record
        // bytes) and collects the indices of the hits in LDS; whenever 256 are waiting - or the queue has ended -
        // they are shaded together.  The order in which paths are shaded is free (results are keyed by out_slot),
        // so no bit changes.
        __shared__ uint32_t agg[2 * WF_SHADE_THREADS];
        __shared__ uint32_t agg_cnt[WF_SHADE_THREADS / 64];
        uint32_t have = 0;   // waiting hits (the same value in every thread)
        uint32_t base = blockIdx.x * WF_SHADE_THREADS;
        while (true) {
            while (have < WF_SHADE_THREADS && base < n) {
                const uint32_t e = base + threadIdx.x;
                const uint32_t i = e < n ? entry_index(e) : e;   // (queue record; the hit is stored by e)
                base += gridDim.x * WF_SHADE_THREADS;
                bool is_hit = false;
                const uint32_t word = e < n ? entry_word(e) : WF_HIT_PENDING;
                if (word != WF_HIT_PENDING) {
                    is_hit = word != 0xffffffffu;
                    if (!is_hit) {
                        const float4* qp = wf_path_rec(queue_in, W.qcap_in, i);
                        const float4 q2 = qp[0], q3 = qp[1];
                        const uint32_t slot = __float_as_uint(q2.w);
                        if (slot != 0xffffffffu) {
                            const f3 c = mk3(q3.x, q3.y, q3.z) + mul_ew(mk3(q2.x, q2.y, q2.z), ld3(S.background));
                            float* out = staging + (size_t)slot * 3;
                            out[0] = c.x;
                            out[1] = c.y;
                            out[2] = c.z;
                        }
                    }
                }
                const unsigned long long m = __ballot(is_hit);
                if ((threadIdx.x & 63u) == 0) agg_cnt[wave] = (uint32_t)__popcll(m);
                __syncthreads();
                uint32_t pos = have + wf_lane_rank(m), total = 0;
                for (uint32_t k = 0; k < WF_SHADE_THREADS / 64; ++k) {
                    if (k < wave) pos += agg_cnt[k];
                    total += agg_cnt[k];
                }
                if (is_hit) agg[pos] = e;
                have += total;
                __syncthreads();
            }
            if (have == 0) break;
            const uint32_t take = have < WF_SHADE_THREADS ? have : (uint32_t)WF_SHADE_THREADS;
            have -= take;
            uint32_t mine = threadIdx.x < take ? agg[have + threadIdx.x] : 0u;
            __syncthreads();
            if (W.sort_octants & 2u) {
                // Material sorting (measured option, PT_WF_SORT=2): the 256 hits of a step are ordered by the model -
                // i.e. the material - of the primitive they hit (8 classes, stable counting sort: ballot per class,
                // mbcnt rank, prefix over classes and waves in LDS), so that a wavefront shades one material's
                // texture set.  The order in which paths are shaded is free: no bit changes.
                __shared__ uint32_t agg_sorted[WF_SHADE_THREADS];
                uint32_t key = 8u, rank = 0u;
                if (threadIdx.x < take) {
                    const uint32_t prim = entry_word(mine) & 0x0fffffffu;
                    key = __float_as_uint(S.prim_attr[(size_t)prim * 4 + 3].w) & 7u;
                }
#pragma unroll
                for (uint32_t k = 0; k < 8u; ++k) {
                    const unsigned long long m = __ballot(key == k);
                    if ((threadIdx.x & 63u) == 0) sh_oct[k][wave] = (uint32_t)__popcll(m);
                    if (key == k) rank = wf_lane_rank(m);
                }
                __syncthreads();
                if (threadIdx.x < 8u * (WF_SHADE_THREADS / 64)) {
                    uint32_t before = 0;
                    for (uint32_t j = 0; j < threadIdx.x; ++j) before += sh_oct[j / (WF_SHADE_THREADS / 64)][j % (WF_SHADE_THREADS / 64)];
                    sh_oct_off[threadIdx.x / (WF_SHADE_THREADS / 64)][threadIdx.x % (WF_SHADE_THREADS / 64)] = before;
                }
                __syncthreads();
                if (threadIdx.x < take) agg_sorted[sh_oct_off[key][wave] + rank] = mine;
                __syncthreads();
                mine = threadIdx.x < take ? agg_sorted[threadIdx.x] : 0u;
                __syncthreads();
            }
            shade_one(mine, threadIdx.x < take);
        }
    }
    if (COUNT && n_draws) atomicAdd(&gctr->rng_draws, (unsigned long long)n_draws);
    if (COUNT && n_new) atomicAdd(&gctr->samples, (unsigned long long)n_new);
    if (COUNT && n_moot) {
        atomicAdd(&gctr->shadow_rays, (unsigned long long)n_moot);
        atomicAdd(&gctr->shadow_skipped, (unsigned long long)n_moot);
    }
    if (COUNT && PRIMARY && n_hits) atomicAdd(&gctr->bounce0_hits, (unsigned long long)n_hits);
    if (COUNT && n_masked) {   // (ray_cast calls the reference makes and this pipeline proves empty)
        atomicAdd(&gctr->segments, (unsigned long long)n_masked);
        atomicAdd(&gctr->masked_casts, (unsigned long long)n_masked);
        if (PRIMARY) atomicAdd(&gctr->bounce0_masked, (unsigned long long)n_masked);
    }
    if (COUNT && GRID != 0) {
        atomicAdd(&gctr->segments, (unsigned long long)lc.segments);
        atomicAdd(&gctr->shadow_rays, (unsigned long long)lc.shadow_rays);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->grid_tris, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
        if (GRID >= 2) {
            atomicAdd(&gctr->trace_tris, (unsigned long long)n_cam_tris);
            atomicAdd(&gctr->bounce0_cam_tris, (unsigned long long)n_cam_tris);
            atomicAdd(&gctr->bounce0_tris, (unsigned long long)lc.tris);
            atomicAdd(&gctr->bounce0_shadow_rays, (unsigned long long)lc.shadow_rays);
        }
        if (ALPHA) atomicAdd(&gctr->shaded_hits, (unsigned long long)lc.shaded);
    }
}

// ---------------------------------------------------------------------------
// shadow: get_light_info for every light of every shaded surface (mod.rs:281-333), adds the
// direct light in light order and retires finished paths.  Same refill / walk / leaf phases as
// k_wf_trace; a job is one shaded surface, its lights are cast one after the other.
// ---------------------------------------------------------------------------
template <bool ALPHA, bool COUNT>
__global__ __launch_bounds__(WF_THREADS, (!COUNT && !ALPHA) ? WF_PRIMARY_WAVES : WF_MIN_WAVES) void k_wf_shadow(DevScene S, WfParams W, float4* __restrict__ shadow_q,
                                                          const float4* __restrict__ contrib,
                                                          float4* __restrict__ queue_next, float* __restrict__ staging,
                                                          uint32_t* __restrict__ offgrid,
                                                          WfCounters* __restrict__ ctr, DevCounters* __restrict__ gctr) {
    __shared__ unsigned long long lds_stack[WF_LDS_STACK * WF_THREADS];
    __shared__ unsigned long long lds_top[WF_LDS_NODES ? WF_LDS_NODES : 1];
    wf_load_tree_top(S, lds_top);
    const uint32_t n = min(ctr[W.bounce].shadow_count, W.scap);
    uint32_t* cursor = ctr[W.bounce].shadow_work;
    uint32_t ov_node[PT_KD_STACK - WF_LDS_STACK];
    float ov_tmax[PT_KD_STACK - WF_LDS_STACK];
    const TravStack st = {(wf_lds_u64*)(lds_stack + threadIdx.x), ov_node, ov_tmax, (const wf_lds_u64*)lds_top};
    Trav T;
    // per-lane job state
    uint32_t idx = 0, li = 0, out_slot = 0, next_idx = 0;
    f3 pos = mk3(0, 0, 0), gn = mk3(0, 0, 0), color = mk3(0, 0, 0), rad = mk3(0, 0, 0);
    f2 uv = {0.f, 0.f};
    bool sphere = false, point = false, blocked = false;
    float ldist = 0.f, limit = INFINITY, t_prev = -INFINITY;
    uint32_t ord_prev = 0;
    RawHit best;
    best.key = INFINITY;
    best.ord = 0xffffffffu;
    best.pid = 0xffffffffu;
    bool active = false, exhausted = false, need_begin = false, handed = false;
    uint32_t lstate = WF_LANE_IDLE;
    uint32_t mailbox = 0xffffffffu;
    LocalCtr lc = {0, 0, 0, 0, 0, 0};
    WaveFetch wf = wf_fetch_init(n);

    auto add_light = [&]() {  // visibility known: add the light (mod.rs:251-261)
        if (!(rad.x == 0.f && rad.y == 0.f && rad.z == 0.f)) {
            float4 c = contrib[(size_t)li * W.scap + idx];
            color = color + mul_ew(mk3(c.x, c.y, c.z), rad);
        }
        ++li;
    };
    // start the cast for light li of the current job; false when the job has no more lights
    auto begin_light = [&]() -> bool {
        while (li < S.n_lights) {
            const DevLight& L = S.lights[li];
            if (S.n_lights > 1) {  // (a single light was already filtered by k_wf_shade)
                float4 c = contrib[(size_t)li * W.scap + idx];
                if (wf_light_is_moot(L, mk3(c.x, c.y, c.z), pos)) {
                    if (COUNT) {
                        lc.shadow_rays++;
                        lc.shaded++;   // (re-used as the skipped-cast counter in this kernel)
                    }
                    ++li;
                    continue;
                }
            }
            point = L.kind == PT_LIGHT_POINT;
            f3 direction;
            rad = ld3(L.color);
            if (point) {
                direction = pos - ld3(L.vec);
                ldist = mag3(direction);
                direction = normalize3(direction);
                rad = rad / (4.f * PT_PI * ldist * ldist);
            } else {
                direction = ld3(L.vec);
                ldist = 0.f;
            }
            f3 so = pos + gn * 0.00001f;
            f3 sd = -1.f * direction;
            // a shadow ray that would need more slack than PT_SLACK_MAX is not walked here (slop model, pt_integrator.h): the
            // job goes on - from this light, with the colour so far - in k_og_shadow_offgrid, on the exact scalar walker
            if (W.exact_handover && slack_is_capped(__builtin_amdgcn_rcpf(sd.x), __builtin_amdgcn_rcpf(sd.y), __builtin_amdgcn_rcpf(sd.z))) {
                handed = true;
                return false;
            }
            if (COUNT) lc.shadow_rays++;
            blocked = false;
            t_prev = -INFINITY;
            ord_prev = 0;
            mailbox = 0xffffffffu;
            best.key = INFINITY;
            best.ord = 0xffffffffu;
            best.pid = 0xffffffffu;
            // opaque scenes: hits farther than the light cannot pass the range test
            limit = (!ALPHA && point) ? (ldist + 1e-4f) * 1.0001f : INFINITY;
            if (trav_start(S, T, so, sd, 0.f)) {
                lstate = WF_LANE_WALK;
                return true;
            }
            add_light();  // the shadow ray misses the scene box: unoccluded
        }
        return false;
    };
    auto retire = [&]() {
        if (next_idx == 0xffffffffu) {
            float* out = staging + (size_t)out_slot * 3;
            out[0] = color.x;
            out[1] = color.y;
            out[2] = color.z;
        } else {
            wf_path_rec(queue_next, W.qcap_out, next_idx)[1] = make_float4(color.x, color.y, color.z, 0.f);
        }
        active = false;
    };
    // the current cast has no further segment (ONE call site per loop iteration)
    auto complete = [&]() {
        lstate = WF_LANE_IDLE;
        if (ALPHA) {
            // best = next entry of the sorted list: attenuate, then look for the following one
            bool more = best.pid != 0xffffffffu;
            if (more && !hit_passes_slab(S, T.o, T.d)) more = false;   // kdtree-ray's box test
            if (more) {
                float opacity = 0.f;
                if (point) {
                    f3 sp = T.o + T.d * ((best.flags & 2u) ? best.u : best.key);
                    if (mag3(sp - pos) > ldist) {
                        more = false;
                    } else {
                        // the SHADED hit's kind / uv with the occluder's material (mod.rs:324)
                        uint32_t smodel = __float_as_uint(S.prim_attr[(size_t)PT_PRIM_INDEX(best.pid) * 4 + 3].w);
                        opacity = material_opacity(S, smodel, sphere, uv);
                    }
                } else {
                    opacity = hit_opacity(S, T.o, T.d, best);
                }
                if (more) {
                    rad = rad * (1.f - opacity);
                    if (sum3(rad) == 0.f) more = false;
                }
            }
            if (more) {
                t_prev = best.key;
                ord_prev = best.ord;
                mailbox = 0xffffffffu;
                best.key = INFINITY;
                best.ord = 0xffffffffu;
                best.pid = 0xffffffffu;
                if (COUNT) lc.restarts++;
                if (trav_start(S, T, T.o, T.d, next_start(t_prev, T.d))) {  // keep walking
                    lstate = WF_LANE_WALK;
                    return;
                }
            }
        } else if (blocked) {
            rad = rad * 0.0f;
        }
        add_light();
        need_begin = true;   // next light (or retire) at the top of the next iteration
    };

    while (true) {
        bool need = !active && !exhausted;
        unsigned long long m_need = __ballot(need);
        if (m_need && ((uint32_t)__popcll(m_need) >= W.refill_min || !__any(active))) {
            bool got;
            uint32_t w = wave_fetch(wf, cursor, n, need, got, exhausted);
            if (got) {
                idx = w;
                const float4* sq = shadow_q + (size_t)idx * 4;
                float4 s0 = sq[0], s1 = sq[1], s2 = sq[2], s3 = sq[3];
                pos = mk3(s0.x, s0.y, s0.z);
                gn = mk3(s0.w, s1.x, s1.y);
                uv = {s1.z, s1.w};
                color = mk3(s2.x, s2.y, s2.z);
                next_idx = __float_as_uint(s2.w);
                out_slot = __float_as_uint(s3.x);
                sphere = (__float_as_uint(s3.y) & WF_FLAG_SPHERE) != 0;
                li = 0;
                active = true;
                need_begin = true;
            }
        }
        if (active && need_begin) {  // the ONE place a light's shadow cast starts
            need_begin = false;
            if (!begin_light()) {
                if (handed) {   // (rare: one atomic per job is fine)
                    handed = false;
                    float4* sq = shadow_q + (size_t)idx * 4;
                    sq[2] = make_float4(color.x, color.y, color.z, __uint_as_float(next_idx));
                    sq[3] = make_float4(__uint_as_float(out_slot), __uint_as_float(sphere ? WF_FLAG_SPHERE : 0u), __uint_as_float(li), 0.f);
                    offgrid[atomicAdd(&ctr[W.bounce].offgrid_count, 1u)] = idx;
                    active = false;
                } else {
                    retire();
                }
            }
        }
        if (!__any(active)) {
            if (__all(exhausted)) break;
            continue;
        }
        // ---- phase A: walk
        for (uint32_t k = 0; k < W.walk_steps; ++k) {
            const bool walking = lstate == WF_LANE_WALK;
            // (kept to a compare + branch on vcc: a popcount threshold here - leave once only a few lanes
            // still walk - put a VALU->SALU dependency on the critical path of every step and cost 18 %)
            if (!wf_any(walking)) break;
            if (walking) lstate = trav_step<COUNT>(S, T, st, ALPHA ? best.key : limit, lc);
        }
        // ---- phase B
        if (lstate == WF_LANE_LEAF) {
            if (!ALPHA) {
                // every opacity is exactly 1: any hit inside the light's range blocks it
                const float4* lp = S.leaf_prims + (size_t)T.leaf.x * 3;
                uint32_t np = T.leaf.y >> 2;
                        for (uint32_t i = 0; i < np && !blocked; ++i) {
                    float4 q0, q1, q2;
                    load_prim_record(lp + 3 * i, q0, q1, q2);
                    uint32_t pid = __float_as_uint(q0.w);
                    if (COUNT) lc.tris++;
                    if (!(pid & PT_PRIM_SPHERE)) {
                        float t, u, v;
                        bool bf;
                        if (!isect_triangle(T.o, T.d, mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z),
                                            mk3(q1.w, q2.x, q2.y), t, u, v, bf))
                            continue;
                        if (point && mag3((T.o + T.d * t) - pos) > ldist) continue;
                        blocked = true;
                    } else {
                        float t[2], key[2];
                        bool ex[2];
                        int nh = isect_sphere(T.o, T.d, mk3(q0.x, q0.y, q0.z), q1.x, t, key, ex);
                        for (int k = 0; k < nh; ++k) {
                            if (!(key[k] == key[k])) continue;
                            if (point && mag3((T.o + T.d * t[k]) - pos) > ldist) continue;
                            blocked = true;
                        }
                    }
                }
                // kdtree-ray's box test: a ray the scene box rejects has no hits at all
                const bool rejected = blocked && !hit_passes_slab(S, T.o, T.d);
                if (rejected) blocked = false;
                lstate = (blocked || rejected || !trav_pop(T, st, limit)) ? WF_LANE_DONE : WF_LANE_WALK;
            } else {
                leaf_closest<COUNT>(S, T, T.leaf, t_prev, ord_prev, best, lc, mailbox);
                lstate = trav_pop(T, st, best.key) ? WF_LANE_WALK : WF_LANE_DONE;
            }
        }
        if (lstate == WF_LANE_DONE) complete();
    }
    if (COUNT) {
        atomicAdd(&gctr->shadow_rays, (unsigned long long)lc.shadow_rays);
        atomicAdd(&gctr->shadow_skipped, (unsigned long long)lc.shaded);
        atomicAdd(&gctr->nodes_visited, (unsigned long long)lc.nodes);
        atomicAdd(&gctr->tris_tested, (unsigned long long)lc.tris);
        atomicAdd(&gctr->restarts, (unsigned long long)lc.restarts);
    }
}

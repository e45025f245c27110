// VALU issue ceilings of one MI355X: wave64 instructions per CU-cycle for a few instruction kinds, at 1..8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_ceiling tools/micro/valu_ceiling.hip && ./valu_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t iters, uint32_t seed) {
    uint32_t a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 7u + i * 13u; f[i] = (float)a[i] * 1e-9f; }
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = a[i] + a[(i + 1) & 7];                       // v_add_u32
                if (KIND == 1) a[i] = a[i] ^ a[(i + 1) & 7];                       // v_xor_b32
                if (KIND == 2) a[i] = __builtin_amdgcn_alignbit(a[i], a[i], 16u + (uint32_t)r);  // v_alignbit_b32 (rotate)
                if (KIND == 3) f[i] = __builtin_fmaf(f[i], 1.0000001f, f[(i + 1) & 7]);  // v_fma_f32
                if (KIND == 4) f[i] = f[i] * f[(i + 1) & 7];                       // v_mul_f32
                if (KIND == 5) a[i] = a[i] * a[(i + 1) & 7];                       // v_mul_lo_u32
                if (KIND == 6) { a[i] += a[(i + 1) & 7]; a[(i + 3) & 7] ^= a[i]; a[(i + 3) & 7] = __builtin_amdgcn_alignbit(a[(i + 3) & 7], a[(i + 3) & 7], 16u); }  // ChaCha-like mix (3 instr)
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + __float_as_uint(f[i]);
    if (s == 0x12345u) out[0] = s;
}
int main() {
    uint32_t* d; hipMalloc(&d, 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const char* names[7] = {"v_add_u32", "v_xor_b32", "v_alignbit_b32", "v_fma_f32", "v_mul_f32", "v_mul_lo_u32", "add+xor+alignbit"};
    void (*kern[7])(uint32_t*, uint32_t, uint32_t) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
    const int per_iter[7] = {64, 64, 64, 64, 64, 64, 192};
    for (int kind = 0; kind < 7; ++kind)
        for (int wg_per_cu : {1, 2, 4, 8}) {   // 256 threads = 1 wave per SIMD
            uint32_t iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern[kind], dim3(p.multiProcessorCount * wg_per_cu), dim3(256), 0, 0, d, iters, 1u);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double wave_insts = (double)p.multiProcessorCount * wg_per_cu * 4 * iters * per_iter[kind];
            double per_cu_s = wave_insts / p.multiProcessorCount / (best * 1e-3);
            printf("%-18s %d waves/SIMD: %7.3f ms  %.3f G wave-instr/s/CU = %.3f per CU-cycle at 2.4 GHz (%.3f at 2.1)\n", names[kind], wg_per_cu, best,
                   per_cu_s / 1e9, per_cu_s / 2.4e9, per_cu_s / 2.1e9);
        }
    return 0;
}

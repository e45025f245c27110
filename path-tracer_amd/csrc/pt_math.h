// f32 vector helpers in the reference's operation order + Rust scalar casts.
//
// The integrator must perform exactly the f32 operations of the reference in
// the same order (cgmath 0.18 conventions): dot = (x*x + y*y) +
// z*z, normalize = v * (1/|v|), `v / s` divides per component, no FMA
// contraction (the file is compiled with -ffp-contract=off), IEEE divide and
// sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#define PT_HD __host__ __device__ __forceinline__
#define PT_D __device__ __forceinline__

struct f3 {
    float x, y, z;
};
struct f2 {
    float x, y;
};

PT_HD f3 mk3(float x, float y, float z) { return {x, y, z}; }
PT_HD f3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
PT_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
PT_HD f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_HD f3 operator*(float s, f3 a) { return {s * a.x, s * a.y, s * a.z}; }
PT_HD f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PT_HD f3 mul_ew(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD f3 div_ew(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD f3 cross3(f3 a, f3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
PT_HD float mag3(f3 a) { return sqrtf(dot3(a, a)); }
PT_HD f3 normalize3(f3 a) { return a * (1.0f / mag3(a)); }
PT_HD float sum3(f3 a) { return a.x + a.y + a.z; }
PT_HD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
PT_HD f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
PT_HD f2 operator*(float s, f2 a) { return {s * a.x, s * a.y}; }

// Rust: f32::max ignores a NaN operand (fmaxf); powi with a constant exponent is
// expanded by LLVM into a multiplication chain.
PT_HD float max_rs(float a, float b) { return fmaxf(a, b); }
PT_HD float powi5(float t) { return t * ((t * t) * (t * t)); }
PT_HD float powi2(float t) { return t * t; }
// `x as u8`: saturating, NaN -> 0.
PT_HD uint8_t as_u8(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.f) return 0;
    if (v >= 255.f) return 255;
    return (uint8_t)v;
}
// `x as i64` then rem_euclid(n) for texture wrap (internal/material.rs:115-130).
PT_HD uint32_t wrap_texel(float c, uint32_t n) {
#ifndef PT_WRAP_SLOW   // (A/B builds: the 64-bit form for every coordinate)
    // Every sane texture coordinate: |c| < 2^31, the conversion is exact in 32 bits and the remainder needs no 64-bit division
    // (the signed 64-bit `%` below is ~1100 of the bounce-0 kernel's ~6500 vector instructions, twice per texel fetch); a
    // power-of-two size needs no division at all (two's complement: i & (n - 1) IS the Euclidean remainder).
    if (fabsf(c) < 2147483648.0f && n != 0u) {   // (NaN: the general form)
        const int32_t i32 = (int32_t)c;
        if ((n & (n - 1u)) == 0u) return (uint32_t)i32 & (n - 1u);
        const uint32_t a = i32 < 0 ? 0u - (uint32_t)i32 : (uint32_t)i32;
        const uint32_t q = a % n;
        return (i32 < 0 && q != 0u) ? n - q : q;
    }
#endif
    long long i;
    if (!(c == c)) i = 0;
    else if (c >= 9223372036854775807.0f) i = 0x7fffffffffffffffLL;
    else if (c <= -9223372036854775808.0f) i = (long long)0x8000000000000000ULL;
    else i = (long long)c;
    long long r = i % (long long)n;
    if (r < 0) r += (long long)n;
    return (uint32_t)r;
}

#define PT_PI 3.14159265358979323846f

// ---------------------------------------------------------------------------
// libm entry points of the hot path.  The reference calls glibc's powf /
// acosf / sinf / cosf through Rust's std; these wrappers are the single place
// where the device implementation is chosen (pt_libm.h restates the glibc
// 2.35 algorithms so that the GPU image can match the CPU bit for bit).
// ---------------------------------------------------------------------------
#include "pt_libm.h"

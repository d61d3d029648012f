// bb_kernels_dev.cuh -- device-side helpers shared by the kernel translation units of the BabyBear backend (bb_kernels.hip, bb_rounds.hip): plane-major
// F_{p^9} element access, lazy 96-bit sums of product columns, wave / block reductions, grid helpers, base-2 digits.
#pragma once
#include "bb_kernels.h"

namespace lfbb {

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
static inline unsigned grid_for(size_t n, unsigned cap = 2048) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}


__device__ __forceinline__ E9 ld9(const fe *tab, size_t ld, u32 slot, size_t i) {
    E9 r;
#pragma unroll
    for (int c = 0; c < TAU; c++) r.c[c] = tab[(size_t)(TAU * slot + c) * ld + i];
    return r;
}
__device__ __forceinline__ void st9(fe *tab, size_t ld, u32 slot, size_t i, const E9 &v) {
#pragma unroll
    for (int c = 0; c < TAU; c++) tab[(size_t)(TAU * slot + c) * ld + i] = v.c[c];
}
__device__ __forceinline__ E9 e9c(const E9C &k) { E9 r; for (int i = 0; i < TAU; i++) r.c[i] = k.c[i]; return r; }
__device__ __forceinline__ E9Pre e9p(const E9PreC &k) { E9Pre r; for (int i = 0; i < TAU; i++) { r.v.c[i] = k.v[i]; r.vn.c[i] = k.vn[i]; } return r; }
// reduce a signed 64-bit sum of residues to a centred word
// |s| <= 9 H^2: two Montgomery steps (s * R^-1, then * R^2 * R^-1) instead of a 64-bit modulo
__device__ __forceinline__ fe fred(i64 s) { return fmul(mred(s), BB_R2C); }

// sum of un-reduced product columns kept as (sum of high halves, sum of low halves); value = Montgomery-reduced total
// (round 4: one signed 96-bit integer in three registers -- add with carry, carry, carry -- instead of two 64-bit sums: three instructions per column instead
// of five and 9 registers less per lazy sum of an F_{p^9} product)
struct HL { u32 a0, a1; int32_t a2; };
__device__ __forceinline__ void hl_zero(HL &a) { a.a0 = 0; a.a1 = 0; a.a2 = 0; }
__device__ __forceinline__ void hl_add(HL &a, i64 T) {
    const u32 t0 = (u32)T, t1 = (u32)((u64)T >> 32);
    const int32_t t2 = (int32_t)t1 >> 31;                     // sign extension word
    asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %5, vcc"
        : "+v"(a.a0), "+v"(a.a1), "+v"(a.a2)
        : "v"(t0), "v"(t1), "v"(t2)
        : "vcc");
}
// V = a2 2^64 + a1 2^32 + a0 (a2 signed): V 2^-32 mod p = a2 2^32 + a1 + a0 2^-32, centred Montgomery word like mred of the total
__device__ __forceinline__ fe hl_finish(const HL &a) {
    const i64 mid = (i64)a.a1;                                // < 2^32 < 2.2 p
    return fadd(fadd(from_small(a.a2), fred(mid)), mred((i64)a.a0));
}

// ---------------------------------------------------------------------------------------------------------
// reductions: every thread holds NV signed 64-bit partial sums (of centred words)
__device__ __forceinline__ i64 wave_sum(i64 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_down((long long)v, off, 64);
    return v;
}
template <int NV>
__device__ __forceinline__ void block_sum_store(i64 (&v)[NV], i64 *dst) {
    __shared__ i64 sm[4][NV];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        i64 s = wave_sum(v[i]);
        if (lane == 0) sm[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += 256) dst[i] = sm[0][i] + sm[1][i] + sm[2][i] + sm[3][i];
}

constexpr u32 RED_BLOCKS = 256;   // partial rows of every two-stage reduction

// bit-plane k of a centred small value: sign(v) * bit_k(|v|)   (base-2 balanced digits, decomposition.rs:159-167)
__device__ __forceinline__ int digit2(int32_t v, u32 k) {
    int32_t m = v < 0 ? -v : v;
    int d = (m >> k) & 1;
    return v < 0 ? -d : d;
}
__device__ __forceinline__ fe fe_from_digit(int d) { return d == 0 ? 0 : (d > 0 ? BB_ONE : -BB_ONE); }

}  // namespace lfbb

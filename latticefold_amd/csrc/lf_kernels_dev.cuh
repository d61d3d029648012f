// lf_kernels_dev.cuh -- device-side helpers shared by the kernel translation units of the Goldilocks backend (lf_kernels.hip, lf_rounds.hip): the F_{p^3}
// product wrappers, the nu-specialised launch macro, grid helpers, wave / block reductions, plane-major element access, base-2 digits.
#pragma once
#include "lf_kernels.h"

namespace lf {

#define NUARG t.nu
template <bool NU> __device__ __forceinline__ Fq3 M3(Fq3 a, Fq3 b, u64 nu) { return fq3_mul<NU>(a, b, nu); }
template <bool NU> __device__ __forceinline__ Fq3 S3(Fq3 a, u64 nu) { return fq3_sqr<NU>(a, nu); }

#define LF_LAUNCH(KERNEL, nuflag, grid, block, stream, ...)                                   \
    do {                                                                                      \
        if (nuflag) hipLaunchKernelGGL((KERNEL<true>), grid, block, 0, stream, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<false>), grid, block, 0, stream, __VA_ARGS__);        \
    } while (0)

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
static inline unsigned grid_for(size_t n, unsigned cap = 2048) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------------------------
// reductions
__device__ __forceinline__ u64 wave_sum_fq(u64 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        u64 o = __shfl_down((unsigned long long)v, off, 64);
        v = fq_add(v, o);
    }
    return v;
}
// sum `v[0..NV)` over the 256 threads of the block, write to dst[0..NV) (thread-0-side); values canonical
template <int NV>
__device__ __forceinline__ void block_sum_store(u64 (&v)[NV], u64 *dst) {
    __shared__ u64 sm[4][NV];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        u64 s = wave_sum_fq(v[i]);
        if (lane == 0) sm[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += 256) dst[i] = fq_add(fq_add(sm[0][i], sm[1][i]), fq_add(sm[2][i], sm[3][i]));
}

__device__ __forceinline__ u64 fq_from_digit(int d) { return d == 0 ? 0 : (d > 0 ? 1 : LF_P - 1); }

// bit-plane k of a centred small value: sign(v) * bit_k(|v|)   (base-2 balanced digits, decomposition.rs:159-167)
__device__ __forceinline__ int digit2(int32_t v, u32 k) {
    int32_t m = v < 0 ? -v : v;
    int d = (m >> k) & 1;
    return v < 0 ? -d : d;
}

__device__ __forceinline__ Fq3 ld3(const u64 *tab, size_t ld, u32 slot, size_t i) {
    return fq3_make(tab[(size_t)(3 * slot) * ld + i], tab[(size_t)(3 * slot + 1) * ld + i], tab[(size_t)(3 * slot + 2) * ld + i]);
}
__device__ __forceinline__ void st3(u64 *tab, size_t ld, u32 slot, size_t i, Fq3 v) {
    tab[(size_t)(3 * slot) * ld + i] = v.c[0]; tab[(size_t)(3 * slot + 1) * ld + i] = v.c[1]; tab[(size_t)(3 * slot + 2) * ld + i] = v.c[2];
}

constexpr u32 RED_BLOCKS = 256;   // partial rows of every two-stage reduction

// out[i] = sum_b partial[b*nv + i]; one block per i
static __global__ void __launch_bounds__(256) k_reduce_rows(const u64 *partial, u32 nblocks, u32 nv, u64 *out) {
    u32 i = blockIdx.x;
    u64 acc[1] = {0};
    for (u32 b = threadIdx.x; b < nblocks; b += 256) acc[0] = fq_add(acc[0], partial[(size_t)b * nv + i]);
    block_sum_store<1>(acc, out + i);
}

}  // namespace lf

// lf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the LatticeFold prover hot path.
//
// Every kernel works on plane-major (SoA) tables so that lane <-> consecutive element index gives coalesced
// 8/16-byte accesses; cross-lane reductions use wave64 shuffles + one LDS hop per 256-thread block; the Ajtai
// mat-vec stages (A, witness) tiles through LDS.  No MFMA: the arithmetic is 64-bit modular (four
// v_mad_u64_u32 per product, see lf_field.cuh).  Reference semantics each kernel replaces are cited inline.
#include "lf_kernels.h"

#include <stdlib.h>

namespace lf {

#define NUARG t.nu
template <bool NU> __device__ __forceinline__ Fq3 M3(Fq3 a, Fq3 b, u64 nu) { return fq3_mul<NU>(a, b, nu); }
template <bool NU> __device__ __forceinline__ Fq3 S3(Fq3 a, u64 nu) { return fq3_sqr<NU>(a, nu); }

#define LF_LAUNCH(KERNEL, nuflag, grid, block, stream, ...)                                   \
    do {                                                                                      \
        if (nuflag) hipLaunchKernelGGL((KERNEL<true>), grid, block, 0, stream, __VA_ARGS__);  \
        else hipLaunchKernelGGL((KERNEL<false>), grid, block, 0, stream, __VA_ARGS__);        \
    } while (0)

DevCrt make_dev_crt(const CrtTables &T) {
    DevCrt d;
    d.nu = T.nu; d.nu2p40 = T.nu_is_2p40;
    d.w4 = T.w4; d.w2 = T.w2; d.w10 = T.w10; d.w1 = T.w1; d.w7 = T.w7; d.w5 = T.w5; d.w11 = T.w11;
    for (int p = 0; p < 8; p++) {
        d.slot_of_pos[p] = T.slot_of_pos[p]; d.pos1[p] = T.pos1[p]; d.pos2[p] = T.pos2[p];
        d.tw1[p] = T.tw1[p]; d.tw2[p] = T.tw2[p];
    }
    return d;
}

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
static inline unsigned grid_for(size_t n, unsigned cap = 2048) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

__device__ __forceinline__ u64 splitmix_fq(u64 seed, u64 index);

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
// out[i] = sum_b partial[b*nv + i]; one block per i
__global__ void __launch_bounds__(256) k_reduce_rows(const u64 *partial, u32 nblocks, u32 nv, u64 *out) {
    u32 i = blockIdx.x;
    u64 acc[1] = {0};
    for (u32 b = threadIdx.x; b < nblocks; b += 256) acc[0] = fq_add(acc[0], partial[(size_t)b * nv + i]);
    block_sum_store<1>(acc, out + i);
}

// sharded exchanges: out[w] = sum_g parts[g*words + w] mod p after the all-gather of the ranks' partial vectors
__global__ void __launch_bounds__(256) k_modsum(const u64 *parts, u32 nparts, size_t words, u64 *out) {
    size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= words) return;
    u64 acc = 0;
    for (u32 g = 0; g < nparts; g++) acc = fq_add(acc, fq_canon(parts[(size_t)g * words + w]));
    out[w] = acc;
}
void launch_modsum(const u64 *parts, u32 nparts, size_t words, u64 *out, hipStream_t s) {
    if (words) hipLaunchKernelGGL(k_modsum, dim3(cdiv(words, 256)), dim3(256), 0, s, parts, nparts, words, out);
}
// all-gathered table slices [rank][plane][lcl] -> full tables [plane][nranks*lcl]
__global__ void __launch_bounds__(256) k_gather_relayout(const u64 *all, u32 nranks, size_t planes, size_t lcl, u64 *full) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, tot = (size_t)nranks * planes * lcl;
    if (i >= tot) return;
    size_t j = i % lcl, w = (i / lcl) % planes, rk = i / (lcl * planes);
    full[w * (nranks * lcl) + rk * lcl + j] = all[i];
}
void launch_gather_relayout(const u64 *all, u32 nranks, size_t planes, size_t lcl, u64 *full, hipStream_t s) {
    size_t tot = (size_t)nranks * planes * lcl;
    if (tot) hipLaunchKernelGGL(k_gather_relayout, dim3(cdiv(tot, 256)), dim3(256), 0, s, all, nranks, planes, lcl, full);
}
// the same for ONE table set of a payload that carries several ([rank][planes_tot][lcl], this set = planes [p0, p0 + planes)): one all-gather per hand-over
__global__ void __launch_bounds__(256) k_gather_relayout_part(const u64 *all, u32 nranks, size_t planes_tot, size_t p0, size_t planes, size_t lcl, u64 *full) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, tot = (size_t)nranks * planes * lcl;
    if (i >= tot) return;
    size_t j = i % lcl, w = (i / lcl) % planes, rk = i / (lcl * planes);
    full[w * (nranks * lcl) + rk * lcl + j] = all[(rk * planes_tot + p0 + w) * lcl + j];
}
void launch_gather_relayout_part(const u64 *all, u32 nranks, size_t planes_tot, size_t p0, size_t planes, size_t lcl, u64 *full, hipStream_t s) {
    size_t tot = (size_t)nranks * planes * lcl;
    if (tot) hipLaunchKernelGGL(k_gather_relayout_part, dim3(cdiv(tot, 256)), dim3(256), 0, s, all, nranks, planes_tot, p0, planes, lcl, full);
}

// ---------------------------------------------------------------------------------------------------------
// layout
__global__ void __launch_bounds__(256) k_aos_to_soa(const u64 *aos, u64 *soa, size_t n) {
    __shared__ u64 tile[64][25];
    size_t base = (size_t)blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * 24; idx += 256) {
        size_t e = base + idx / 24;
        tile[idx / 24][idx % 24] = e < n ? aos[e * 24 + idx % 24] : 0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 24; idx += 256) {
        int w = idx / 64, j = idx % 64;
        if (base + j < n) soa[(size_t)w * n + base + j] = tile[j][w];
    }
}
__global__ void __launch_bounds__(256) k_soa_to_aos(const u64 *soa, u64 *aos, size_t n) {
    __shared__ u64 tile[64][25];
    size_t base = (size_t)blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * 24; idx += 256) {
        int w = idx / 64, j = idx % 64;
        tile[j][w] = base + j < n ? soa[(size_t)w * n + base + j] : 0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 24; idx += 256) {
        size_t e = base + idx / 24;
        if (e < n) aos[e * 24 + idx % 24] = tile[idx / 24][idx % 24];
    }
}
void launch_aos_to_soa(const u64 *aos, u64 *soa, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_aos_to_soa, dim3(cdiv(n, 64)), dim3(256), 0, s, aos, soa, n);
}
void launch_soa_to_aos(const u64 *soa, u64 *aos, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_soa_to_aos, dim3(cdiv(n, 64)), dim3(256), 0, s, soa, aos, n);
}

__device__ __forceinline__ u64 splitmix_fq(u64 seed, u64 index) {
    u64 z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return z >= LF_P ? z - LF_P : z;
}
__global__ void __launch_bounds__(256) k_fill_uniform(u64 *dst, size_t words, u64 seed, size_t start) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < words; i += st) dst[i] = splitmix_fq(seed, start + i);
}
void launch_fill_uniform(u64 *dst, size_t words, u64 seed, size_t start, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_uniform, dim3(grid_for(words, 4096)), dim3(256), 0, s, dst, words, seed, start);
}
__global__ void __launch_bounds__(256) k_fill_ajtai(u64 *A, u32 kappa, size_t n, size_t n_total, size_t col0, u64 seed, u32 row0) {
    size_t total = (size_t)kappa * 24 * n;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < total; i += st) {
        size_t j = i % n, w = (i / n) % 24, row = row0 + i / (24 * n);
        A[i] = splitmix_fq(seed, (row * n_total + col0 + j) * 24 + w);
    }
}
// rows [row0, row0 + kappa) of the synthetic matrix into A [kappa][24][n]
void launch_fill_ajtai(u64 *A, u32 kappa, size_t n, size_t n_total, size_t col0, u64 seed, hipStream_t s, u32 row0) {
    hipLaunchKernelGGL(k_fill_ajtai, dim3(4096), dim3(256), 0, s, A, kappa, n, n_total, col0, seed, row0);
}

__device__ __forceinline__ u64 fq_from_digit(int d) { return d == 0 ? 0 : (d > 0 ? 1 : LF_P - 1); }

// ---------------------------------------------------------------------------------------------------------
// CRT: structured forward transform.  a(X) = sum_u X^u A_u(X^3); A_u is evaluated at the 8 primitive 24th
// roots by three radix-2 layers over Y^8 - Y^4 + 1 = (Y^4 - w^4)(Y^4 - w^20), then the per-slot monomial
// twist maps F_p[X]/(X^3 - zeta_k) onto F_p[Y]/(Y^3 - nu).  (stark-rings CRT; call sites arith.rs:238,327.)
__device__ __forceinline__ void crt8(const u64 x[8], u64 o[8], const DevCrt &t) {
    u64 lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 tt = fq_mul(t.w4, x[i + 4]);
        lo[i] = fq_add(x[i], tt);
        hi[i] = fq_sub(fq_add(x[i], x[i + 4]), tt);
    }
    u64 l0[2], l1[2], h0[2], h1[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        u64 tt = fq_mul(t.w2, lo[i + 2]);
        l0[i] = fq_add(lo[i], tt); l1[i] = fq_sub(lo[i], tt);
        u64 uu = fq_mul(t.w10, hi[i + 2]);
        h0[i] = fq_add(hi[i], uu); h1[i] = fq_sub(hi[i], uu);
    }
    u64 a = fq_mul(t.w1, l0[1]);  o[0] = fq_add(l0[0], a); o[1] = fq_sub(l0[0], a);
    u64 b = fq_mul(t.w7, l1[1]);  o[2] = fq_add(l1[0], b); o[3] = fq_sub(l1[0], b);
    u64 c = fq_mul(t.w5, h0[1]);  o[4] = fq_add(h0[0], c); o[5] = fq_sub(h0[0], c);
    u64 d = fq_mul(t.w11, h1[1]); o[6] = fq_add(h1[0], d); o[7] = fq_sub(h1[0], d);
}
// same butterflies for a TERNARY input (digits in {-1,0,1}): the first layer needs no multiplication (+-w4 or 0)
__device__ __forceinline__ void crt8_ternary(const int x[8], u64 o[8], const DevCrt &t) {
    u64 lo[4], hi[4];
    const u64 nw4 = LF_P - t.w4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u64 a = fq_from_digit(x[i]), b = fq_from_digit(x[i + 4]);
        u64 tt = x[i + 4] == 0 ? 0 : (x[i + 4] > 0 ? t.w4 : nw4);
        lo[i] = fq_add(a, tt);
        hi[i] = fq_sub(fq_add(a, b), tt);
    }
    u64 l0[2], l1[2], h0[2], h1[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        u64 tt = fq_mul(t.w2, lo[i + 2]);
        l0[i] = fq_add(lo[i], tt); l1[i] = fq_sub(lo[i], tt);
        u64 uu = fq_mul(t.w10, hi[i + 2]);
        h0[i] = fq_add(hi[i], uu); h1[i] = fq_sub(hi[i], uu);
    }
    u64 a = fq_mul(t.w1, l0[1]);  o[0] = fq_add(l0[0], a); o[1] = fq_sub(l0[0], a);
    u64 b = fq_mul(t.w7, l1[1]);  o[2] = fq_add(l1[0], b); o[3] = fq_sub(l1[0], b);
    u64 c = fq_mul(t.w5, h0[1]);  o[4] = fq_add(h0[0], c); o[5] = fq_sub(h0[0], c);
    u64 d = fq_mul(t.w11, h1[1]); o[6] = fq_add(h1[0], d); o[7] = fq_sub(h1[0], d);
}
__device__ __forceinline__ void crt_store_ternary(const int dg[24], u64 *out, size_t ld, size_t j, const DevCrt &t) {
    int x[8];
    u64 A0[8], A1[8], A2[8];
#pragma unroll
    for (int v = 0; v < 8; v++) x[v] = dg[3 * v];
    crt8_ternary(x, A0, t);
#pragma unroll
    for (int v = 0; v < 8; v++) x[v] = dg[3 * v + 1];
    crt8_ternary(x, A1, t);
#pragma unroll
    for (int v = 0; v < 8; v++) x[v] = dg[3 * v + 2];
    crt8_ternary(x, A2, t);
#pragma unroll
    for (int p = 0; p < 8; p++) {
        int s3 = 3 * t.slot_of_pos[p];
        out[(size_t)s3 * ld + j] = A0[p];
        out[(size_t)(s3 + t.pos1[p]) * ld + j] = fq_mul(t.tw1[p], A1[p]);
        out[(size_t)(s3 + t.pos2[p]) * ld + j] = fq_mul(t.tw2[p], A2[p]);
    }
}
// coefficients a[24] (canonical) -> stores the 24 NTT words of element j into plane table `out` (ld = n)
__device__ __forceinline__ void crt_store(const u64 a[24], u64 *out, size_t ld, size_t j, const DevCrt &t) {
    u64 x[8], A0[8], A1[8], A2[8];
#pragma unroll
    for (int v = 0; v < 8; v++) x[v] = a[3 * v];
    crt8(x, A0, t);
#pragma unroll
    for (int v = 0; v < 8; v++) x[v] = a[3 * v + 1];
    crt8(x, A1, t);
#pragma unroll
    for (int v = 0; v < 8; v++) x[v] = a[3 * v + 2];
    crt8(x, A2, t);
#pragma unroll
    for (int p = 0; p < 8; p++) {
        int s3 = 3 * t.slot_of_pos[p];
        out[(size_t)s3 * ld + j] = A0[p];
        out[(size_t)(s3 + t.pos1[p]) * ld + j] = fq_mul(t.tw1[p], A1[p]);
        out[(size_t)(s3 + t.pos2[p]) * ld + j] = fq_mul(t.tw2[p], A2[p]);
    }
}
__global__ void __launch_bounds__(256) k_crt_fwd(DevCrt t, const u64 *coef, u64 *ntt, size_t n) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    u64 a[24];
#pragma unroll
    for (int c = 0; c < 24; c++) a[c] = coef[(size_t)c * n + j];
    crt_store(a, ntt, n, j, t);
}
void launch_crt_fwd(const DevCrt &t, const u64 *coef, u64 *ntt, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_crt_fwd, dim3(cdiv(n, 256)), dim3(256), 0, s, t, coef, ntt, n);
}
// ICRT as the dense 24x24 F_p matrix (rare: ingest / export only)
__global__ void __launch_bounds__(256) k_icrt_dense(const u64 *mat, const u64 *ntt, u64 *coef, size_t n) {
    __shared__ u64 M[24 * 24];
    for (int i = threadIdx.x; i < 576; i += 256) M[i] = mat[i];
    __syncthreads();
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    u64 x[24];
#pragma unroll
    for (int c = 0; c < 24; c++) x[c] = ntt[(size_t)c * n + j];
    for (int i = 0; i < 24; i++) {
        Acc a;
        acc_set(a, M[i * 24], x[0]);
#pragma unroll
        for (int c = 1; c < 24; c++) acc_mad(a, M[i * 24 + c], x[c]);
        coef[(size_t)i * n + j] = acc_reduce(a);
    }
}
void launch_icrt_dense(const u64 *mat, const u64 *ntt, u64 *coef, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_icrt_dense, dim3(cdiv(n, 256)), dim3(256), 0, s, mat, ntt, coef, n);
}

// ---------------------------------------------------------------------------------------------------------
// balanced decomposition on canonical coefficients, power-of-two base (stark_rings::balanced_decomposition;
// call sites arith.rs:235, decomposition/utils.rs:23-31,48).  Sign-magnitude, |digit| <= base/2, ties kept.
__global__ void __launch_bounds__(256) k_decompose(const u64 *coef, size_t n, u32 log_base, u32 digits, int layout, u64 *out, int mode) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * 24) return;
    size_t c = idx / n, i = idx % n;
    u64 v = coef[idx];
    bool neg = v > (LF_P - 1) / 2;
    u64 mag = neg ? LF_P - v : v;
    u64 half = 1ULL << (log_base - 1), mask = (1ULL << log_base) - 1;
    size_t n_out = layout == 0 ? n * digits : n;
    int64_t cur = neg ? -(int64_t)mag : (int64_t)mag;   // |centred lift| <= (p-1)/2 < 2^63
    for (u32 k = 0; k < digits; k++) {
        int64_t dg;
        if (mode == 1 && log_base > 1) {
            // digit mode 1 (data, lf_set_digit_mode): floor / Euclidean rule, digits in [-base/2, base/2): rem = cur mod base, minus base if >= base/2
            int64_t rem = (int64_t)((u64)cur & mask);
            if ((u64)rem >= half) rem -= (int64_t)(mask + 1);
            cur = (cur - rem) >> log_base;
            dg = rem;
        } else {
            u64 rem = mag & mask;
            mag >>= log_base;
            if (rem > half) { dg = (int64_t)rem - (int64_t)(mask + 1); mag += 1; }
            else dg = (int64_t)rem;
            if (neg) dg = -dg;
        }
        size_t o = layout == 0 ? (c * n_out + i * digits + k) : ((size_t)k * 24 * n + c * n + i);
        out[o] = fq_from_i64(dg);
    }
}
void launch_decompose(const u64 *coef, size_t n, u64 base, u32 digits, int layout, u64 *out, hipStream_t s, int mode) {
    u32 lb = 0;
    while ((1ULL << lb) < base) lb++;
    if (n) hipLaunchKernelGGL(k_decompose, dim3(cdiv(n * 24, 256)), dim3(256), 0, s, coef, n, lb, digits, layout, out, mode);
}
// out[i] = sum_j base^j in[i*digits + j] on any table (linear, either form)
__global__ void __launch_bounds__(256) k_recompose(const u64 *in, size_t n_out, u64 base, u32 digits, u64 *out) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out * 24) return;
    size_t w = idx / n_out, i = idx % n_out;
    size_t n_in = n_out * digits;
    u64 acc = 0, pw = 1;
    for (u32 j = 0; j < digits; j++) {
        acc = fq_add(acc, fq_mul(in[w * n_in + i * digits + j], pw));
        pw = fq_mul(pw, base);
    }
    out[idx] = acc;
}
void launch_recompose(const u64 *in, size_t n_out, u64 base, u32 digits, u64 *out, hipStream_t s) {
    if (n_out) hipLaunchKernelGGL(k_recompose, dim3(cdiv(n_out * 24, 256)), dim3(256), 0, s, in, n_out, base % LF_P, digits, out);
}
__global__ void __launch_bounds__(256) k_coef_to_i32(const u64 *coef, int32_t *planes, size_t total, u32 bound, int *viol) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    int bad = 0;
    for (; i < total; i += st) {
        u64 v = coef[i];
        bool neg = v > (LF_P - 1) / 2;
        u64 mag = neg ? LF_P - v : v;
        if (mag > bound) { bad |= 1; mag = 0; }
        if (!neg && mag > 0x7fffffffull) { bad |= 2; mag = 0; }   // +2^31 (possible only with B = 2^32) has no int32 representation
        planes[i] = neg ? (int32_t)(0u - (u32)mag) : (int32_t)mag;
    }
    if (bad) atomicOr(viol, bad);
}
void launch_coef_to_i32(const u64 *coef, int32_t *planes, size_t n, u32 bound, int *viol, hipStream_t s) {
    hipLaunchKernelGGL(k_coef_to_i32, dim3(grid_for(n * 24, 4096)), dim3(256), 0, s, coef, planes, n * 24, bound, viol);
}
__global__ void __launch_bounds__(256) k_i32_to_coef(const int32_t *planes, u64 *coef, size_t total) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < total; i += st) coef[i] = fq_from_i64(planes[i]);
}
void launch_i32_to_coef(const int32_t *planes, u64 *coef, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_i32_to_coef, dim3(grid_for(n * 24, 4096)), dim3(256), 0, s, planes, coef, n * 24);
}
__global__ void __launch_bounds__(256) k_linf(const u64 *coef, size_t total, unsigned long long *out_max) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    u64 mx = 0;
    for (; i < total; i += st) {
        u64 v = coef[i];
        u64 mag = v > (LF_P - 1) / 2 ? LF_P - v : v;
        mx = mag > mx ? mag : mx;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        u64 o = __shfl_down((unsigned long long)mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_max, (unsigned long long)mx);
}
void launch_linf(const u64 *coef, size_t n, u64 *out_max, hipStream_t s) {
    (void)hipMemsetAsync(out_max, 0, 8, s);
    hipLaunchKernelGGL(k_linf, dim3(grid_for(n * 24, 4096)), dim3(256), 0, s, coef, n * 24, (unsigned long long *)out_max);
}

// bit-plane k of a centred small value: sign(v) * bit_k(|v|)   (base-2 balanced digits, decomposition.rs:159-167)
__device__ __forceinline__ int digit2(int32_t v, u32 k) {
    int32_t m = v < 0 ? -v : v;
    int d = (m >> k) & 1;
    return v < 0 ? -d : d;
}

struct BPow { u64 v[8]; };
__global__ void __launch_bounds__(256) k_recompose_crt(DevCrt t, const int32_t *planes, size_t n_planes, u32 wit_len, u32 L, BPow bp,
                                                        u32 K, int mode_bits, u64 *out, size_t ldz, size_t off) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 k = blockIdx.y;
    if (i >= wit_len) return;
    u64 a[24];
#pragma unroll
    for (int c = 0; c < 24; c++) {
        u64 acc = 0;
        for (u32 l = 0; l < L; l++) {
            int32_t v = planes[(size_t)c * n_planes + i * L + l];
            if (mode_bits) {
                int d = digit2(v, k);
                if (d > 0) acc = fq_add(acc, bp.v[l]);
                else if (d < 0) acc = fq_sub(acc, bp.v[l]);
            } else {
                u64 mag = (u64)(v < 0 ? -v : v);
                u64 term = fq_mul(bp.v[l], mag);
                acc = v < 0 ? fq_sub(acc, term) : fq_add(acc, term);
            }
        }
        a[c] = acc;
    }
    crt_store(a, out + (size_t)k * 24 * ldz, ldz, off + i, t);
}
// bit-plane mode with L = 4 and B^3 < 2^61: the recomposed coefficient sum_l digit_l B^l is an exact signed 64-bit integer (one
// conversion to a canonical residue instead of four conditional modular additions), the four digits' plane entries come in one 16-byte load
struct BInt4 { long long v[4]; };
// thread = (element i, residue class u of the coefficient index), as in k_bitplane_crt: the 8 x 4 plane entries it needs are loaded
// ONCE and all K bit-planes are produced from registers (the planes are read once per launch instead of K times: at 2^20 rows
// 2.40 -> 0.92 GB of traffic per call)
__global__ void __launch_bounds__(256) k_recompose_crt_b4(DevCrt t, const int32_t *planes, size_t n_planes, u32 wit_len, BInt4 bi, u32 K,
                                                           u64 *out, size_t ldz, size_t off) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 u = blockIdx.y;
    if (i >= wit_len) return;
    const size_t jj = off + i;
    int4 w[8];
#pragma unroll
    for (int v8 = 0; v8 < 8; v8++) w[v8] = *(const int4 *)(planes + (size_t)(3 * v8 + u) * n_planes + 4 * i);
    int plane[8];
    u64 tw[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        int s3 = 3 * t.slot_of_pos[p];
        plane[p] = u == 0 ? s3 : (u == 1 ? s3 + t.pos1[p] : s3 + t.pos2[p]);
        tw[p] = u == 1 ? t.tw1[p] : t.tw2[p];
    }
    for (u32 k = 0; k < K; k++) {
        u64 x[8], A[8];
#pragma unroll
        for (int v8 = 0; v8 < 8; v8++) {
            const int32_t v[4] = {w[v8].x, w[v8].y, w[v8].z, w[v8].w};
            long long sacc = 0;
#pragma unroll
            for (int l = 0; l < 4; l++) {
                int d = digit2(v[l], k);
                sacc += d > 0 ? bi.v[l] : (d < 0 ? -bi.v[l] : 0ll);
            }
            x[v8] = sacc < 0 ? LF_P - (u64)(-sacc) : (u64)sacc;
        }
        crt8(x, A, t);
        u64 *o = out + (size_t)k * 24 * ldz;
#pragma unroll
        for (int p = 0; p < 8; p++) o[(size_t)plane[p] * ldz + jj] = u == 0 ? A[p] : fq_mul(tw[p], A[p]);
    }
}
void launch_recompose_crt(const DevCrt &t, const int32_t *planes, size_t n_planes, u32 wit_len, u32 L, u64 B, u32 K, int mode_bits,
                          u64 *out, size_t ldz, size_t off, hipStream_t s) {
    if (mode_bits && L == 4 && B < ((u64)1 << 20) && (n_planes & 3) == 0 && ((uintptr_t)planes & 15) == 0) {
        BInt4 bi;
        bi.v[0] = 1;
        for (int l = 1; l < 4; l++) bi.v[l] = bi.v[l - 1] * (long long)B;
        hipLaunchKernelGGL(k_recompose_crt_b4, dim3(cdiv(wit_len, 256), 3), dim3(256), 0, s, t, planes, n_planes, wit_len, bi, K, out, ldz, off);
        return;
    }
    BPow bp;
    u64 pw = 1;
    for (int l = 0; l < 8; l++) { bp.v[l] = pw; pw = fq_mul(pw, B % LF_P); }
    hipLaunchKernelGGL(k_recompose_crt, dim3(cdiv(wit_len, 256), K), dim3(256), 0, s, t, planes, n_planes, wit_len, L, bp, K, mode_bits,
                       out, ldz, off);
}

__device__ __forceinline__ Fq3 ld3(const u64 *tab, size_t ld, u32 slot, size_t i) {
    return fq3_make(tab[(size_t)(3 * slot) * ld + i], tab[(size_t)(3 * slot + 1) * ld + i], tab[(size_t)(3 * slot + 2) * ld + i]);
}
__device__ __forceinline__ void st3(u64 *tab, size_t ld, u32 slot, size_t i, Fq3 v) {
    tab[(size_t)(3 * slot) * ld + i] = v.c[0]; tab[(size_t)(3 * slot + 1) * ld + i] = v.c[1]; tab[(size_t)(3 * slot + 2) * ld + i] = v.c[2];
}

// ---------------------------------------------------------------------------------------------------------
// Lazy F_{p^3} inner products: the five schoolbook column sums of a product as un-reduced 160-bit accumulators over a whole range, one reduction per output.
// (The Ajtai commitments themselves run on the int8 matrix cores: lf_ajtai_i8.hip, lf_ajtai_i8g.hip.)
struct Acc5 { AccP s[5]; };  // the five schoolbook column sums of an F_{p^3} product, un-reduced
__device__ __forceinline__ void acc5_zero(Acc5 &a) {
#pragma unroll
    for (int i = 0; i < 5; i++) accp_zero(a.s[i]);
}
__device__ __forceinline__ void acc5_mac(Acc5 &a, const u64 x[3], const u64 y[3]) {
    accp_mad(a.s[0], x[0], y[0]);
    accp_mad(a.s[1], x[0], y[1]); accp_mad(a.s[1], x[1], y[0]);
    accp_mad(a.s[2], x[0], y[2]); accp_mad(a.s[2], x[1], y[1]); accp_mad(a.s[2], x[2], y[0]);
    accp_mad(a.s[3], x[1], y[2]); accp_mad(a.s[3], x[2], y[1]);
    accp_mad(a.s[4], x[2], y[2]);
}
template <bool NU>
__device__ __forceinline__ Fq3 acc5_finish(const Acc5 &a, u64 nu) {
    Fq3 r;
    r.c[0] = fq_add(accp_reduce(a.s[0]), fq_mul_nu<NU>(accp_reduce(a.s[3]), nu));
    r.c[1] = fq_add(accp_reduce(a.s[1]), fq_mul_nu<NU>(accp_reduce(a.s[4]), nu));
    r.c[2] = accp_reduce(a.s[2]);
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// arithmetic self-test: the fast NU = 2^40 product, the lazy (L,H) accumulator and the partial-product accumulator
// against the generic schoolbook path on pseudo-random and edge operands; counts mismatching words.
__global__ void __launch_bounds__(256) k_selftest_field(u64 seed, u32 n, unsigned long long *mism) {
    u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 edge[8] = {0, 1, LF_P - 1, LF_P - 2, 0xFFFFFFFFULL, 0xFFFFFFFF00000000ULL, 1ULL << 32, (LF_P - 1) / 2};
    u64 w[6];
    for (int k = 0; k < 6; k++) {
        u64 v = splitmix_fq(seed, (u64)i * 6 + k);
        if (((v >> 7) & 3) == 0) v = edge[(v >> 3) & 7];  // a quarter of the operands are edge values
        w[k] = v;
    }
    Fq3 a = fq3_make(w[0], w[1], w[2]), b = fq3_make(w[3], w[4], w[5]);
    const u64 nu = 1ULL << 40;
    Fq3 ref = fq3_mul<false>(a, b, nu), fast = fq3_mul_2p40(a, b);
    unsigned bad = !fq3_eq(ref, fast);
    // lazy sums of 37 products (with repeated operands) vs reduced sums
    LH5 lz; lh5_zero(lz);
    Acc5 ap; acc5_zero(ap);
    Fq3 sum = fq3_zero();
    Fq3 x = a, y = b;
    for (int r = 0; r < 37; r++) {
        lh5_mac(lz, x, y);
        acc5_mac(ap, x.c, y.c);
        sum = fq3_add(sum, fq3_mul<false>(x, y, nu));
        Fq3 t = fq3_add(x, y); x = y; y = t;
    }
    bad += !fq3_eq(sum, lh5_finish(lz));
    bad += !fq3_eq(sum, acc5_finish<true>(ap, nu));
    bad += !fq3_eq(sum, acc5_finish<false>(ap, nu));
    if (bad) atomicAdd(mism, (unsigned long long)bad);
}
void launch_selftest_field(u64 seed, u32 n, u64 *mism_dev, hipStream_t s) {
    (void)hipMemsetAsync(mism_dev, 0, 8, s);
    hipLaunchKernelGGL(k_selftest_field, dim3(cdiv(n, 256)), dim3(256), 0, s, seed, n, (unsigned long long *)mism_dev);
}

// ---------------------------------------------------------------------------------------------------------
// eq(x, r) table, LSB-first (build_eq_x_r, utils/sumcheck/utils.rs:100-170): eq[i] = prod_j (i_j ? r_j : 1-r_j)
template <bool NU>
__global__ void __launch_bounds__(256) k_build_eq(DevCrt t, const Fq3Const *r, u32 nv, u64 *eq) {
    __shared__ u64 sr[64][2][3];
    for (u32 idx = threadIdx.x; idx < nv * 3; idx += 256) {
        u32 j = idx / 3, c = idx % 3;
        u64 rv = r[j].c[c];
        sr[j][1][c] = rv;
        sr[j][0][c] = fq_sub(c == 0 ? 1 : 0, rv);
    }
    __syncthreads();
    size_t n = (size_t)1 << nv;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fq3 acc = fq3_one();
    for (u32 j = 0; j < nv; j++) {
        u32 b = (i >> j) & 1;
        acc = M3<NU>(acc, fq3_make(sr[j][b][0], sr[j][b][1], sr[j][b][2]), t.nu);
    }
    eq[i] = acc.c[0]; eq[n + i] = acc.c[1]; eq[2 * n + i] = acc.c[2];
}
void launch_build_eq(const DevCrt &t, const Fq3Const *r_dev, u32 nv, u64 *eq, hipStream_t s) {
    LF_LAUNCH(k_build_eq, t.nu2p40, dim3(cdiv((size_t)1 << nv, 256)), dim3(256), s, t, r_dev, nv, eq);
}
// eq(r, i) = eq(r_lo, i mod 2^hl) * eq(r_hi, i >> hl): one product per entry instead of nv (field arithmetic is exact, so the
// table is identical to k_build_eq's)
template <bool NU>
__global__ void __launch_bounds__(256) k_eq_outer(DevCrt t, const u64 *lo, u32 hl, const u64 *hi, u32 hh, u64 *eq) {
    size_t n = (size_t)1 << (hl + hh), nl = (size_t)1 << hl, nh = (size_t)1 << hh;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    size_t a = i & (nl - 1), b = i >> hl;
    Fq3 r = M3<NU>(fq3_make(lo[a], lo[nl + a], lo[2 * nl + a]), fq3_make(hi[b], hi[nh + b], hi[2 * nh + b]), t.nu);
    eq[i] = r.c[0]; eq[n + i] = r.c[1]; eq[2 * n + i] = r.c[2];
}
size_t build_eq_scratch_words(u32 nv) { u32 hl = nv / 2; return 3 * (((size_t)1 << hl) + ((size_t)1 << (nv - hl))); }
void launch_build_eq2(const DevCrt &t, const Fq3Const *r_dev, u32 nv, u64 *scratch, u64 *eq, hipStream_t s) {
    u32 hl = nv / 2, hh = nv - hl;
    u64 *lo = scratch, *hi = scratch + 3 * ((size_t)1 << hl);
    LF_LAUNCH(k_build_eq, t.nu2p40, dim3(cdiv((size_t)1 << hl, 256)), dim3(256), s, t, r_dev, hl, lo);
    LF_LAUNCH(k_build_eq, t.nu2p40, dim3(cdiv((size_t)1 << hh, 256)), dim3(256), s, t, r_dev + hl, hh, hi);
    LF_LAUNCH(k_eq_outer, t.nu2p40, dim3(cdiv((size_t)1 << nv, 256)), dim3(256), s, t, lo, hl, hi, hh, eq);
}

// mat_vec_mul (arith/utils.rs:52-65) on CSR.  (The thread mapping of k_spmv_t_eq -- the slots of a row side by side -- was measured here too: the gathers of z then come in
// 64-byte pieces and the kernel takes 163 instead of 120 us at 2^20 rows.)
template <bool NU>
__global__ void __launch_bounds__(256) k_spmv(DevCrt t, const u32 *rowptr, const u32 *col, const u64 *val, const u64 *z, size_t ldz,
                                              u64 *out, size_t m, int accumulate, size_t r0, size_t rcnt) {
    size_t row = r0 + (size_t)blockIdx.x * 256 + threadIdx.x;   // rows [r0, r0 + rcnt): a sharded rank's slice of the output table (global layout)
    u32 slot = blockIdx.y;
    if (row >= r0 + rcnt) return;
    Fq3 acc = accumulate ? ld3(out, m, slot, row) : fq3_zero();
    for (u32 k = rowptr[row]; k < rowptr[row + 1]; k++) {
        const u64 *v = val + (size_t)k * 24 + 3 * slot;
        acc = fq3_add(acc, M3<NU>(fq3_make(v[0], v[1], v[2]), ld3(z, ldz, slot, col[k]), t.nu));
    }
    st3(out, m, slot, row, acc);
}
void launch_spmv(const DevCrt &t, const u32 *rowptr, const u32 *col, const u64 *val, const u64 *z, size_t ldz, u64 *out, size_t m,
                 int accumulate, hipStream_t s, size_t r0, size_t rcnt) {
    if (rcnt == (size_t)-1) { r0 = 0; rcnt = m; }
    if (!rcnt) return;
    LF_LAUNCH(k_spmv, t.nu2p40, dim3(cdiv(rcnt, 256), 8), dim3(256), s, t, rowptr, col, val, z, ldz, out, m, accumulate, r0, rcnt);
}
// out = sum_{j<nm} M_j z_j in one pass (fold prepare: G = sum_j M_j (sum_k zeta_k^{j+1} z_k)): one launch and one write of the
// output instead of nm launches that each read-modify-write it
struct SpmvSet { const u32 *rowptr[4]; const u32 *col[4]; const u64 *val[4]; const u64 *z[4]; u32 nm; };
template <bool NU>
__global__ void __launch_bounds__(256) k_spmv_sum(DevCrt t, SpmvSet ms, size_t ldz, u64 *out, size_t m, size_t r0, size_t rcnt) {
    size_t row = r0 + (size_t)blockIdx.x * 256 + threadIdx.x;   // rows [r0, r0 + rcnt) of the m-row table
    u32 slot = blockIdx.y;
    if (row >= r0 + rcnt) return;
    Fq3 acc = fq3_zero();
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        if (j < ms.nm) {
            const u32 *rp = ms.rowptr[j], *cl = ms.col[j];
            for (u32 k = rp[row]; k < rp[row + 1]; k++) {
                const u64 *v = ms.val[j] + (size_t)k * 24 + 3 * slot;
                acc = fq3_add(acc, M3<NU>(fq3_make(v[0], v[1], v[2]), ld3(ms.z[j], ldz, slot, cl[k]), t.nu));
            }
        }
    }
    st3(out, m, slot, row, acc);
}
void launch_spmv_sum(const DevCrt &t, u32 nm, const u32 *const *rowptr, const u32 *const *col, const u64 *const *val, const u64 *z,
                     size_t z_stride, size_t ldz, u64 *out, size_t m, hipStream_t s, size_t r0, size_t rcnt) {
    if (rcnt == (size_t)-1) { r0 = 0; rcnt = m; }
    if (!rcnt) return;
    SpmvSet ms = {};
    ms.nm = nm;
    for (u32 j = 0; j < nm && j < 4; j++) { ms.rowptr[j] = rowptr[j]; ms.col[j] = col[j]; ms.val[j] = val[j]; ms.z[j] = z + (size_t)j * z_stride; }
    LF_LAUNCH(k_spmv_sum, t.nu2p40, dim3(cdiv(rcnt, 256), 8), dim3(256), s, t, ms, ldz, out, m, r0, rcnt);
}
// General matrices (several entries per row at arbitrary columns, ring-valued entries: arith/utils.rs:52-65 as a real CSR SpMV).  k_spmv / k_spmv_sum above are the
// shape of the reference's bench matrices (one entry per row, neighbouring rows at neighbouring columns): thread = (row, one slot per block), z plane-major.  With k
// entries per row at random columns that mapping reads 24 bytes out of every 192-byte entry, eight blocks over, and gathers every word of z from a line of its
// own: 4.84 ms per k_spmv_sum at 2^18 rows x 16 entries (0.7 TB/s of useful bytes).  Here block = 32 rows x 8 slots, the slots of a row side by side: a wave reads
// eight whole entries as one 1.5 KB run and gathers z as whole 192-byte ELEMENTS from an element-major copy zaos [n][24] (launch_soa_to_aos, once per vector);
// the sums cross LDS so that the stores are runs of 32 rows per output plane.  out = (accumulate ? out : 0) + sum_{j<nm} M_j z_j.
struct SpmvRowsSet { const u32 *rowptr[4]; const u32 *col[4]; const u64 *val[4]; const u64 *zaos[4]; u32 nm; };
template <bool NU>
__global__ void __launch_bounds__(256) k_spmv_rows(DevCrt t, SpmvRowsSet ms, u64 *out, size_t m, int accumulate, size_t r0, size_t rcnt) {
    const u32 rl = threadIdx.x >> 3, slot = threadIdx.x & 7;
    const size_t rb = r0 + (size_t)blockIdx.x * 32, row = rb + rl;
    __shared__ u64 sm[24][33];
    Fq3 acc = fq3_zero();
    if (row < r0 + rcnt) {
#pragma unroll
        for (u32 j = 0; j < 4; j++) {
            if (j < ms.nm) {
                const u32 *rp = ms.rowptr[j], *cl = ms.col[j];
                const u64 *za = ms.zaos[j] + 3 * slot;
                for (u32 k = rp[row]; k < rp[row + 1]; k++) {
                    const u64 *v = ms.val[j] + (size_t)k * 24 + 3 * slot, *zz = za + (size_t)cl[k] * 24;
                    acc = fq3_add(acc, M3<NU>(fq3_make(v[0], v[1], v[2]), fq3_make(zz[0], zz[1], zz[2]), t.nu));
                }
            }
        }
    }
    sm[3 * slot][rl] = acc.c[0]; sm[3 * slot + 1][rl] = acc.c[1]; sm[3 * slot + 2][rl] = acc.c[2];
    __syncthreads();
    for (u32 o = threadIdx.x; o < 24 * 32; o += 256) {
        const u32 pl = o >> 5, rr = o & 31;
        if (rb + rr < r0 + rcnt) {
            u64 *dst = out + (size_t)pl * m + rb + rr;
            *dst = accumulate ? fq_add(*dst, sm[pl][rr]) : sm[pl][rr];
        }
    }
}
// z: nm vectors [24][ldz] plane-major, z_stride words apart (n columns each); zaos: scratch of nm * n * 24 words
void launch_spmv_rows(const DevCrt &t, u32 nm, const u32 *const *rowptr, const u32 *const *col, const u64 *const *val, const u64 *z, size_t z_stride, size_t n,
                      u64 *zaos, u64 *out, size_t m, int accumulate, hipStream_t s, size_t r0, size_t rcnt) {
    if (rcnt == (size_t)-1) { r0 = 0; rcnt = m; }
    if (!rcnt || !nm) return;
    SpmvRowsSet ms = {};
    ms.nm = nm;
    for (u32 j = 0; j < nm && j < 4; j++) {
        u64 *za = zaos + (size_t)j * n * 24;
        if (z) launch_soa_to_aos(z + (size_t)j * z_stride, za, n, s);        // (z null: zaos holds the element-major copies already)
        ms.rowptr[j] = rowptr[j]; ms.col[j] = col[j]; ms.val[j] = val[j]; ms.zaos[j] = za;
    }
    LF_LAUNCH(k_spmv_rows, t.nu2p40, dim3(cdiv(rcnt, 32)), dim3(256), s, t, ms, out, m, accumulate, r0, rcnt);
}
// block = 32 columns x 8 slots, the slots of a column side by side: a wave reads the 24 coefficient words of eight non-zeros as one contiguous 1.5 KB run (with
// thread = column and one slot per block a load instruction touched 64 cache lines for 24 bytes each, eight blocks re-reading them: 63 us for 100 MB at 2^18
// columns), the eq words of a row once for its eight slots; the sums cross LDS so that the stores are runs of 32 columns per output row.
template <bool NU>
__global__ void __launch_bounds__(256) k_spmv_t_eq(DevCrt t, const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t m,
                                                   u64 *q, size_t n, size_t c0, size_t ccnt) {
    const u32 cl = threadIdx.x >> 3, slot = threadIdx.x & 7;
    const size_t cb = c0 + (size_t)blockIdx.x * 32, c = cb + cl;   // columns [c0, c0 + ccnt): the slice a sharded rank's inner products read
    __shared__ u64 sm[24][33];
    Fq3 acc = fq3_zero();
    if (c < c0 + ccnt)
        for (u32 k = colptr[c]; k < colptr[c + 1]; k++) {
            const u64 *v = val + (size_t)k * 24 + 3 * slot;
            size_t r = rowidx[k];
            acc = fq3_add(acc, M3<NU>(fq3_make(v[0], v[1], v[2]), fq3_make(eq[r], eq[m + r], eq[2 * m + r]), t.nu));
        }
    sm[3 * slot][cl] = acc.c[0]; sm[3 * slot + 1][cl] = acc.c[1]; sm[3 * slot + 2][cl] = acc.c[2];
    __syncthreads();
    for (u32 o = threadIdx.x; o < 24 * 32; o += 256) {
        const u32 row = o >> 5, cc = o & 31;
        if (cb + cc < c0 + ccnt) q[(size_t)row * n + cb + cc] = sm[row][cc];
    }
}
void launch_spmv_t_eq(const DevCrt &t, const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t m, u64 *q, size_t n,
                      hipStream_t s, size_t c0, size_t ccnt) {
    if (ccnt == (size_t)-1) { c0 = 0; ccnt = n; }
    if (!ccnt) return;
    LF_LAUNCH(k_spmv_t_eq, t.nu2p40, dim3(cdiv(ccnt, 32)), dim3(256), s, t, colptr, rowidx, val, eq, m, q, n, c0, ccnt);
}

// ---------------------------------------------------------------------------------------------------------
// batched inner products (evaluate_mles, utils/mle_helpers.rs:65-88, restructured as dot products)
constexpr u32 RED_BLOCKS = 256;
constexpr u32 DOT_NA_MAX = 32;   // left-hand tables of k_dot_batch (K <= 32 bit-planes)
// NB = number of Y tables (compile time: the accumulators of unused tables would otherwise cost a wave of occupancy)
template <bool NU, int NB>
__global__ void __launch_bounds__(256) k_dot_batch(DevCrt t, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n,
                                                   u64 *partial) {
    // grid (8 slots, na, column blocks); each block streams its X_a once against all nb <= 4 tables Y_b.  The linear workgroup id is
    // slot + 8 * (a + na * block): the na blocks that read the same slice of Y run back to back on one XCD and share it in that L2
    // (with the column block as the fastest index every a re-read Y from HBM: 3.4 GB fetched for 0.96 GB of tables).
    const u32 slot = blockIdx.x, a = blockIdx.y, bx = blockIdx.z, nbx = gridDim.z;
    LH5 acc[NB];
    Fq3 accg[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) { lh5_zero(acc[b]); accg[b] = fq3_zero(); }
    const u64 *Xa = X + (size_t)a * 24 * ldx;
    for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n; i += (size_t)nbx * 256) {
        Fq3 x = ld3(Xa, ldx, slot, i);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            Fq3 y = ld3(Y + (size_t)b * 24 * ldy, ldy, slot, i);
            if (NU) lh5_mac(acc[b], x, y);
            else accg[b] = fq3_add(accg[b], M3<NU>(x, y, t.nu));
        }
    }
    u64 v[3 * NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        Fq3 r = NU ? lh5_finish(acc[b]) : accg[b];
        v[3 * b] = r.c[0]; v[3 * b + 1] = r.c[1]; v[3 * b + 2] = r.c[2];
    }
    // partial[block][ (a*nb + b)*24 + 3*slot + c ]
    __shared__ u64 red[3 * NB];
    block_sum_store<3 * NB>(v, red);
    __syncthreads();
    if (threadIdx.x < 3 * NB) {
        u32 b = threadIdx.x / 3, c = threadIdx.x % 3;
        if (b < nb) partial[(size_t)bx * (DOT_NA_MAX * nb * 24) + ((size_t)a * nb + b) * 24 + 3 * slot + c] = red[threadIdx.x];
    }
}
size_t dot_partial_words(u32 na, u32 nb) { return (size_t)RED_BLOCKS * DOT_NA_MAX * nb * 24; }
void launch_dot_batch(const DevCrt &t, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n, u64 *partial,
                      u64 *out, hipStream_t s) {
    u32 gb = (u32)((n + 255) / 256);
    if (gb > 64) gb = 64;   // fatter threads: the 12-value block reduction per block is not free
    if (gb < 1) gb = 1;
#define LF_DB(N)                                                                                                                           \
    do {                                                                                                                                \
        if (t.nu2p40) hipLaunchKernelGGL((k_dot_batch<true, N>), dim3(8, na, gb), dim3(256), 0, s, t, X, ldx, na, Y, ldy, nb, n, partial);        \
        else hipLaunchKernelGGL((k_dot_batch<false, N>), dim3(8, na, gb), dim3(256), 0, s, t, X, ldx, na, Y, ldy, nb, n, partial);                \
    } while (0)
    switch (nb) {
        case 1: LF_DB(1); break;
        case 2: LF_DB(2); break;
        case 3: LF_DB(3); break;
        default: LF_DB(4); break;
    }
#undef LF_DB
    hipLaunchKernelGGL(k_reduce_rows, dim3(na * nb * 24), dim3(256), 0, s, partial, gb, DOT_NA_MAX * nb * 24, out);
}
template <bool NU>
__global__ void __launch_bounds__(256) k_dot_eq(DevCrt t, const u64 *X, size_t ldx, const u64 *eq, size_t ldeq, size_t n, u64 *partial) {
    // grid (RED_BLOCKS, 8 slots, na)
    u32 slot = blockIdx.y, a = blockIdx.z, na = gridDim.z;
    Acc5 acc;
    acc5_zero(acc);
    const u64 *Xa = X + (size_t)a * 24 * ldx;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        Fq3 x = ld3(Xa, ldx, slot, i);
        u64 e[3] = {eq[i], eq[ldeq + i], eq[2 * ldeq + i]};
        acc5_mac(acc, x.c, e);
    }
    Fq3 r = acc5_finish<NU>(acc, t.nu);
    u64 v[3] = {r.c[0], r.c[1], r.c[2]};
    block_sum_store<3>(v, partial + (size_t)blockIdx.x * (na * 24) + (size_t)a * 24 + 3 * slot);
}
void launch_dot_eq(const DevCrt &t, const u64 *X, size_t ldx, u32 na, const u64 *eq, size_t ldeq, size_t n, u64 *partial, u64 *out,
                   hipStream_t s) {
    u32 gb = (u32)((n + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    LF_LAUNCH(k_dot_eq, t.nu2p40, dim3(gb, 8, na), dim3(256), s, t, X, ldx, eq, ldeq, n, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3(na * 24), dim3(256), 0, s, partial, gb, na * 24, out);
}

// f-hat evaluations without materialising f-hat (Witness::get_fhat, arith.rs:273-297, is a re-layout of
// f_coeff): T[k][c] = sum_i eq[i] * digit_k(f[i][c]);  v_d slot s = T[k][8d+s].
// KG bit-planes per thread: the kernel is bound by the cache traffic of eq (every (coefficient, plane-group) block streams it
// again), so one thread takes all K <= 16 planes of its coefficient and eq is read once per (coefficient, element).
template <int KG>
__global__ void __launch_bounds__(256) k_coef_eval(const int32_t *planes, size_t ldp, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits,
                                                   u64 *partial) {
    // grid (RED_BLOCKS, 24 coefficients, K-groups of KG bit-planes)
    u32 c = blockIdx.y, kg = blockIdx.z * KG;
    u64 acc[3 * KG];
    u32 cy[3 * KG];   // mode_bits: lazy 64-bit sums with carry counters (add, add-with-carry, count) instead of a modular add per term
#pragma unroll
    for (int i = 0; i < 3 * KG; i++) { acc[i] = 0; cy[i] = 0; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int32_t v = planes[(size_t)c * ldp + i];
        u64 e[3] = {eq[i], eq[ldeq + i], eq[2 * ldeq + i]};
        bool neg = v < 0;
        u32 mg = (u32)(neg ? -v : v);
        if (mode_bits) {
            // +-eq[i] selected by the sign once, then masked (branch-free) adds per bit-plane
#pragma unroll
            for (int q = 0; q < 3; q++) e[q] = neg ? fq_neg(e[q]) : e[q];
#pragma unroll
            for (int k = 0; k < KG; k++) {
                u64 mask = (u64)0 - (u64)((mg >> (kg + k)) & 1);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    u64 tq = e[q] & mask, sum = acc[3 * k + q] + tq;
                    cy[3 * k + q] += sum < tq;
                    acc[3 * k + q] = sum;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                u64 tq = fq_mul(e[q], (u64)mg);
                acc[q] = neg ? fq_sub(acc[q], tq) : fq_add(acc[q], tq);
            }
        }
    }
    if (mode_bits) {
#pragma unroll
        for (int i = 0; i < 3 * KG; i++) acc[i] = fq_canon(fq_reduce128_loose(acc[i], (u64)cy[i]));   // lo + 2^64 * carries
    }
    // partial[block][k][c][3]
    __shared__ u64 red[3 * KG];
    block_sum_store<3 * KG>(acc, red);
    __syncthreads();
    if (threadIdx.x < 3 * KG) {
        u32 k = kg + threadIdx.x / 3, q = threadIdx.x % 3;
        if (k < K) partial[(size_t)blockIdx.x * (K * 72) + ((size_t)k * 24 + c) * 3 + q] = red[threadIdx.x];
    }
}
size_t coef_eval_partial_words(u32 K) { return (size_t)RED_BLOCKS * K * 72; }
void launch_coef_eval(const DevCrt &t, const int32_t *planes, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits, u64 *partial,
                      u64 *out, hipStream_t s, size_t ldp) {
    if (!ldp) ldp = n;
    u32 gb = (u32)((n + 255) / 256);
    if (gb > 64) gb = 64;   // >= 64 elements per thread at 2^20: the 12-value block reduction is a third of the work otherwise
    if (gb < 1) gb = 1;
    static int kg = -1;
    if (kg < 0) { const char *e = getenv("LF_COEF_KG"); kg = e ? atoi(e) : 8; }   // planes per thread: 8 halves the HBM re-reads of the 4-plane version at the same speed (16 spills into the latency chain: slower)
    if (mode_bits && kg == 16) hipLaunchKernelGGL(k_coef_eval<16>, dim3(gb, 24, (K + 15) / 16), dim3(256), 0, s, planes, ldp, n, eq, ldeq, K, mode_bits, partial);
    else if (mode_bits && kg == 8) hipLaunchKernelGGL(k_coef_eval<8>, dim3(gb, 24, (K + 7) / 8), dim3(256), 0, s, planes, ldp, n, eq, ldeq, K, mode_bits, partial);
    else hipLaunchKernelGGL(k_coef_eval<4>, dim3(gb, 24, mode_bits ? (K + 3) / 4 : 1), dim3(256), 0, s, planes, ldp, n, eq, ldeq, K, mode_bits, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3(K * 72), dim3(256), 0, s, partial, gb, K * 72, out);
}

template <bool NU, int TT>
__global__ void __launch_bounds__(256) k_lincomb_z(DevCrt t, const u64 *z, size_t ldz, u32 K, const Fq3Const *coef, size_t n, u64 *out, u32 per_slot) {
    // TT output tables (compile time: only the accumulators that are used occupy registers)
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 slot = blockIdx.y;
    if (i >= n) return;
    LH5 acc[TT];
    Fq3 accg[TT];
#pragma unroll
    for (int j = 0; j < TT; j++) { lh5_zero(acc[j]); accg[j] = fq3_zero(); }
    auto cf = [&](u32 k, int j) {
        Fq3Const cc = coef[per_slot ? (size_t)(k * TT + j) * 8 + slot : (size_t)(k * TT + j)];   // per_slot: ring-element coefficients
        return fq3_make(cc.c[0], cc.c[1], cc.c[2]);
    };
    u32 k = 0;
    if (NU) {
        for (; k + 3 < K; k += 4) {      // four terms per fold of the column sums (lh5_macn): the fold is the larger half of a lazy product
            Fq3 x[4], cv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = ld3(z + (size_t)(k + q) * 24 * ldz, ldz, slot, i);
#pragma unroll
            for (int j = 0; j < TT; j++) {
#pragma unroll
                for (int q = 0; q < 4; q++) cv[q] = cf(k + q, j);
                lh5_macn<4>(acc[j], x, cv);
            }
        }
        for (; k + 1 < K; k += 2) {
            Fq3 x0 = ld3(z + (size_t)k * 24 * ldz, ldz, slot, i), x1 = ld3(z + (size_t)(k + 1) * 24 * ldz, ldz, slot, i);
#pragma unroll
            for (int j = 0; j < TT; j++) lh5_mac2(acc[j], x0, cf(k, j), x1, cf(k + 1, j));
        }
    }
    for (; k < K; k++) {
        Fq3 x = ld3(z + (size_t)k * 24 * ldz, ldz, slot, i);
#pragma unroll
        for (int j = 0; j < TT; j++) {
            Fq3 cv = cf(k, j);
            if (NU) lh5_mac(acc[j], x, cv);
            else accg[j] = fq3_add(accg[j], M3<NU>(x, cv, t.nu));
        }
    }
#pragma unroll
    for (int j = 0; j < TT; j++) st3(out + (size_t)j * 24 * ldz, ldz, slot, i, NU ? lh5_finish(acc[j]) : accg[j]);
}
void launch_lincomb_z(const DevCrt &t, const u64 *z, size_t ldz, u32 K, const Fq3Const *coef_dev, u32 tt, size_t n, u64 *out, hipStream_t s, u32 per_slot) {
    if (!n) return;
#define LF_LZ(N)                                                                                                                              \
    do {                                                                                                                                      \
        if (t.nu2p40) hipLaunchKernelGGL((k_lincomb_z<true, N>), dim3(cdiv(n, 256), 8), dim3(256), 0, s, t, z, ldz, K, coef_dev, n, out, per_slot);    \
        else hipLaunchKernelGGL((k_lincomb_z<false, N>), dim3(cdiv(n, 256), 8), dim3(256), 0, s, t, z, ldz, K, coef_dev, n, out, per_slot);            \
    } while (0)
    switch (tt) {
        case 1: LF_LZ(1); break;
        case 2: LF_LZ(2); break;
        case 3: LF_LZ(3); break;
        default: LF_LZ(4); break;   // callers keep tt <= 4
    }
#undef LF_LZ
}

// Nibble tables: sum_k apow[k][d] * digit_k(v) = sign(v) * sum_q T[d][q][(|v| >> 4q) & 15], T[d][q][val] = sum_{b<4, bit b of val} apow[4q+b][d]
// (192 F_{p^3} values built in LDS per block): 12 look-ups and additions per row and slot instead of a 48-iteration bit loop.
template <int NQ>   // nibbles of |v|: 4 for K <= 16 bit-planes, 8 for K <= 32
__global__ void __launch_bounds__(256) k_add_fhat_comb(const int32_t *planes, size_t n_planes, u32 K, const Fq3Const *apow, u64 *G, size_t m, size_t r0, size_t r1) {
    __shared__ u64 T[3 * NQ * 16][4];
    for (u32 e = threadIdx.x; e < 3 * NQ * 16; e += 256) {
        u32 val = e % 16, q = (e / 16) % NQ, d = e / (16 * NQ);
        Fq3 sum = fq3_zero();
        for (u32 b = 0; b < 4; b++)
            if (4 * q + b < K && ((val >> b) & 1)) {
                Fq3Const a = apow[(4 * q + b) * 3 + d];
                sum = fq3_add(sum, fq3_make(a.c[0], a.c[1], a.c[2]));
            }
        T[e][0] = sum.c[0]; T[e][1] = sum.c[1]; T[e][2] = sum.c[2]; T[e][3] = 0;
    }
    __syncthreads();
    size_t row = r0 + (size_t)blockIdx.x * 256 + threadIdx.x;   // positions [r0, r1)
    u32 slot = blockIdx.y;
    if (row >= r1) return;
    Fq3 acc = ld3(G, m, slot, row);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int32_t v = planes[(size_t)(8 * d + slot) * n_planes + row];
        u32 mg = (u32)(v < 0 ? -v : v);
        Fq3 part = fq3_zero();
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const u64 *e = T[(d * NQ + q) * 16 + ((mg >> (4 * q)) & 15)];
            part = fq3_add(part, fq3_make(e[0], e[1], e[2]));
        }
        acc = v < 0 ? fq3_sub(acc, part) : fq3_add(acc, part);
    }
    st3(G, m, slot, row, acc);
}
void launch_add_fhat_comb(const DevCrt &t, const int32_t *planes, size_t n_planes, u32 K, const Fq3Const *apow_dev, u64 *G, size_t m,
                          hipStream_t s, size_t r0, size_t rcnt) {
    size_t r1 = rcnt == (size_t)-1 ? n_planes : (r0 + rcnt < n_planes ? r0 + rcnt : n_planes);   // positions [r0, r1) of the n_planes the witness covers
    if (rcnt == (size_t)-1) r0 = 0;
    if (r1 <= r0) return;
    if (K <= 16) hipLaunchKernelGGL(k_add_fhat_comb<4>, dim3(cdiv(r1 - r0, 256), 8), dim3(256), 0, s, planes, n_planes, K, apow_dev, G, m, r0, r1);
    else hipLaunchKernelGGL(k_add_fhat_comb<8>, dim3(cdiv(r1 - r0, 256), 8), dim3(256), 0, s, planes, n_planes, K, apow_dev, G, m, r0, r1);
}

// ---------------------------------------------------------------------------------------------------------
// fix_variables (DenseMultilinearExtension, sumcheck/prover.rs:70-72): new[j] = old[2j] + r*(old[2j+1]-old[2j])
// `rows3` independent F_{p^3} rows (8 per ring table, 1 per eq table), planes of leading dimension ld.
template <bool NU>
__global__ void __launch_bounds__(256) k_fix(DevCrt t, const u64 *in, size_t ld_in, u64 *out, size_t ld_out, size_t n_out, Fq3Const r) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t row = blockIdx.y;
    if (j >= n_out) return;
    const u64 *p = in + row * 3 * ld_in;
    Fq3 rr = fq3_make(r.c[0], r.c[1], r.c[2]);
    ulonglong2 a0 = *(const ulonglong2 *)(p + 2 * j), a1 = *(const ulonglong2 *)(p + ld_in + 2 * j), a2 = *(const ulonglong2 *)(p + 2 * ld_in + 2 * j);
    Fq3 v0 = fq3_make(a0.x, a1.x, a2.x), v1 = fq3_make(a0.y, a1.y, a2.y);
    Fq3 res = fq3_add(v0, M3<NU>(fq3_sub(v1, v0), rr, t.nu));
    u64 *q = out + row * 3 * ld_out;
    q[j] = res.c[0]; q[ld_out + j] = res.c[1]; q[2 * ld_out + j] = res.c[2];
}
void launch_fix_ring(const DevCrt &t, const u64 *in, u64 *out, size_t n_in, Fq3Const r, hipStream_t s) {
    size_t n_out = n_in / 2;
    LF_LAUNCH(k_fix, t.nu2p40, dim3(cdiv(n_out, 256), 8), dim3(256), s, t, in, n_in, out, n_out, n_out, r);
}
void launch_fix_fq3(const DevCrt &t, const u64 *in, u64 *out, size_t n_in, Fq3Const r, hipStream_t s) {
    size_t n_out = n_in / 2;
    LF_LAUNCH(k_fix, t.nu2p40, dim3(cdiv(n_out, 256), 1), dim3(256), s, t, in, n_in, out, n_out, n_out, r);
}
// many tables at once: tables [ntab][24][ld] -> [ntab][24][ld/2]
void launch_fix_many(const DevCrt &t, const u64 *in, size_t ld_in, u64 *out, size_t ld_out, size_t n_in, u32 rows3, Fq3Const r, hipStream_t s) {
    size_t n_out = n_in / 2;
    LF_LAUNCH(k_fix, t.nu2p40, dim3(cdiv(n_out, 256), rows3), dim3(256), s, t, in, ld_in, out, ld_out, n_out, r);
}

// last fix of the folding sumcheck's f-hat tables (2 entries per row): the fully fixed tables ARE the evaluations theta = f-hat(r_o)
// that folding.rs:236-242 recomputes with evaluate_mles; canonical words, laid out [row][3] = theta's flat order.
template <bool NU>
__global__ void __launch_bounds__(256) k_fix_final(DevCrt t, const u64 *in, u32 rows3, Fq3Const r, u64 *out) {
    u32 row = blockIdx.x * 256 + threadIdx.x;
    if (row >= rows3) return;
    const u64 *p = in + (size_t)row * 6;
    Fq3 rr = fq3_make(r.c[0], r.c[1], r.c[2]);
    Fq3 v0 = fq3_make(p[0], p[2], p[4]), v1 = fq3_make(p[1], p[3], p[5]);
    Fq3 res = fq3_add(v0, M3<NU>(fq3_sub(v1, v0), rr, t.nu));
#pragma unroll
    for (int q = 0; q < 3; q++) out[(size_t)row * 3 + q] = fq_canon(res.c[q]);
}
// v = sum_k 2^k v_s[k]: the MLE evaluation of the witness coefficients from the evaluations of their K binary digit planes
// (linearization.rs:126-139 v and decomposition.rs:204-211 v_s at the same point are the same sums)
__global__ void __launch_bounds__(128) k_vs_combine(const u64 *vs, u32 K, u64 *v) {
    const u32 i = threadIdx.x;
    if (i >= 72) return;
    u64 acc = 0, pw = 1;
    for (u32 k = 0; k < K; k++) {
        acc = fq_add(acc, fq_mul(fq_canon(vs[(size_t)k * 72 + i]), pw));
        pw = fq_add(pw, pw);
    }
    v[i] = fq_canon(acc);
}
void launch_vs_combine(const u64 *vs, u32 K, u64 *v, hipStream_t s) { hipLaunchKernelGGL(k_vs_combine, dim3(1), dim3(128), 0, s, vs, K, v); }
void launch_fix_final(const DevCrt &t, const u64 *in, u32 rows3, Fq3Const r, u64 *out, hipStream_t s) {
    LF_LAUNCH(k_fix_final, t.nu2p40, dim3(cdiv(rows3, 256)), dim3(256), s, t, in, rows3, r, out);
}

// evaluate a quadratic/cubic given by coefficients at X = 0..deg and add into acc
template <int NP>
__device__ __forceinline__ void add_poly_evals(Fq3 (&acc)[NP], const Fq3 *co, int ncoef) {
#pragma unroll
    for (int X = 0; X < NP; X++) {
        Fq3 v = co[ncoef - 1];
        for (int e = ncoef - 2; e >= 0; e--) v = fq3_add(fq3_mul_small(v, X), co[e]);
        acc[X] = fq3_add(acc[X], v);
    }
}

// ---------------------------------------------------------------------------------------------------------
// linearization sumcheck round (sumcheck/prover.rs:56-162 with comb = linearization/utils.rs:90-107)
// FUSED: fix_variables of the previous round's tables (mz / eq hold 2n entries per row, ld / ldeq their strides) with rfix happens here: pair p is built from
// the entries 4p..4p+3 and stored to mzo / eqo (n entries per row) for the next round -- no separate k_fix pass over the tables
struct LinFix { Fq3Const r; u64 *mzo; size_t ldo; u64 *eqo; size_t ldeo; };
// SPLIT (xmask != 0 at the launch): eq(beta, (r_1..r_{i-1}, X, x)) = c_i * eq(beta_i, X) * E_i[x] with E_i = eq((beta_{i+1}..beta_s), .), one entry per
// PAIR and no X in it -- the kernel sums E_i[p] * h(X, p) for the X of `xmask` only (the host multiplies by c_i eq(beta_i, X), derives the value at X = 1
// from the previous round's message and extrapolates the top one: exact field arithmetic, the same message words).  `eq` is then E_i (one entry per pair;
// FUSED: E_{i-1}, whose pair sums are E_i, stored through fx.eqo).  Half the products per pair of the plain form.
// c_i[3 slot ..] of the by-value descriptor, read from the kernel-argument segment itself (constant memory; the descriptor is the second argument of every kernel
// that takes it, behind DevCrt): indexing the by-value copy with i and slot would put it in scratch
__device__ __forceinline__ const u64 *lin_desc_coef(const LinCombDesc &, u32 i, u32 slot) {
    constexpr size_t off = (sizeof(DevCrt) + alignof(LinCombDesc) - 1) / alignof(LinCombDesc) * alignof(LinCombDesc);
    const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    return (const u64 *)(ka + off + offsetof(LinCombDesc, c)) + (size_t)i * 24 + 3 * slot;
}
template <bool NU, bool FUSED, bool SPLIT>
__global__ void __launch_bounds__(256) k_lin_round(DevCrt t, LinCombDesc desc, const u64 *mz, size_t ld, const u64 *eq, size_t ldeq, size_t n,
                                                   u32 deg, u64 *partial, LinFix fx, u32 xmask) {
    u32 slot = blockIdx.y;
    size_t pairs = n / 2;
    Fq3 acc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) acc[i] = fq3_zero();
    const Fq3 rfix = fq3_make(fx.r.c[0], fx.r.c[1], fx.r.c[2]);
    // the fixed pair (entries 2p, 2p+1 of the new tables) of one F_{p^3} row: from the entries 4p..4p+3 of the previous one, stored when `out` is set
    auto fixed_pair = [&](const u64 *row, size_t ldr, size_t p, u64 *out, size_t ldout, Fq3 &f0, Fq3 &f1) {
        const u64 *fp = row + 4 * p;
        const ulonglong2 a0 = *(const ulonglong2 *)(fp), a1 = *(const ulonglong2 *)(fp + ldr), a2 = *(const ulonglong2 *)(fp + 2 * ldr);
        const ulonglong2 b0 = *(const ulonglong2 *)(fp + 2), b1 = *(const ulonglong2 *)(fp + ldr + 2), b2 = *(const ulonglong2 *)(fp + 2 * ldr + 2);
        const Fq3 lo = fq3_make(a0.x, a1.x, a2.x), hi = fq3_make(b0.x, b1.x, b2.x);
        f0 = fq3_add(lo, M3<NU>(fq3_sub(fq3_make(a0.y, a1.y, a2.y), lo), rfix, t.nu));
        f1 = fq3_add(hi, M3<NU>(fq3_sub(fq3_make(b0.y, b1.y, b2.y), hi), rfix, t.nu));
        if (out) {
            u64 *op = out + 2 * p;
            *(ulonglong2 *)(op) = make_ulonglong2(f0.c[0], f1.c[0]);
            *(ulonglong2 *)(op + ldout) = make_ulonglong2(f0.c[1], f1.c[1]);
            *(ulonglong2 *)(op + 2 * ldout) = make_ulonglong2(f0.c[2], f1.c[2]);
        }
    };
    // the unit coefficients of the tables' multisets, selected once (a dynamically indexed field of the by-value descriptor makes the compiler keep a copy of it in
    // scratch memory: 128 bytes per lane, read twenty times per pair)
    int cu_j[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32 i = desc.ms[j];
        int r = desc.c_unit[0];
#pragma unroll
        for (int q = 1; q < 8; q++) r = i == (u32)q ? desc.c_unit[q] : r;
        cu_j[j] = r;
    }
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (size_t)gridDim.x * 256) {
        Fq3 v[4], st[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((u32)j < desc.t) {
                const u64 *tb = mz + ((size_t)j * 24 + 3 * slot) * ld;
                if (FUSED) {
                    Fq3 f1;
                    fixed_pair(tb, ld, p, fx.mzo + ((size_t)j * 24 + 3 * slot) * fx.ldo, fx.ldo, v[j], f1);
                    st[j] = fq3_sub(f1, v[j]);
                } else {
                    ulonglong2 a0 = *(const ulonglong2 *)(tb + 2 * p), a1 = *(const ulonglong2 *)(tb + ld + 2 * p), a2 = *(const ulonglong2 *)(tb + 2 * ld + 2 * p);
                    v[j] = fq3_make(a0.x, a1.x, a2.x);
                    st[j] = fq3_sub(fq3_make(a0.y, a1.y, a2.y), v[j]);
                }
            } else { v[j] = fq3_zero(); st[j] = fq3_zero(); }
        }
        Fq3 ev, es;
        if (SPLIT) {
            es = fq3_zero();
            if (FUSED) {   // E_i[p] = E_{i-1}[2p] + E_{i-1}[2p+1]  (eq(beta_i, 0) + eq(beta_i, 1) = 1)
                const ulonglong2 e0 = *(const ulonglong2 *)(eq + 2 * p), e1 = *(const ulonglong2 *)(eq + ldeq + 2 * p), e2 = *(const ulonglong2 *)(eq + 2 * ldeq + 2 * p);
                ev = fq3_make(fq_add(e0.x, e0.y), fq_add(e1.x, e1.y), fq_add(e2.x, e2.y));
                if (slot == 0) { fx.eqo[p] = ev.c[0]; fx.eqo[fx.ldeo + p] = ev.c[1]; fx.eqo[2 * fx.ldeo + p] = ev.c[2]; }
            } else ev = fq3_make(eq[p], eq[ldeq + p], eq[2 * ldeq + p]);
        } else
        if (FUSED) {      // eq is one row shared by the 8 slot blocks: every block fixes it, block row 0 stores it
            Fq3 e1v;
            fixed_pair(eq, ldeq, p, slot == 0 ? fx.eqo : nullptr, fx.ldeo, ev, e1v);
            es = fq3_sub(e1v, ev);
        } else {
            ulonglong2 e0 = *(const ulonglong2 *)(eq + 2 * p), e1 = *(const ulonglong2 *)(eq + ldeq + 2 * p), e2 = *(const ulonglong2 *)(eq + 2 * ldeq + 2 * p);
            ev = fq3_make(e0.x, e1.x, e2.x);
            es = fq3_sub(fq3_make(e0.y, e1.y, e2.y), ev);
        }
#pragma unroll
        for (int X = 0; X < 5; X++) {
            if ((u32)X <= deg) {
                if (!SPLIT || ((xmask >> X) & 1)) {   // (wave-uniform; SPLIT: the points not in xmask are only stepped past)
                // comb = (sum_i c_i prod_{j in S_i} v_j) * eq ; table j belongs to multiset ms[j], first[j] marks its start
                Fq3 res = fq3_zero(), term = fq3_zero();
                int sgn = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if ((u32)j < desc.t) {
                        if (desc.first[j]) {  // wave-uniform
                            if (sgn) res = sgn < 0 ? fq3_sub(res, term) : fq3_add(res, term);
                            if (cu_j[j]) { term = v[j]; sgn = cu_j[j]; }
                            else {
                                const u64 *cp = lin_desc_coef(desc, desc.ms[j], slot);
                                term = M3<NU>(fq3_make(cp[0], cp[1], cp[2]), v[j], t.nu); sgn = 1;
                            }
                        } else term = M3<NU>(term, v[j], t.nu);
                    }
                }
                if (sgn) res = sgn < 0 ? fq3_sub(res, term) : fq3_add(res, term);
                // (the loop over X stays rolled -- its body is seven products --, so acc[X] would be a dynamically indexed array: 120 bytes of scratch per lane)
                const Fq3 gx = M3<NU>(res, ev, t.nu);
                if (X == 0) acc[0] = fq3_add(acc[0], gx);
                else if (X == 1) acc[1] = fq3_add(acc[1], gx);
                else if (X == 2) acc[2] = fq3_add(acc[2], gx);
                else if (X == 3) acc[3] = fq3_add(acc[3], gx);
                else acc[4] = fq3_add(acc[4], gx);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = fq3_add(v[j], st[j]);
                ev = fq3_add(ev, es);
            }
        }
    }
    u64 vv[15];
#pragma unroll
    for (int i = 0; i < 5; i++) { vv[3 * i] = acc[i].c[0]; vv[3 * i + 1] = acc[i].c[1]; vv[3 * i + 2] = acc[i].c[2]; }
    // partial[block][X][3*slot+c]
    __shared__ u64 red[15];
    block_sum_store<15>(vv, red);
    __syncthreads();
    if (threadIdx.x < 15) partial[(size_t)blockIdx.x * 120 + (threadIdx.x / 3) * 24 + 3 * slot + threadIdx.x % 3] = red[threadIdx.x];
}
size_t round_partial_words() { return (size_t)RED_BLOCKS * 120; }
// leaving the split form: the ordinary eq table of a round's n entries from the per-pair table E,  out[2p + b] = w_b * E[p]  (w_b = c eq(beta_i, b))
template <bool NU>
__global__ void __launch_bounds__(256) k_eq_expand(DevCrt t, const u64 *E, size_t lde, size_t pairs, Fq3Const w0, Fq3Const w1, u64 *out, size_t ldo) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= pairs) return;
    const Fq3 e = fq3_make(E[p], E[lde + p], E[2 * lde + p]);
    const Fq3 a = M3<NU>(e, fq3_make(w0.c[0], w0.c[1], w0.c[2]), t.nu), b = M3<NU>(e, fq3_make(w1.c[0], w1.c[1], w1.c[2]), t.nu);
    *(ulonglong2 *)(out + 2 * p) = make_ulonglong2(a.c[0], b.c[0]);
    *(ulonglong2 *)(out + ldo + 2 * p) = make_ulonglong2(a.c[1], b.c[1]);
    *(ulonglong2 *)(out + 2 * ldo + 2 * p) = make_ulonglong2(a.c[2], b.c[2]);
}
// E_{i+1}[p] = E_i[2p] + E_i[2p+1]  (per-pair eq tables of the split form: eq(beta, 0) + eq(beta, 1) = 1); [3][ld] planes
__global__ void __launch_bounds__(256) k_eq_pairsum(const u64 *in, size_t ld_in, size_t n_out, u64 *out, size_t ld_out) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_out) return;
    const u32 q = blockIdx.y;
    const ulonglong2 a = *(const ulonglong2 *)(in + (size_t)q * ld_in + 2 * p);
    out[(size_t)q * ld_out + p] = fq_add(a.x, a.y);
}
void launch_eq_pairsum(const u64 *in, size_t ld_in, size_t n_out, u64 *out, size_t ld_out, hipStream_t s) {
    if (n_out) hipLaunchKernelGGL(k_eq_pairsum, dim3(cdiv(n_out, 256), 3), dim3(256), 0, s, in, ld_in, n_out, out, ld_out);
}
void launch_eq_expand(const DevCrt &t, const u64 *E, size_t lde, size_t pairs, Fq3Const w0, Fq3Const w1, u64 *out, size_t ldo, hipStream_t s) {
    if (!pairs) return;
    LF_LAUNCH(k_eq_expand, t.nu2p40, dim3(cdiv(pairs, 256)), dim3(256), s, t, E, lde, pairs, w0, w1, out, ldo);
}
void launch_lin_round(const DevCrt &t, const LinCombDesc &desc, const u64 *mz, size_t ld, const u64 *eq, size_t ldeq, size_t n, u32 deg,
                      u64 *partial, u64 *out, hipStream_t s, u32 max_blocks, u32 xmask) {
    u32 gb = (u32)((n / 2 + 255) / 256);
    const u32 cap = max_blocks && max_blocks < RED_BLOCKS ? max_blocks : RED_BLOCKS;
    if (gb > cap) gb = cap;
    if (gb < 1) gb = 1;
    LinFix fx = {};
    if (xmask) {
        if (t.nu2p40) hipLaunchKernelGGL((k_lin_round<true, false, true>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz, ld, eq, ldeq, n, deg, partial, fx, xmask);
        else hipLaunchKernelGGL((k_lin_round<false, false, true>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz, ld, eq, ldeq, n, deg, partial, fx, xmask);
    } else
    if (t.nu2p40) hipLaunchKernelGGL((k_lin_round<true, false, false>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz, ld, eq, ldeq, n, deg, partial, fx, 0u);
    else hipLaunchKernelGGL((k_lin_round<false, false, false>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz, ld, eq, ldeq, n, deg, partial, fx, 0u);
    hipLaunchKernelGGL(k_reduce_rows, dim3((deg + 1) * 24), dim3(256), 0, s, partial, gb, 120, out);
}
// round message with fix_variables fused: mz_prev / eq_prev hold 2n entries per row (strides ld_prev / ldeq_prev); the tables fixed with r are written to
// mz_out / eq_out (n entries per row, strides ld_out / ldeq_out) and the message is that of the fixed tables
void launch_lin_round_fused(const DevCrt &t, const LinCombDesc &desc, const u64 *mz_prev, size_t ld_prev, const u64 *eq_prev, size_t ldeq_prev, Fq3Const r, u64 *mz_out,
                            size_t ld_out, u64 *eq_out, size_t ldeq_out, size_t n, u32 deg, u64 *partial, u64 *out, hipStream_t s, u32 max_blocks, u32 xmask) {
    u32 gb = (u32)((n / 2 + 255) / 256);
    const u32 cap = max_blocks && max_blocks < RED_BLOCKS ? max_blocks : RED_BLOCKS;
    if (gb > cap) gb = cap;
    if (gb < 1) gb = 1;
    LinFix fx = {r, mz_out, ld_out, eq_out, ldeq_out};
    if (xmask) {
        if (t.nu2p40) hipLaunchKernelGGL((k_lin_round<true, true, true>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz_prev, ld_prev, eq_prev, ldeq_prev, n, deg, partial, fx, xmask);
        else hipLaunchKernelGGL((k_lin_round<false, true, true>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz_prev, ld_prev, eq_prev, ldeq_prev, n, deg, partial, fx, xmask);
    } else
    if (t.nu2p40) hipLaunchKernelGGL((k_lin_round<true, true, false>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz_prev, ld_prev, eq_prev, ldeq_prev, n, deg, partial, fx, 0u);
    else hipLaunchKernelGGL((k_lin_round<false, true, false>), dim3(gb, 8), dim3(256), 0, s, t, desc, mz_prev, ld_prev, eq_prev, ldeq_prev, n, deg, partial, fx, 0u);
    hipLaunchKernelGGL(k_reduce_rows, dim3((deg + 1) * 24), dim3(256), 0, s, partial, gb, 120, out);
}

// ---------------------------------------------------------------------------------------------------------
// folding sumcheck (comb = nifs/folding/utils.rs:273-325, b = 2):
//   g(X) = eqL*G1 + eqR*G2 + eqB * sum_{k,d} mu_k^{d+1} * fhat_kd (fhat_kd^2 - 1)
// Both kernels evaluate the pair-polynomials in coefficient form (exact in F_p, identical sums).
// G part in coefficient form: gco += coefficients of (e0 + X de)(g0 + X dg) for the two halves.  The callers evaluate the accumulated
// quadratic at X = 0..4 ONCE per thread (add_poly_evals) instead of once per pair (was: 36 small-constant field products per pair and slot).
template <bool NU>
__device__ __forceinline__ void fold_g13(Fq3 (&gco)[3], const FoldRoundArgs &a, u32 slot, size_t p, u64 nu) {
    for (int h = 0; h < 2; h++) {
        const u64 *eq = h ? a.eqR : a.eqL;
        const u64 *G = h ? a.G2 : a.G1;
        ulonglong2 e0 = *(const ulonglong2 *)(eq + 2 * p), e1 = *(const ulonglong2 *)(eq + a.ld + 2 * p), e2 = *(const ulonglong2 *)(eq + 2 * a.ld + 2 * p);
        const u64 *gp = G + (size_t)(3 * slot) * a.ld;
        ulonglong2 g0 = *(const ulonglong2 *)(gp + 2 * p), g1 = *(const ulonglong2 *)(gp + a.ld + 2 * p), g2 = *(const ulonglong2 *)(gp + 2 * a.ld + 2 * p);
        Fq3 ea = fq3_make(e0.x, e1.x, e2.x), eb = fq3_make(e0.y, e1.y, e2.y);
        Fq3 ga = fq3_make(g0.x, g1.x, g2.x), gb = fq3_make(g0.y, g1.y, g2.y);
        Fq3 co[3];
        co[0] = M3<NU>(ea, ga, nu);
        co[2] = M3<NU>(fq3_sub(eb, ea), fq3_sub(gb, ga), nu);
        co[1] = fq3_sub(fq3_sub(M3<NU>(eb, gb, nu), co[0]), co[2]);
#pragma unroll
        for (int e = 0; e < 3; e++) gco[e] = fq3_add(gco[e], co[e]);
    }
}
// the same with the evaluations at X = 0..4 added per pair (k_fold_round: no registers to spare for the coefficient accumulators)
template <bool NU>
__device__ __forceinline__ void fold_g13_evals(Fq3 (&acc)[5], const FoldRoundArgs &a, u32 slot, size_t p, u64 nu) {
    for (int h = 0; h < 2; h++) {
        const u64 *eq = h ? a.eqR : a.eqL;
        const u64 *G = h ? a.G2 : a.G1;
        ulonglong2 e0 = *(const ulonglong2 *)(eq + 2 * p), e1 = *(const ulonglong2 *)(eq + a.ld + 2 * p), e2 = *(const ulonglong2 *)(eq + 2 * a.ld + 2 * p);
        const u64 *gp = G + (size_t)(3 * slot) * a.ld;
        ulonglong2 g0 = *(const ulonglong2 *)(gp + 2 * p), g1 = *(const ulonglong2 *)(gp + a.ld + 2 * p), g2 = *(const ulonglong2 *)(gp + 2 * a.ld + 2 * p);
        Fq3 ea = fq3_make(e0.x, e1.x, e2.x), eb = fq3_make(e0.y, e1.y, e2.y);
        Fq3 ga = fq3_make(g0.x, g1.x, g2.x), gb = fq3_make(g0.y, g1.y, g2.y);
        Fq3 co[3];
        co[0] = M3<NU>(ea, ga, nu);
        co[2] = M3<NU>(fq3_sub(eb, ea), fq3_sub(gb, ga), nu);
        co[1] = fq3_sub(fq3_sub(M3<NU>(eb, gb, nu), co[0]), co[2]);
        add_poly_evals<5>(acc, co, 3);
    }
}
template <bool NU>
__device__ __forceinline__ void fold_g2_finish(Fq3 (&acc)[5], const Fq3 Q[4], const FoldRoundArgs &a, size_t p, u64 nu) {
    ulonglong2 b0 = *(const ulonglong2 *)(a.eqB + 2 * p), b1 = *(const ulonglong2 *)(a.eqB + a.ld + 2 * p), b2 = *(const ulonglong2 *)(a.eqB + 2 * a.ld + 2 * p);
    Fq3 ea = fq3_make(b0.x, b1.x, b2.x), es = fq3_sub(fq3_make(b0.y, b1.y, b2.y), ea);
#pragma unroll
    for (int X = 0; X < 5; X++) {
        Fq3 v = Q[3];
        for (int e = 2; e >= 0; e--) v = fq3_add(fq3_mul_small(v, X), Q[e]);
        acc[X] = fq3_add(acc[X], M3<NU>(v, ea, nu));
        ea = fq3_add(ea, es);
    }
}
__device__ __forceinline__ void store_round_partial(Fq3 (&acc)[5], u32 slot, u64 *partial) {
    u64 vv[15];
#pragma unroll
    for (int i = 0; i < 5; i++) { vv[3 * i] = acc[i].c[0]; vv[3 * i + 1] = acc[i].c[1]; vv[3 * i + 2] = acc[i].c[2]; }
    __shared__ u64 red[15];
    block_sum_store<15>(vv, red);
    __syncthreads();
    if (threadIdx.x < 15) partial[(size_t)blockIdx.x * 120 + (threadIdx.x / 3) * 24 + 3 * slot + threadIdx.x % 3] = red[threadIdx.x];
}

// round 1: f-hat entries are base-field digits in {-1,0,1}; P(f(X)) = f^3 - f is a small integer, so the
// mu-weighted sum is accumulated as exact 64-bit integer dot products (no modular multiply in the inner loop).
template <bool NU>
__global__ void __launch_bounds__(256) k_fold_round1(DevCrt t, FoldRoundArgs a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                                                     u32 K, const Fq3Const *mu_pow, u64 *partial) {
    u32 slot = blockIdx.y;
    const size_t pend = a.p0 + a.pcnt;
    Fq3 acc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) acc[i] = fq3_zero();
    Fq3 gco[3] = {fq3_zero(), fq3_zero(), fq3_zero()};   // G part, coefficient form
    for (size_t p = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; p < pend; p += (size_t)gridDim.x * 256) {
        fold_g13<NU>(gco, a, slot, p, t.nu);
        // cubic coefficients of sum_kd mu_kd * P(f0 + X*df): integer parts split in lo/hi 32-bit halves of mu
        int64_t lo[4][3], hi[4][3];
        int32_t cs[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int c = 0; c < 3; c++) { lo[e][c] = 0; hi[e][c] = 0; }
        if (2 * p < n_planes) {
            for (int side = 0; side < 2; side++) {
                const int32_t *pl = side ? planesR : planesL;
                for (int d = 0; d < 3; d++) {
                    const int32_t *src = pl + (size_t)(8 * d + slot) * n_planes + 2 * p;
                    int32_t v0 = src[0], v1 = (2 * p + 1 < n_planes) ? src[1] : 0;
                    for (u32 k = 0; k < K; k++) {
                        int f0 = digit2(v0, k), df = digit2(v1, k) - f0;
                        // P(f0 + X df) = (f0^3 - f0) + (3 f0^2 - 1) df X + 3 f0 df^2 X^2 + df^3 X^3
                        int c0 = f0 * f0 * f0 - f0, c1 = (3 * f0 * f0 - 1) * df, c2 = 3 * f0 * df * df, c3 = df * df * df;
                        Fq3Const m = mu_pow[(side * K + k) * 3 + d];
                        cs[0] += c0; cs[1] += c1; cs[2] += c2; cs[3] += c3;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            // signed 32x32->64 multiply-adds (v_mad_i64_i32): word - 2^31, corrected with 2^31 * sum(coef)
                            int32_t ml = (int32_t)((u32)m.c[c] ^ 0x80000000u), mh = (int32_t)((u32)(m.c[c] >> 32) ^ 0x80000000u);
                            lo[0][c] += (int64_t)ml * c0; hi[0][c] += (int64_t)mh * c0;
                            lo[1][c] += (int64_t)ml * c1; hi[1][c] += (int64_t)mh * c1;
                            lo[2][c] += (int64_t)ml * c2; hi[2][c] += (int64_t)mh * c2;
                            lo[3][c] += (int64_t)ml * c3; hi[3][c] += (int64_t)mh * c3;
                        }
                    }
                }
            }
        }
        Fq3 Q[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                int64_t off = (int64_t)cs[e] << 31;
                Q[e].c[c] = fq_add(fq_from_i64(lo[e][c] + off), fq_mul(fq_from_i64(hi[e][c] + off), 1ULL << 32));
            }
        fold_g2_finish<NU>(acc, Q, a, p, t.nu);
    }
    add_poly_evals<5>(acc, gco, 3);
    store_round_partial(acc, slot, partial);
}
void launch_fold_round1(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                        const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((a.pcnt + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    LF_LAUNCH(k_fold_round1, t.nu2p40, dim3(gb, 8), dim3(256), s, t, a, planesL, planesR, n_planes, K, mu_pow_dev, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3(5 * 24), dim3(256), 0, s, partial, gb, 120, out);
}

// G part of a round message only (eqL*G1 + eqR*G2): the norm part comes from the int8 GEMM of lf_sv_rounds.hip
template <bool NU>
__global__ void __launch_bounds__(256) k_fold_round_g(DevCrt t, FoldRoundArgs a, u64 *partial) {
    u32 slot = blockIdx.y;
    const size_t pend = a.p0 + a.pcnt;
    Fq3 acc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) acc[i] = fq3_zero();
    Fq3 gco[3] = {fq3_zero(), fq3_zero(), fq3_zero()};   // G part, coefficient form
    for (size_t p = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; p < pend; p += (size_t)gridDim.x * 256) fold_g13<NU>(gco, a, slot, p, t.nu);
    add_poly_evals<5>(acc, gco, 3);
    store_round_partial(acc, slot, partial);
}
void launch_fold_round_g(const DevCrt &t, const FoldRoundArgs &a, u64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((a.pcnt + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    LF_LAUNCH(k_fold_round_g, t.nu2p40, dim3(gb, 8), dim3(256), s, t, a, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3(5 * 24), dim3(256), 0, s, partial, gb, 120, out);
}

// Rounds 1 and 2 as table look-ups.  Before round 3 a table entry is a function of one (round 1) or two (round 2) ternary digits, so
// the cubic mu_kd * (h^3 - h), h = f0 + X (f1 - f0), of a pair depends only on mu_kd and the 2 / 4 digits behind the pair: 9 / 81
// possible coefficient quadruples per table.  k_fold_polytab multiplies the (host-built) quadruples Poly[code][4] with every mu_kd once
// per round; the round kernel then only gathers TP[kd][code] (96 bytes) and adds -- no multiplication in the table loop.
template <bool NU>
__global__ void __launch_bounds__(256) k_fold_polytab(DevCrt t, const u64 *poly /*[ncode][12]*/, const Fq3Const *mu_pow, u32 ncode, u64 *tp) {
    u32 kd = blockIdx.x;
    Fq3Const mc = mu_pow[kd];
    Fq3 mu = fq3_make(mc.c[0], mc.c[1], mc.c[2]);
    for (u32 i = threadIdx.x; i < ncode * 4; i += 256) {
        Fq3 v = M3<NU>(mu, fq3_make(poly[3 * i], poly[3 * i + 1], poly[3 * i + 2]), t.nu);
        u64 *o = tp + ((size_t)kd * ncode * 4 + i) * 3;
        o[0] = v.c[0]; o[1] = v.c[1]; o[2] = v.c[2];
    }
}
__device__ __forceinline__ u32 digit_code4(const int32_t *v, u32 k);
__device__ __forceinline__ u32 digit_code2(const int32_t *v, u32 k) {   // 4 + sign_0 bit_0 + 3 sign_1 bit_1
    int code = 4;
#pragma unroll
    for (int b = 0; b < 2; b++) {
        int32_t x = v[b], mg = x < 0 ? -x : x;
        int bit = (mg >> k) & 1, w = b ? 3 : 1;
        code += x < 0 ? -bit * w : bit * w;
    }
    return (u32)code;
}
template <bool NU, int R>
__global__ void __launch_bounds__(256) k_fold_round_tab(DevCrt t, FoldRoundArgs a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                                                        u32 K, const u64 *tp, u64 *partial) {
    constexpr int E = R == 1 ? 2 : 4;          // plane entries behind one pair
    constexpr u32 NC = R == 1 ? 9 : 81;
    u32 slot = blockIdx.y;
    const size_t pend = a.p0 + a.pcnt;
    Fq3 acc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) acc[i] = fq3_zero();
    Fq3 gco[3] = {fq3_zero(), fq3_zero(), fq3_zero()};   // G part, coefficient form
    for (size_t p = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; p < pend; p += (size_t)gridDim.x * 256) {
        fold_g13<NU>(gco, a, slot, p, t.nu);
        u64 s64[12];      // lazy 64-bit sums with carry counters
        u32 scy[12];
#pragma unroll
        for (int i = 0; i < 12; i++) { s64[i] = 0; scy[i] = 0; }
        if ((size_t)E * p < n_planes) {
            for (int side = 0; side < 2; side++) {
                const int32_t *pl = side ? planesR : planesL;
                for (int d = 0; d < 3; d++) {
                    const int32_t *src = pl + (size_t)(8 * d + slot) * n_planes + (size_t)E * p;
                    int32_t v[E];
#pragma unroll
                    for (int q = 0; q < E; q++) v[q] = (size_t)E * p + q < n_planes ? src[q] : 0;
                    for (u32 k = 0; k < K; k++) {
                        const u32 code = R == 1 ? digit_code2(v, k) : digit_code4(v, k);
                        const ulonglong2 *e = (const ulonglong2 *)(tp + ((size_t)((side * K + k) * 3 + d) * NC + code) * 12);
#pragma unroll
                        for (int q = 0; q < 6; q++) {
                            ulonglong2 w = e[q];
                            u64 sm = s64[2 * q] + w.x; scy[2 * q] += sm < w.x; s64[2 * q] = sm;
                            sm = s64[2 * q + 1] + w.y; scy[2 * q + 1] += sm < w.y; s64[2 * q + 1] = sm;
                        }
                    }
                }
            }
        }
        Fq3 Q[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int c = 0; c < 3; c++) Q[e].c[c] = fq_canon(fq_reduce128_loose(s64[3 * e + c], (u64)scy[3 * e + c]));
        fold_g2_finish<NU>(acc, Q, a, p, t.nu);
    }
    add_poly_evals<5>(acc, gco, 3);
    store_round_partial(acc, slot, partial);
}
// poly_dev: [ncode][4][3] coefficient quadruples (c0..c3 of h^3 - h) of the 9 (round 1) / 81 (round 2) digit codes; tp_dev: 2K*3 * ncode * 12 words
void launch_fold_round_tab(const DevCrt &t, int round, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                           const Fq3Const *mu_pow_dev, const u64 *poly_dev, u64 *tp_dev, u64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((a.pcnt + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    const u32 ncode = round == 1 ? 9 : 81;
    LF_LAUNCH(k_fold_polytab, t.nu2p40, dim3(2 * K * 3), dim3(256), s, t, poly_dev, mu_pow_dev, ncode, tp_dev);
    if (round == 1) {
        if (t.nu2p40) hipLaunchKernelGGL((k_fold_round_tab<true, 1>), dim3(gb, 8), dim3(256), 0, s, t, a, planesL, planesR, n_planes, K, tp_dev, partial);
        else hipLaunchKernelGGL((k_fold_round_tab<false, 1>), dim3(gb, 8), dim3(256), 0, s, t, a, planesL, planesR, n_planes, K, tp_dev, partial);
    } else {
        if (t.nu2p40) hipLaunchKernelGGL((k_fold_round_tab<true, 2>), dim3(gb, 8), dim3(256), 0, s, t, a, planesL, planesR, n_planes, K, tp_dev, partial);
        else hipLaunchKernelGGL((k_fold_round_tab<false, 2>), dim3(gb, 8), dim3(256), 0, s, t, a, planesL, planesR, n_planes, K, tp_dev, partial);
    }
    hipLaunchKernelGGL(k_reduce_rows, dim3(5 * 24), dim3(256), 0, s, partial, gb, 120, out);
}

// round 2: after one fix the virtual f-hat entries are  d_a + (d_b - d_a) * r1  with digits d in {-1,0,1}.  For a pair
// f(X) = u(X) + v(X) r1 with small-integer linear u, v, so  f^3 - f = (u^3-u) + (3u^2 v - v) r1 + 3 u v^2 r1^2 + v^3 r1^3
// and the mu-weighted sums of the 16 integer coefficients are exact 64-bit integer dot products again.
template <bool NU>
__global__ void __launch_bounds__(256, 2) k_fold_round2(DevCrt t, FoldRoundArgs a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                                                     u32 K, const Fq3Const *mu_pow, Fq3Const r1c, u64 *partial) {
    u32 slot = blockIdx.y;
    const size_t pend = a.p0 + a.pcnt;
    const u64 nu = t.nu;
    const Fq3 r1 = fq3_make(r1c.c[0], r1c.c[1], r1c.c[2]);
    const Fq3 r1s = S3<NU>(r1, nu), r1c3 = M3<NU>(r1s, r1, nu);
    Fq3 acc[5];
#pragma unroll
    for (int i = 0; i < 5; i++) acc[i] = fq3_zero();
    Fq3 gco[3] = {fq3_zero(), fq3_zero(), fq3_zero()};   // G part, coefficient form
    for (size_t p = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; p < pend; p += (size_t)gridDim.x * 256) {
        fold_g13<NU>(gco, a, slot, p, nu);
        Fq3 Q[4] = {fq3_zero(), fq3_zero(), fq3_zero(), fq3_zero()};
        if (4 * p < n_planes) {
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {  // pass 0: powers r1^0, r1^1 ; pass 1: r1^2, r1^3
                int64_t lo[2][4][3], hi[2][4][3];
                int32_t cs[2][4];
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        cs[e][j] = 0;
#pragma unroll
                        for (int c = 0; c < 3; c++) { lo[e][j][c] = 0; hi[e][j][c] = 0; }
                    }
                for (int side = 0; side < 2; side++) {
                    const int32_t *pl = side ? planesR : planesL;
                    for (int d = 0; d < 3; d++) {
                        const int32_t *src = pl + (size_t)(8 * d + slot) * n_planes + 4 * p;
                        int32_t w0 = src[0];
                        int32_t w1 = 4 * p + 1 < n_planes ? src[1] : 0, w2 = 4 * p + 2 < n_planes ? src[2] : 0, w3 = 4 * p + 3 < n_planes ? src[3] : 0;
                        for (u32 k = 0; k < K; k++) {
                            int d0 = digit2(w0, k), d1 = digit2(w1, k), d2 = digit2(w2, k), d3 = digit2(w3, k);
                            int u0 = d0, u1 = d2 - d0, v0 = d1 - d0, v1 = (d3 - d2) - v0;
                            int cf[2][4];
                            if (pass == 0) {
                                int u0s = u0 * u0;
                                cf[0][0] = u0s * u0 - u0; cf[0][1] = 3 * u0s * u1 - u1; cf[0][2] = 3 * u0 * u1 * u1; cf[0][3] = u1 * u1 * u1;
                                cf[1][0] = 3 * u0s * v0 - v0; cf[1][1] = 3 * (u0s * v1 + 2 * u0 * u1 * v0) - v1;
                                cf[1][2] = 3 * (2 * u0 * u1 * v1 + u1 * u1 * v0); cf[1][3] = 3 * u1 * u1 * v1;
                            } else {
                                int v0s = v0 * v0, v1s = v1 * v1;
                                cf[0][0] = 3 * u0 * v0s; cf[0][1] = 3 * (2 * u0 * v0 * v1 + u1 * v0s); cf[0][2] = 3 * (u0 * v1s + 2 * u1 * v0 * v1);
                                cf[0][3] = 3 * u1 * v1s;
                                cf[1][0] = v0s * v0; cf[1][1] = 3 * v0s * v1; cf[1][2] = 3 * v0 * v1s; cf[1][3] = v1s * v1;
                            }
                            Fq3Const m = mu_pow[(side * K + k) * 3 + d];
#pragma unroll
                            for (int e = 0; e < 2; e++)
#pragma unroll
                                for (int j = 0; j < 4; j++) cs[e][j] += cf[e][j];
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                int32_t ml = (int32_t)((u32)m.c[c] ^ 0x80000000u), mh = (int32_t)((u32)(m.c[c] >> 32) ^ 0x80000000u);
#pragma unroll
                                for (int e = 0; e < 2; e++)
#pragma unroll
                                    for (int j = 0; j < 4; j++) { lo[e][j][c] += (int64_t)ml * cf[e][j]; hi[e][j][c] += (int64_t)mh * cf[e][j]; }
                            }
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; e++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        Fq3 T;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            int64_t off = (int64_t)cs[e][j] << 31;
                            T.c[c] = fq_add(fq_from_i64(lo[e][j][c] + off), fq_mul(fq_from_i64(hi[e][j][c] + off), 1ULL << 32));
                        }
                        if (pass == 0 && e == 0) Q[j] = fq3_add(Q[j], T);
                        else Q[j] = fq3_add(Q[j], M3<NU>(T, pass == 0 ? r1 : (e == 0 ? r1s : r1c3), nu));
                    }
            }
        }
        fold_g2_finish<NU>(acc, Q, a, p, nu);
    }
    add_poly_evals<5>(acc, gco, 3);
    store_round_partial(acc, slot, partial);
}
void launch_fold_round2(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                        const Fq3Const *mu_pow_dev, Fq3Const r1, u64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((a.pcnt + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    LF_LAUNCH(k_fold_round2, t.nu2p40, dim3(gb, 8), dim3(256), s, t, a, planesL, planesR, n_planes, K, mu_pow_dev, r1, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3(5 * 24), dim3(256), 0, s, partial, gb, 120, out);
}

// after r_2: F[(side*K+k)*3+d][3*slot+c][j] = sum_{b<4} W_b * digit(f[4j+b]),  W = eq((r1,r2), .),  j < m/4
__global__ void __launch_bounds__(256) k_fold_materialize2(const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t j0, size_t q, u32 K,
                                                           Fq3Const W0, Fq3Const W1, Fq3Const W2, Fq3Const W3, u64 *F) {
    size_t jl = (size_t)blockIdx.x * 256 + threadIdx.x;   // local entry; global entry j = j0 + jl, F holds q local entries
    u32 cidx = blockIdx.y;
    if (jl >= q) return;
    size_t j = j0 + jl;
    u32 d = cidx / 8, slot = cidx % 8;
    const Fq3Const W[4] = {W0, W1, W2, W3};
    for (int side = 0; side < 2; side++) {
        const int32_t *pl = (side ? planesR : planesL) + (size_t)cidx * n_planes;
        int32_t v[4];
#pragma unroll
        for (int b = 0; b < 4; b++) v[b] = 4 * j + b < n_planes ? pl[4 * j + b] : 0;
        for (u32 k = 0; k < K; k++) {
            u64 acc[3] = {0, 0, 0};
#pragma unroll
            for (int b = 0; b < 4; b++) {
                int dg = digit2(v[b], k);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    u64 w = W[b].c[c];
                    acc[c] = dg > 0 ? fq_add(acc[c], w) : (dg < 0 ? fq_sub(acc[c], w) : acc[c]);
                }
            }
            u64 *dst = F + (((size_t)(side * K + k) * 3 + d) * 24 + 3 * slot) * q + jl;
            dst[0] = acc[0]; dst[q] = acc[1]; dst[2 * q] = acc[2];
        }
    }
}
void launch_fold_materialize2(const DevCrt &t, const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t j0, size_t q, u32 K,
                              const Fq3Const W[4], u64 *F, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_materialize2, dim3(cdiv(q, 256), 24), dim3(256), 0, s, planesL, planesR, n_planes, j0, q, K, W[0], W[1], W[2], W[3], F);
}

// F[(side*K+k)*3+d][3*slot+c][j] = f0 + r1*(f1-f0), j < m/2  (first fix of the virtual f-hat tables)
__global__ void __launch_bounds__(256) k_fold_materialize(const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t m, u32 K,
                                                          Fq3Const r1, u64 *F) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 cidx = blockIdx.y;  // coefficient 0..23 -> d = cidx/8, slot = cidx%8
    size_t half = m / 2;
    if (j >= half) return;
    u32 d = cidx / 8, slot = cidx % 8;
    // multiples -2..2 of r1
    u64 mul[5][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        u64 r = r1.c[c], r2 = fq_add(r, r);
        mul[0][c] = fq_neg(r2); mul[1][c] = fq_neg(r); mul[2][c] = 0; mul[3][c] = r; mul[4][c] = r2;
    }
    for (int side = 0; side < 2; side++) {
        const int32_t *pl = (side ? planesR : planesL) + (size_t)cidx * n_planes;
        int32_t v0 = 2 * j < n_planes ? pl[2 * j] : 0, v1 = 2 * j + 1 < n_planes ? pl[2 * j + 1] : 0;
        for (u32 k = 0; k < K; k++) {
            int f0 = digit2(v0, k), df = digit2(v1, k) - f0;
            u64 *dst = F + (((size_t)(side * K + k) * 3 + d) * 24 + 3 * slot) * half + j;
            dst[0] = fq_add(fq_from_digit(f0), mul[df + 2][0]);
            dst[half] = mul[df + 2][1];
            dst[2 * half] = mul[df + 2][2];
        }
    }
}
void launch_fold_materialize(const DevCrt &t, const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t m, u32 K, Fq3Const r1,
                             u64 *F, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_materialize, dim3(cdiv(m / 2, 256), 24), dim3(256), 0, s, planesL, planesR, n_planes, m, K, r1, F);
}

// general round on materialised f-hat tables.  grid = (pair blocks, 8 slots, kd chunks): when few pairs remain the
// 2K*3 tables are split over blockIdx.z -- the round message is linear in the per-chunk partial sums Q, so each chunk
// contributes eqB(X)*Q_chunk(X) and chunk 0 adds the g1/g3 products.
//
// MODE selects where a pair (f0, f1) of table kd comes from (the multiply phase is ALU-bound, so the producer's memory
// traffic hides under it and the separate memory-bound pass disappears):
//   0  F holds the current tables:                         f0 = F[2p], f1 = F[2p+1]
//   1  fused fix_variables: Fsrc.F holds the PREVIOUS round's tables,  f0 = F[4p] + r (F[4p+1] - F[4p]),  f1 likewise from
//      4p+2, 4p+3; the fixed pair is also stored to Fsrc.out (ld = Fsrc.ldo) for the next round
//   3  round 3 without materialised tables: after two rounds an entry of table (side,k,d) is sum_b W_b * digit_k(plane[4j+b]) with
//      four ternary digits, i.e. one of 81 values that do not depend on the table or the slot -- a look-up table in LDS indexed by
//      the digit code replaces both the 4.8 GB k_fold_materialize2 pass and the table reads of this round
//   4  round 4 on top of mode 3: the four round-3 entries 4p..4p+3 come from the same look-up table, are fixed with r and the pair
//      is stored to Fsrc.out like in mode 1 (the first materialised tables are the m/8-entry ones)
//   6  mode 4 without a single reduced product: a fixed entry is f = (1-r) L[c_lo] + r L[c_hi], one of 81^2 values, so its square comes from a 6561-entry
//      table and mu_kd f = (mu_kd (1-r) L)[c_lo] + (mu_kd r L)[c_hi] from two 81-entry tables per table kd (k_fold_r4tab, rebuilt per step: they depend
//      on r_3 and mu); the cubic's four sums are then P0..P3 = sum mu f0^3, mu f1 f0^2, mu f0 f1^2, mu f1^3 as in mode 3 -- four lazy products per table
//   7  round 5 still from the planes: an entry of the m/16-entry tables is A'[c0] + B'[c1] + C'[c2] + D'[c3] = X + Y with four 81-entry tables
//      ((1-r4)(1-r3) L, (1-r4) r3 L, r4 (1-r3) L, r4 r3 L), so its square is X^2 + Y^2 (two 6561-entry tables) + 2 X Y (the one reduced product left per
//      entry) and mu_kd f four gathers; the fixed pair is stored for round 6.  Mode 6 then stores nothing (src.out = null): the 2.4 GB of m/8-entry tables
//      are never written or read
// (An earlier variant of mode 3 that rebuilt the entries with conditional modular additions measured slower than the separate pass.)
struct FoldSrc {
    u64 *out; size_t ldo;                 // modes 1, 4: where the fixed pair is written (entries 2p, 2p+1)
    Fq3Const r;
    const int32_t *planesL, *planesR;     // modes 3, 4
    size_t n_planes;
    const u64 *lut;                       // [2][81][3]: sum_b (t_b - 1) W_b for code = sum_b t_b 3^b, then the squares of those
    const u64 *mutab;                     // mode 5: [3][2K*3][81][4] = mu_kd * value, mu_kd * value^2, mu_kd * value^3 (k_fold_mutab)
    Fq3Const r_prev;                      // mode 7: the challenge fixed one round earlier (r_3; r = r_4)
    const u64 *xx5, *yy5, *mt5;           // mode 7: [81*81][4] squares of the low / high halves of a fixed entry, [2K*3][4][81][4] = mu_kd {A', B', C', D'} (k_fold_r5tab)
    const u64 *sq4, *mt4;                 // mode 6: [81*81][4] squares of the fixed look-up values, [2K*3][2][81][4] = mu_kd (1 - r) L, mu_kd r L (k_fold_r4tab)
    const u64 *E; size_t ldE;             // SPLIT kernels (modes 1, 6, 7): the per-pair eq table E_i [3][ldE] of the split form (run of the folding sumcheck in lf_capi.cpp)
};
// per-table products of the 81 look-up values with mu_kd (round 3, mode 5): with them a table costs two lazy products instead of six
template <bool NU>
__global__ void __launch_bounds__(128) k_fold_mutab(DevCrt t, const u64 *lut, const Fq3Const *mu_pow, u32 nkd, u64 *mutab) {
    u32 kd = blockIdx.x, code = threadIdx.x;
    if (code >= 81) return;
    Fq3 L = fq3_make(lut[3 * code], lut[3 * code + 1], lut[3 * code + 2]);
    Fq3Const mc = mu_pow[kd];
    Fq3 m1 = M3<NU>(fq3_make(mc.c[0], mc.c[1], mc.c[2]), L, t.nu), m2 = M3<NU>(m1, L, t.nu), m3 = M3<NU>(m2, L, t.nu);
    const Fq3 v[3] = {m1, m2, m3};
#pragma unroll
    for (int q = 0; q < 3; q++) {
        u64 *o = mutab + (((size_t)q * nkd + kd) * 81 + code) * 4;
        o[0] = v[q].c[0]; o[1] = v[q].c[1]; o[2] = v[q].c[2]; o[3] = 0;
    }
}
// tables of mode 6 (see above): sq[c_lo * 81 + c_hi] = ((1-r) L[c_lo] + r L[c_hi])^2,  mt[kd][0][c] = mu_kd (1-r) L[c],  mt[kd][1][c] = mu_kd r L[c]
template <bool NU>
__global__ void __launch_bounds__(256) k_fold_r4tab(DevCrt t, const u64 *lut, Fq3Const r, const Fq3Const *mu_pow, u32 nkd, u64 *sq, u64 *mt) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const Fq3 rr = fq3_make(r.c[0], r.c[1], r.c[2]);
    auto L = [&](u32 c) { return fq3_make(lut[3 * c], lut[3 * c + 1], lut[3 * c + 2]); };
    if (i < 6561) {
        const u32 c0 = i / 81, c1 = i % 81;
        const Fq3 l0 = L(c0), f = fq3_add(l0, M3<NU>(fq3_sub(L(c1), l0), rr, t.nu)), sv = M3<NU>(f, f, t.nu);
        u64 *o = sq + (size_t)i * 4;
        o[0] = sv.c[0]; o[1] = sv.c[1]; o[2] = sv.c[2]; o[3] = 0;
    } else if (i < 6561 + nkd * 162) {
        const u32 j = i - 6561, kd = j / 162, w = (j % 162) / 81, c = j % 81;
        const Fq3 l = L(c), rl = M3<NU>(l, rr, t.nu);
        const Fq3Const mc = mu_pow[kd];
        const Fq3 m = M3<NU>(fq3_make(mc.c[0], mc.c[1], mc.c[2]), w ? rl : fq3_sub(l, rl), t.nu);
        u64 *o = mt + (((size_t)kd * 2 + w) * 81 + c) * 4;
        o[0] = m.c[0]; o[1] = m.c[1]; o[2] = m.c[2]; o[3] = 0;
    }
}
// tables of mode 7: T[0..3] = A', B', C', D' (see above); xx[c0 * 81 + c1] = (A'[c0] + B'[c1])^2, yy[c2 * 81 + c3] = (C'[c2] + D'[c3])^2, mt[kd][w][c] = mu_kd T[w][c]
template <bool NU>
__device__ __forceinline__ Fq3 r5_entry(const u64 *lut, u32 w, u32 c, Fq3 r3, Fq3 r4, u64 nu) {
    const Fq3 l = fq3_make(lut[3 * c], lut[3 * c + 1], lut[3 * c + 2]), rl = M3<NU>(l, r3, nu), a = (w & 1) ? rl : fq3_sub(l, rl), ra = M3<NU>(a, r4, nu);
    return (w & 2) ? ra : fq3_sub(a, ra);
}
template <bool NU>
__global__ void __launch_bounds__(256) k_fold_r5tab(DevCrt t, const u64 *lut, Fq3Const r3c, Fq3Const r4c, const Fq3Const *mu_pow, u32 nkd, u64 *xx, u64 *yy, u64 *mt) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    const Fq3 r3 = fq3_make(r3c.c[0], r3c.c[1], r3c.c[2]), r4 = fq3_make(r4c.c[0], r4c.c[1], r4c.c[2]);
    if (i < 2 * 6561) {
        const u32 hi = i / 6561, j = i % 6561, c0 = j / 81, c1 = j % 81;
        const Fq3 f = fq3_add(r5_entry<NU>(lut, 2 * hi, c0, r3, r4, t.nu), r5_entry<NU>(lut, 2 * hi + 1, c1, r3, r4, t.nu)), sv = M3<NU>(f, f, t.nu);
        u64 *o = (hi ? yy : xx) + (size_t)j * 4;
        o[0] = sv.c[0]; o[1] = sv.c[1]; o[2] = sv.c[2]; o[3] = 0;
    } else if (i < 2 * 6561 + nkd * 324) {
        const u32 j = i - 2 * 6561, kd = j / 324, w = (j % 324) / 81, c = j % 81;
        const Fq3Const mc = mu_pow[kd];
        const Fq3 m = M3<NU>(fq3_make(mc.c[0], mc.c[1], mc.c[2]), r5_entry<NU>(lut, w, c, r3, r4, t.nu), t.nu);
        u64 *o = mt + (((size_t)kd * 4 + w) * 81 + c) * 4;
        o[0] = m.c[0]; o[1] = m.c[1]; o[2] = m.c[2]; o[3] = 0;
    }
}
// digit code of four consecutive plane entries at bit k: 40 + sum_b sign_b * bit_k(|v_b|) * 3^b
__device__ __forceinline__ u32 digit_code4(const int32_t *v, u32 k) {
    int code = 40;
    const int w[4] = {1, 3, 9, 27};
#pragma unroll
    for (int b = 0; b < 4; b++) {
        int32_t x = v[b], mg = x < 0 ? -x : x;
        int bit = (mg >> k) & 1;
        code += x < 0 ? -bit * w[b] : bit * w[b];
    }
    return (u32)code;
}
// SPLIT (modes 1, 6, 7): eqB fixed at r_1..r_{i-1} is c_i eq(beta_i, b) E_i[p] at entry 2p + b, so the norm part of the message is c_i eq(beta_i, X) T(X) with
// T(X) = sum_p E_i[p] (Q0 + Q1 X + Q2 X^2 + Q3 X^3)(p).  The kernel leaves  sum_p E_i[p] Q_e(p), e = 0..2  (three values per slot instead of five evaluations; the G part
// comes from k_fold_round_g); the host takes sum E Q3 from g(0) + g(1) = the previous message at its challenge.  Q3 is the only coefficient that needs the
// fourth lazy product of a table (P3): three products per table instead of four.
template <bool NU, int MODE, bool SPLIT = false>
__global__ void __launch_bounds__(256) k_fold_round(DevCrt t, FoldRoundArgs a, const u64 *F, size_t ldF, u32 K, const Fq3Const *mu_pow,
                                                    FoldSrc src, u64 *partial) {
    static_assert(!SPLIT || (NU && (MODE == 1 || MODE == 6 || MODE == 7)), "split form: the large-round modes of the 2^40 non-residue path");
    constexpr int NA = SPLIT ? 3 : 5;   // SPLIT: the G part comes from its own small kernel (k_fold_round_g) -- three live accumulators instead of five
    u32 slot = blockIdx.y;
    const size_t pend = a.p0 + a.pcnt;
    const u64 nu = t.nu;
    const u32 nkd = 2 * K * 3, per = (nkd + gridDim.z - 1) / gridDim.z;
    const u32 kd0 = blockIdx.z * per, kd1 = kd0 + per < nkd ? kd0 + per : nkd;
    if (MODE == 0) F -= 2 * a.pF0;  // the f-hat buffer starts at pair a.pF0 (sharded rounds hold only the rank's slice)
    if (MODE == 1) { F -= 4 * a.pF0; src.out -= 2 * a.pF0; }   // fused fix: previous tables from entry 4 pF0, the fixed ones from entry 2 pF0
    if ((MODE == 4 || MODE == 6 || MODE == 7) && src.out) src.out -= 2 * a.pF0;   // first materialised tables of a rank's slice
    const Fq3 rfix = fq3_make(src.r.c[0], src.r.c[1], src.r.c[2]);
    __shared__ u64 slut[MODE == 7 ? 4 * 81 * 3 : (MODE >= 3 ? 3 * 81 * 3 : 1)];   // (mode 5 uses the values only)   // the 81 values, their squares, (mode 4) r times the values
    if (MODE == 7) {   // the four tables A', B', C', D'
        for (u32 i = threadIdx.x; i < 4 * 81; i += 256) {
            const Fq3 e = r5_entry<NU>(src.lut, i / 81, i % 81, fq3_make(src.r_prev.c[0], src.r_prev.c[1], src.r_prev.c[2]), rfix, nu);
            slut[3 * i] = e.c[0]; slut[3 * i + 1] = e.c[1]; slut[3 * i + 2] = e.c[2];
        }
        __syncthreads();
    } else
    if (MODE >= 3) {
        for (u32 i = threadIdx.x; i < 2 * 81 * 3; i += 256) slut[i] = src.lut[i];
        __syncthreads();
        if (MODE == 4 || MODE == 6) {   // fix_variables on look-up values needs no product per entry: f = g0 + r g1 - r g0
            if (threadIdx.x < 81) {
                Fq3 rv = M3<NU>(fq3_make(slut[3 * threadIdx.x], slut[3 * threadIdx.x + 1], slut[3 * threadIdx.x + 2]), rfix, nu);
                slut[3 * (162 + threadIdx.x)] = rv.c[0]; slut[3 * (162 + threadIdx.x) + 1] = rv.c[1]; slut[3 * (162 + threadIdx.x) + 2] = rv.c[2];
            }
            __syncthreads();
        }
    }
    auto lut3 = [&](u32 code) { return fq3_make(slut[3 * code], slut[3 * code + 1], slut[3 * code + 2]); };
    Fq3 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) acc[i] = fq3_zero();
    for (size_t p = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; p < pend; p += (size_t)gridDim.x * 256) {
        if constexpr (!SPLIT)
        if (blockIdx.z == 0) fold_g13_evals<NU>(acc, a, slot, p, nu);   // per pair here: three more live F_{p^3} accumulators cost this kernel its occupancy (rounds 4-6: 5.0 -> 6.7 ms)
        // pair of table kd
        auto load_pair = [&](u32 kd, Fq3 &f0, Fq3 &df) {
            if (MODE == 0) {
                const u64 *fp = F + ((size_t)kd * 24 + 3 * slot) * ldF;
                ulonglong2 x0 = *(const ulonglong2 *)(fp + 2 * p), x1 = *(const ulonglong2 *)(fp + ldF + 2 * p), x2 = *(const ulonglong2 *)(fp + 2 * ldF + 2 * p);
                f0 = fq3_make(x0.x, x1.x, x2.x);
                df = fq3_sub(fq3_make(x0.y, x1.y, x2.y), f0);
            } else if (MODE == 1) {
                const u64 *fp = F + ((size_t)kd * 24 + 3 * slot) * ldF + 4 * p;
                ulonglong2 a0 = *(const ulonglong2 *)(fp), a1 = *(const ulonglong2 *)(fp + ldF), a2 = *(const ulonglong2 *)(fp + 2 * ldF);
                ulonglong2 b0 = *(const ulonglong2 *)(fp + 2), b1 = *(const ulonglong2 *)(fp + ldF + 2), b2 = *(const ulonglong2 *)(fp + 2 * ldF + 2);
                Fq3 lo = fq3_make(a0.x, a1.x, a2.x), hi = fq3_make(b0.x, b1.x, b2.x);
                f0 = fq3_add(lo, M3<NU>(fq3_sub(fq3_make(a0.y, a1.y, a2.y), lo), rfix, nu));
                Fq3 f1 = fq3_add(hi, M3<NU>(fq3_sub(fq3_make(b0.y, b1.y, b2.y), hi), rfix, nu));
                u64 *op = src.out + ((size_t)kd * 24 + 3 * slot) * src.ldo + 2 * p;
                *(ulonglong2 *)(op) = make_ulonglong2(f0.c[0], f1.c[0]);
                *(ulonglong2 *)(op + src.ldo) = make_ulonglong2(f0.c[1], f1.c[1]);
                *(ulonglong2 *)(op + 2 * src.ldo) = make_ulonglong2(f0.c[2], f1.c[2]);
                df = fq3_sub(f1, f0);
            } else {
                constexpr int NE = MODE == 4 ? 16 : 8;      // plane entries behind one pair
                const u32 side = kd / (3 * K), k = (kd / 3) % K, d = kd % 3;
                const int32_t *pl = (side ? src.planesR : src.planesL) + (size_t)(d * 8 + slot) * src.n_planes + (size_t)NE * p;
                int32_t v[NE];
                if ((size_t)NE * p + NE <= src.n_planes && (src.n_planes & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < NE / 4; q++) {
                        int4 w = *(const int4 *)(pl + 4 * q);
                        v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NE; q++) v[q] = (size_t)NE * p + q < src.n_planes ? pl[q] : 0;
                }
                if (MODE != 4) {
                    f0 = lut3(digit_code4(v, k));
                    df = fq3_sub(lut3(digit_code4(v + 4, k)), f0);
                } else {
                    const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k), c2 = digit_code4(v + 8, k), c3 = digit_code4(v + 12, k);
                    f0 = fq3_add(lut3(c0), fq3_sub(lut3(162 + c1), lut3(162 + c0)));
                    Fq3 f1 = fq3_add(lut3(c2), fq3_sub(lut3(162 + c3), lut3(162 + c2)));
                    u64 *op = src.out + ((size_t)kd * 24 + 3 * slot) * src.ldo + 2 * p;
                    *(ulonglong2 *)(op) = make_ulonglong2(f0.c[0], f1.c[0]);
                    *(ulonglong2 *)(op + src.ldo) = make_ulonglong2(f0.c[1], f1.c[1]);
                    *(ulonglong2 *)(op + 2 * src.ldo) = make_ulonglong2(f0.c[2], f1.c[2]);
                    df = fq3_sub(f1, f0);
                }
            }
        };
        Fq3 Q[4];
        if (NU && MODE == 5) {
            // Round 3 with per-table products of the look-up values (k_fold_mutab): with T1 = mu L, T2 = mu L^2, T3 = mu L^3
            //   P0 = sum T3[c0], P3 = sum T3[c1]  (look-ups and additions),  P1 = sum T2[c0] * L[c1],  P2 = sum T2[c1] * L[c0]  (two lazy products)
            LH5 A1, A2;
            lh5_zero(A1); lh5_zero(A2);
            u64 s64[12];     // lazy 64-bit sums with carry counters: P0, P3, sp, su (3 words each)
            u32 scy[12];
#pragma unroll
            for (int i = 0; i < 12; i++) { s64[i] = 0; scy[i] = 0; }
            auto ladd = [&](int base, const ulonglong2 &a, u64 b) {
                const u64 w[3] = {a.x, a.y, b};
#pragma unroll
                for (int q = 0; q < 3; q++) { u64 sm = s64[base + q] + w[q]; scy[base + q] += sm < w[q]; s64[base + q] = sm; }
            };
            const u32 nkd_all = 2 * K * 3;
            for (u32 kd = kd0; kd < kd1; kd++) {
                const u32 side = kd / (3 * K), k = (kd / 3) % K, d = kd % 3;
                const int32_t *pl = (side ? src.planesR : src.planesL) + (size_t)(d * 8 + slot) * src.n_planes + (size_t)8 * p;
                int32_t v[8];
                if ((size_t)8 * p + 8 <= src.n_planes && (src.n_planes & 3) == 0) {
                    int4 w0 = *(const int4 *)pl, w1 = *(const int4 *)(pl + 4);
                    v[0] = w0.x; v[1] = w0.y; v[2] = w0.z; v[3] = w0.w; v[4] = w1.x; v[5] = w1.y; v[6] = w1.z; v[7] = w1.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = (size_t)8 * p + q < src.n_planes ? pl[q] : 0;
                }
                const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k);
                const u64 *t1 = src.mutab + ((size_t)kd * 81) * 4, *t2 = t1 + (size_t)nkd_all * 81 * 4, *t3 = t2 + (size_t)nkd_all * 81 * 4;
                ulonglong2 m10 = *(const ulonglong2 *)(t1 + 4 * c0), m11 = *(const ulonglong2 *)(t1 + 4 * c1);
                ulonglong2 m20 = *(const ulonglong2 *)(t2 + 4 * c0), m21 = *(const ulonglong2 *)(t2 + 4 * c1);
                ulonglong2 m30 = *(const ulonglong2 *)(t3 + 4 * c0), m31 = *(const ulonglong2 *)(t3 + 4 * c1);
                u64 m10c = t1[4 * c0 + 2], m11c = t1[4 * c1 + 2], m20c = t2[4 * c0 + 2], m21c = t2[4 * c1 + 2], m30c = t3[4 * c0 + 2], m31c = t3[4 * c1 + 2];
                ladd(0, m30, m30c); ladd(3, m31, m31c); ladd(6, m10, m10c); ladd(9, m11, m11c);
                lh5_mac(A1, fq3_make(m20.x, m20.y, m20c), lut3(c1));
                lh5_mac(A2, fq3_make(m21.x, m21.y, m21c), lut3(c0));
            }
            auto lfin = [&](int base) {
                return fq3_make(fq_canon(fq_reduce128_loose(s64[base], (u64)scy[base])), fq_canon(fq_reduce128_loose(s64[base + 1], (u64)scy[base + 1])),
                                fq_canon(fq_reduce128_loose(s64[base + 2], (u64)scy[base + 2])));
            };
            Fq3 P0 = lfin(0), P3 = lfin(3), sp = lfin(6), su = lfin(9), P1 = lh5_finish(A1), P2 = lh5_finish(A2);
            Fq3 a1 = fq3_sub(P1, P0);
            Fq3 a2 = fq3_add(fq3_sub(P2, fq3_add(P1, P1)), P0);
            Fq3 p12 = fq3_sub(P1, P2);
            Fq3 a3 = fq3_add(fq3_sub(P3, P0), fq3_add(fq3_add(p12, p12), p12));
            Q[0] = fq3_sub(P0, sp);
            Q[1] = fq3_sub(fq3_add(fq3_add(a1, a1), a1), fq3_sub(su, sp));
            Q[2] = fq3_add(fq3_add(a2, a2), a2);
            Q[3] = a3;
        } else if (NU && MODE == 7) {
            LH5 A0, A1, A2, A3;
            lh5_zero(A0); lh5_zero(A1); lh5_zero(A2); lh5_zero(A3);
            Fq3 sp = fq3_zero(), su = fq3_zero();
            for (u32 kd = kd0; kd < kd1; kd++) {     // (pairing the tables as in mode 6 needs 408 registers here)
                const u32 side = kd / (3 * K), k = (kd / 3) % K, d = kd % 3;
                const int32_t *pl = (side ? src.planesR : src.planesL) + (size_t)(d * 8 + slot) * src.n_planes + (size_t)32 * p;
                const u64 *mt = src.mt5 + (size_t)kd * 4 * 81 * 4;
                Fq3 fv[2], sq[2], mf[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {     // the two entries of the pair: plane entries 16 e .. 16 e + 15
                    int32_t v[16];
                    if ((size_t)32 * p + 16 * e + 16 <= src.n_planes && (src.n_planes & 3) == 0) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            int4 w = *(const int4 *)(pl + 16 * e + 4 * q);
                            v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; q++) v[q] = (size_t)32 * p + 16 * e + q < src.n_planes ? pl[16 * e + q] : 0;
                    }
                    const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k), c2 = digit_code4(v + 8, k), c3 = digit_code4(v + 12, k);
                    const u64 *qx = src.xx5 + (size_t)(c0 * 81 + c1) * 4, *qy = src.yy5 + (size_t)(c2 * 81 + c3) * 4;
                    const ulonglong2 xa = *(const ulonglong2 *)qx, ya = *(const ulonglong2 *)qy;
                    const u64 xc = qx[2], yc = qy[2];
                    const ulonglong2 m0 = *(const ulonglong2 *)(mt + 4 * c0), m1 = *(const ulonglong2 *)(mt + 4 * (81 + c1));
                    const ulonglong2 m2 = *(const ulonglong2 *)(mt + 4 * (162 + c2)), m3 = *(const ulonglong2 *)(mt + 4 * (243 + c3));
                    const u64 m0c = mt[4 * c0 + 2], m1c = mt[4 * (81 + c1) + 2], m2c = mt[4 * (162 + c2) + 2], m3c = mt[4 * (243 + c3) + 2];
                    const Fq3 X = fq3_add(lut3(c0), lut3(81 + c1)), Y = fq3_add(lut3(162 + c2), lut3(243 + c3));
                    const Fq3 xy = fq3_mul_2p40(X, Y);
                    fv[e] = fq3_add(X, Y);
                    sq[e] = fq3_add(fq3_add(fq3_make(xa.x, xa.y, xc), fq3_make(ya.x, ya.y, yc)), fq3_add(xy, xy));
                    mf[e] = fq3_add(fq3_add(fq3_make(m0.x, m0.y, m0c), fq3_make(m1.x, m1.y, m1c)), fq3_add(fq3_make(m2.x, m2.y, m2c), fq3_make(m3.x, m3.y, m3c)));
                }
                u64 *op = src.out + ((size_t)kd * 24 + 3 * slot) * src.ldo + 2 * p;
                *(ulonglong2 *)(op) = make_ulonglong2(fv[0].c[0], fv[1].c[0]);
                *(ulonglong2 *)(op + src.ldo) = make_ulonglong2(fv[0].c[1], fv[1].c[1]);
                *(ulonglong2 *)(op + 2 * src.ldo) = make_ulonglong2(fv[0].c[2], fv[1].c[2]);
                lh5_mac(A0, mf[0], sq[0]); lh5_mac(A1, mf[1], sq[0]); lh5_mac(A2, mf[0], sq[1]);
                if constexpr (!SPLIT) lh5_mac(A3, mf[1], sq[1]);
                sp = fq3_add(sp, mf[0]); su = fq3_add(su, mf[1]);
            }
            Fq3 P0 = lh5_finish(A0), P1 = lh5_finish(A1), P2 = lh5_finish(A2);
            Fq3 a1 = fq3_sub(P1, P0);
            Fq3 a2 = fq3_add(fq3_sub(P2, fq3_add(P1, P1)), P0);
            Q[0] = fq3_sub(P0, sp);
            Q[1] = fq3_sub(fq3_add(fq3_add(a1, a1), a1), fq3_sub(su, sp));
            Q[2] = fq3_add(fq3_add(a2, a2), a2);
            if constexpr (!SPLIT) {
                Fq3 P3 = lh5_finish(A3), p12 = fq3_sub(P1, P2);
                Q[3] = fq3_add(fq3_sub(P3, P0), fq3_add(fq3_add(p12, p12), p12));
            } else Q[3] = fq3_zero();
        } else if (NU && MODE == 6) {
            LH5 A0, A1, A2, A3;
            lh5_zero(A0); lh5_zero(A1); lh5_zero(A2); lh5_zero(A3);
            Fq3 sp = fq3_zero(), su = fq3_zero();
            // operands of table kd's four lazy products: gathers, no multiplication (and the fixed pair stored for round 5)
            auto gen = [&](u32 kd, Fq3 &tt, Fq3 &uu, Fq3 &s0, Fq3 &s1) {
                const u32 side = kd / (3 * K), k = (kd / 3) % K, d = kd % 3;
                const int32_t *pl = (side ? src.planesR : src.planesL) + (size_t)(d * 8 + slot) * src.n_planes + (size_t)16 * p;
                int32_t v[16];
                if ((size_t)16 * p + 16 <= src.n_planes && (src.n_planes & 3) == 0) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        int4 w = *(const int4 *)(pl + 4 * q);
                        v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q++) v[q] = (size_t)16 * p + q < src.n_planes ? pl[q] : 0;
                }
                const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k), c2 = digit_code4(v + 8, k), c3 = digit_code4(v + 12, k);
                const u64 *q0 = src.sq4 + (size_t)(c0 * 81 + c1) * 4, *q1 = src.sq4 + (size_t)(c2 * 81 + c3) * 4;
                const u64 *ma = src.mt4 + (size_t)kd * 2 * 81 * 4, *mb = ma + 81 * 4;
                const ulonglong2 s0a = *(const ulonglong2 *)q0, s1a = *(const ulonglong2 *)q1;
                const u64 s0c = q0[2], s1c = q1[2];
                const ulonglong2 a0 = *(const ulonglong2 *)(ma + 4 * c0), b1 = *(const ulonglong2 *)(mb + 4 * c1);
                const ulonglong2 a2 = *(const ulonglong2 *)(ma + 4 * c2), b3 = *(const ulonglong2 *)(mb + 4 * c3);
                const u64 a0c = ma[4 * c0 + 2], b1c = mb[4 * c1 + 2], a2c = ma[4 * c2 + 2], b3c = mb[4 * c3 + 2];
                if (src.out) {      // (null when round 5 works from the planes as well: mode 7)
                    const Fq3 f0 = fq3_add(lut3(c0), fq3_sub(lut3(162 + c1), lut3(162 + c0)));
                    const Fq3 f1 = fq3_add(lut3(c2), fq3_sub(lut3(162 + c3), lut3(162 + c2)));
                    u64 *op = src.out + ((size_t)kd * 24 + 3 * slot) * src.ldo + 2 * p;
                    *(ulonglong2 *)(op) = make_ulonglong2(f0.c[0], f1.c[0]);
                    *(ulonglong2 *)(op + src.ldo) = make_ulonglong2(f0.c[1], f1.c[1]);
                    *(ulonglong2 *)(op + 2 * src.ldo) = make_ulonglong2(f0.c[2], f1.c[2]);
                }
                tt = fq3_add(fq3_make(a0.x, a0.y, a0c), fq3_make(b1.x, b1.y, b1c));
                uu = fq3_add(fq3_make(a2.x, a2.y, a2c), fq3_make(b3.x, b3.y, b3c));
                s0 = fq3_make(s0a.x, s0a.y, s0c);
                s1 = fq3_make(s1a.x, s1a.y, s1c);
                sp = fq3_add(sp, tt); su = fq3_add(su, uu);
            };
            u32 kd = kd0;
            for (; kd + 1 < kd1; kd += 2) {     // two tables per iteration: their partial products share the column sums (lh5_mac2)
                Fq3 tA, uA, xA, yA, tB, uB, xB, yB;
                gen(kd, tA, uA, xA, yA);
                gen(kd + 1, tB, uB, xB, yB);
                lh5_mac2(A0, tA, xA, tB, xB); lh5_mac2(A1, uA, xA, uB, xB); lh5_mac2(A2, tA, yA, tB, yB);
                if constexpr (!SPLIT) lh5_mac2(A3, uA, yA, uB, yB);
            }
            if (kd < kd1) {
                Fq3 tA, uA, xA, yA;
                gen(kd, tA, uA, xA, yA);
                lh5_mac(A0, tA, xA); lh5_mac(A1, uA, xA); lh5_mac(A2, tA, yA);
                if constexpr (!SPLIT) lh5_mac(A3, uA, yA);
            }
            Fq3 P0 = lh5_finish(A0), P1 = lh5_finish(A1), P2 = lh5_finish(A2);
            Fq3 a1 = fq3_sub(P1, P0);                                           // sum mu f0^2 df
            Fq3 a2 = fq3_add(fq3_sub(P2, fq3_add(P1, P1)), P0);                 // sum mu f0 df^2
            Q[0] = fq3_sub(P0, sp);
            Q[1] = fq3_sub(fq3_add(fq3_add(a1, a1), a1), fq3_sub(su, sp));
            Q[2] = fq3_add(fq3_add(a2, a2), a2);
            if constexpr (!SPLIT) {
                Fq3 P3 = lh5_finish(A3), p12 = fq3_sub(P1, P2);
                Q[3] = fq3_add(fq3_sub(P3, P0), fq3_add(fq3_add(p12, p12), p12)); // sum mu df^3
            } else Q[3] = fq3_zero();
        } else if (NU && MODE == 3) {
            // Both ends of a pair are look-up values, so their squares are too: with t = mu f0, u = mu f1 the four lazy sums
            //   P0 = sum t f0^2, P1 = sum u f0^2, P2 = sum t f1^2, P3 = sum u f1^2   (= sum mu f0^3, mu f0^2 f1, mu f0 f1^2, mu f1^3)
            // need two reduced products per table instead of four; the cubic coefficients in df = f1 - f0 follow by binomials.
            LH5 A0, A1, A2, A3;
            lh5_zero(A0); lh5_zero(A1); lh5_zero(A2); lh5_zero(A3);
            Fq3 sp = fq3_zero(), su = fq3_zero();
            for (u32 kd = kd0; kd < kd1; kd++) {
                const u32 side = kd / (3 * K), k = (kd / 3) % K, d = kd % 3;
                const int32_t *pl = (side ? src.planesR : src.planesL) + (size_t)(d * 8 + slot) * src.n_planes + (size_t)8 * p;
                int32_t v[8];
                if ((size_t)8 * p + 8 <= src.n_planes && (src.n_planes & 3) == 0) {
                    int4 w0 = *(const int4 *)pl, w1 = *(const int4 *)(pl + 4);
                    v[0] = w0.x; v[1] = w0.y; v[2] = w0.z; v[3] = w0.w; v[4] = w1.x; v[5] = w1.y; v[6] = w1.z; v[7] = w1.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = (size_t)8 * p + q < src.n_planes ? pl[q] : 0;
                }
                const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k);
                Fq3 f0 = lut3(c0), f1 = lut3(c1), s0 = lut3(81 + c0), s1 = lut3(81 + c1);
                Fq3Const mc = mu_pow[kd];
                Fq3 mu = fq3_make(mc.c[0], mc.c[1], mc.c[2]);
                Fq3 tt = fq3_mul_2p40(mu, f0), uu = fq3_mul_2p40(mu, f1);
                lh5_mac(A0, tt, s0); lh5_mac(A1, uu, s0); lh5_mac(A2, tt, s1); lh5_mac(A3, uu, s1);
                sp = fq3_add(sp, tt); su = fq3_add(su, uu);
            }
            Fq3 P0 = lh5_finish(A0), P1 = lh5_finish(A1), P2 = lh5_finish(A2), P3 = lh5_finish(A3);
            Fq3 a1 = fq3_sub(P1, P0);                                           // sum mu f0^2 df
            Fq3 a2 = fq3_add(fq3_sub(P2, fq3_add(P1, P1)), P0);                 // sum mu f0 df^2
            Fq3 p12 = fq3_sub(P1, P2);
            Fq3 a3 = fq3_add(fq3_sub(P3, P0), fq3_add(fq3_add(p12, p12), p12)); // sum mu df^3 = P3 - 3 P2 + 3 P1 - P0
            Q[0] = fq3_sub(P0, sp);
            Q[1] = fq3_sub(fq3_add(fq3_add(a1, a1), a1), fq3_sub(su, sp));
            Q[2] = fq3_add(fq3_add(a2, a2), a2);
            Q[3] = a3;
        } else if (NU) {
            // sum_kd mu (f0 + X df)^3 - mu (f0 + X df):  with p = mu f0, q = mu df the cubic coefficients are
            //   sum p f0^2,  3 sum q f0^2,  3 sum p df^2,  sum q df^2   -- four LAZY sums, 4 reduced products per table
            LH5 A0, A1, A2, A3;
            lh5_zero(A0); lh5_zero(A1); lh5_zero(A2); lh5_zero(A3);
            Fq3 sp = fq3_zero(), sq = fq3_zero();
            // (pairing the tables of a step as in mode 6 -- lh5_mac2 -- costs this branch its second wave per SIMD: 268 registers with the fused fix)
            for (u32 kd = kd0; kd < kd1; kd++) {
                Fq3 f0, df;
                load_pair(kd, f0, df);
                Fq3Const mc = mu_pow[kd];
                Fq3 mu = fq3_make(mc.c[0], mc.c[1], mc.c[2]);
                Fq3 f0s = fq3_mul_2p40(f0, f0), dfs = fq3_mul_2p40(df, df);
                Fq3 pp = fq3_mul_2p40(mu, f0), qq = fq3_mul_2p40(mu, df);
                lh5_mac(A0, pp, f0s); lh5_mac(A1, qq, f0s); lh5_mac(A2, pp, dfs);
                if constexpr (!SPLIT) lh5_mac(A3, qq, dfs);
                sp = fq3_add(sp, pp); sq = fq3_add(sq, qq);
            }
            Fq3 t1 = lh5_finish(A1), t2 = lh5_finish(A2);
            Q[0] = fq3_sub(lh5_finish(A0), sp);
            Q[1] = fq3_sub(fq3_add(fq3_add(t1, t1), t1), sq);
            Q[2] = fq3_add(fq3_add(t2, t2), t2);
            if constexpr (!SPLIT) Q[3] = lh5_finish(A3);
            else Q[3] = fq3_zero();
        } else {
            Q[0] = Q[1] = Q[2] = Q[3] = fq3_zero();
            for (u32 kd = kd0; kd < kd1; kd++) {
                Fq3 f0, df;
                load_pair(kd, f0, df);
                Fq3 f0s = S3<NU>(f0, nu), dfs = S3<NU>(df, nu);
                Fq3 c0 = fq3_sub(M3<NU>(f0s, f0, nu), f0);
                Fq3 c3 = M3<NU>(dfs, df, nu);
                Fq3 t1 = M3<NU>(f0s, df, nu), t2 = M3<NU>(dfs, f0, nu);
                Fq3 c1 = fq3_sub(fq3_add(fq3_add(t1, t1), t1), df);
                Fq3 c2 = fq3_add(fq3_add(t2, t2), t2);
                Fq3Const mc = mu_pow[kd];
                Fq3 mu = fq3_make(mc.c[0], mc.c[1], mc.c[2]);
                Q[0] = fq3_add(Q[0], M3<NU>(c0, mu, nu));
                Q[1] = fq3_add(Q[1], M3<NU>(c1, mu, nu));
                Q[2] = fq3_add(Q[2], M3<NU>(c2, mu, nu));
                Q[3] = fq3_add(Q[3], M3<NU>(c3, mu, nu));
            }
        }
        if constexpr (SPLIT) {
            const Fq3 E = fq3_make(src.E[p], src.E[src.ldE + p], src.E[2 * src.ldE + p]);
#pragma unroll
            for (int e = 0; e < 3; e++) acc[e] = fq3_add(acc[e], M3<NU>(Q[e], E, nu));
        } else fold_g2_finish<NU>(acc, Q, a, p, nu);
    }
    // partial row = blockIdx.x + gridDim.x * blockIdx.z
    u64 vv[3 * NA];
#pragma unroll
    for (int i = 0; i < NA; i++) { vv[3 * i] = acc[i].c[0]; vv[3 * i + 1] = acc[i].c[1]; vv[3 * i + 2] = acc[i].c[2]; }
    __shared__ u64 red[3 * NA];
    block_sum_store<3 * NA>(vv, red);
    __syncthreads();
    size_t row = (size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.z;
    if (threadIdx.x < 3 * NA) partial[row * (24 * NA) + (threadIdx.x / 3) * 24 + 3 * slot + threadIdx.x % 3] = red[threadIdx.x];
}
// threads a round should have before its tables stop being split over blockIdx.z (LF_FOLD_CHUNK_THREADS; a thread walks its chunk of the 96 tables serially)
static size_t fold_chunk_threads() {
    static const size_t v = [] { const char *e = getenv("LF_FOLD_CHUNK_THREADS"); return e ? (size_t)atoll(e) : (size_t)131072; }();   // measured at C4: 65536 / 131072 / 262144 / 524288 -> 19.58 / 19.28 / 19.35 / 19.74 ms per step
    return v;
}
template <int MODE>
static void launch_fold_round_mode(const DevCrt &t, const FoldRoundArgs &a, const u64 *F, size_t ldF, u32 K, const Fq3Const *mu_pow_dev,
                                   const FoldSrc &src, u64 *partial, u64 *out, hipStream_t s) {
    size_t pairs = a.pcnt;
    u32 gb = (u32)((pairs + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    // aim for >= 64k threads: split the 2K*3 tables when pairs*8 is small (chunk count divides into RED_BLOCKS rows)
    u32 nkd = 2 * K * 3, chunks = 1;
    static const u32 cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 96};
    for (u32 cc : cand) {
        if (cc > nkd) break;
        chunks = cc;
        if (pairs * 8 * cc >= fold_chunk_threads()) break;
    }
    while (gb * chunks > RED_BLOCKS && chunks > 1) chunks--;
    if (MODE >= 3) chunks = 1;   // the planes of one (side, d) serve all K tables: no table split (the driver uses these modes on large rounds only)
    if constexpr (MODE == 1 || MODE == 6 || MODE == 7) {
        if (src.E && t.nu2p40) {   // split form: three sums per slot (see k_fold_round); the caller adds the G part (launch_fold_round_g)
            hipLaunchKernelGGL((k_fold_round<true, MODE, true>), dim3(gb, 8, chunks), dim3(256), 0, s, t, a, F, ldF, K, mu_pow_dev, src, partial);
            hipLaunchKernelGGL(k_reduce_rows, dim3(3 * 24), dim3(256), 0, s, partial, gb * chunks, 72, out);
            return;
        }
    }
    if (t.nu2p40) hipLaunchKernelGGL((k_fold_round<true, MODE>), dim3(gb, 8, chunks), dim3(256), 0, s, t, a, F, ldF, K, mu_pow_dev, src, partial);
    else hipLaunchKernelGGL((k_fold_round<false, MODE>), dim3(gb, 8, chunks), dim3(256), 0, s, t, a, F, ldF, K, mu_pow_dev, src, partial);
    hipLaunchKernelGGL(k_reduce_rows, dim3(5 * 24), dim3(256), 0, s, partial, gb * chunks, 120, out);
}
void launch_fold_round(const DevCrt &t, const FoldRoundArgs &a, const u64 *F, size_t ldF, u32 K, const Fq3Const *mu_pow_dev, u64 *partial,
                       u64 *out, hipStream_t s) {
    FoldSrc src = {};
    launch_fold_round_mode<0>(t, a, F, ldF, K, mu_pow_dev, src, partial, out, s);
}
// ---------------------------------------------------------------------------------------------------------
// Poseidon sponge on the device.  PoseidonTranscript (transcript/poseidon.rs:29-75) = arkworks-0.4 duplex sponge,
// width 24 = 4 capacity + 20 rate, 8 full + 22 partial rounds, alpha 7; round = ARK -> S-box -> MDS (row . state), the textbook form
// of lf_host.cpp's permute_plain.  One wave runs one sponge: lane i < 24 owns state word i (kept in LDS), the MDS row of a lane is a
// 24-term lazy dot product.  A permutation is a serial chain (30 rounds x (S-box of 4 dependent modmuls + a 24-term dot product)):
// ~20 us on one wave against 1.5 us on a host core with AVX-512 IFMA -- which is why the default transcript stays on the host and the
// device sponge is the opt-in LF_DEVICE_TRANSCRIPT=1 mode of the persistent tail (bit-identical proofs, slower).
struct SpongeDev {   // uniform across the wave
    int idx;         // next absorb / squeeze position in the rate
    int squeezing;
};
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
__device__ void poseidon_permute_wave(u64 *st /* LDS [24] */, u64 *tmp /* LDS [24] */, const u64 *ark, const u64 *mds) {
    const int lane = threadIdx.x & 63;
    for (int r = 0; r < 30; r++) {
        const bool full = r < 4 || r >= 26;
        if (lane < 24) {
            u64 x = fq_add(st[lane], ark[r * 24 + lane]);
            if (full || lane == 0) {
                u64 x2 = fq_mul(x, x), x4 = fq_mul(x2, x2), x6 = fq_mul(x4, x2);
                x = fq_mul(x6, x);
            }
            tmp[lane] = fq_canon(x);
        }
        wave_lds_sync();
        if (lane < 24) {
            const u64 *row = mds + lane * 24;
            Acc a;
            acc_set(a, row[0], tmp[0]);
#pragma unroll 4
            for (int j = 1; j < 24; j++) acc_mad(a, row[j], tmp[j]);
            st[lane] = fq_canon(acc_reduce(a));
        }
        wave_lds_sync();
    }
}
// sponge.absorb / squeeze exactly as lf_host.cpp::Transcript (arkworks duplex: a squeeze after an absorb permutes first and vice versa)
__device__ void sponge_absorb_wave(SpongeDev &sp, u64 *st, u64 *tmp, const u64 *ark, const u64 *mds, const u64 *x /* LDS or global */, int n) {
    const int lane = threadIdx.x & 63;
    if (n <= 0) return;
    int idx;
    if (!sp.squeezing) {
        idx = sp.idx;
        if (idx == 20) { poseidon_permute_wave(st, tmp, ark, mds); idx = 0; }
    } else {
        poseidon_permute_wave(st, tmp, ark, mds);
        idx = 0;
    }
    for (;;) {
        const int take = idx + n <= 20 ? n : 20 - idx;
        if (lane < take) st[4 + idx + lane] = fq_add(st[4 + idx + lane], fq_canon(x[lane]));
        wave_lds_sync();
        if (take == n) { sp.squeezing = 0; sp.idx = idx + n; return; }
        poseidon_permute_wave(st, tmp, ark, mds);
        x += take; n -= take; idx = 0;
    }
}
__device__ void sponge_squeeze_wave(SpongeDev &sp, u64 *st, u64 *tmp, const u64 *ark, const u64 *mds, u64 *out /* LDS or global */, int n) {
    const int lane = threadIdx.x & 63;
    int idx;
    if (!sp.squeezing) { poseidon_permute_wave(st, tmp, ark, mds); idx = 0; }
    else {
        idx = sp.idx;
        if (idx == 20) { poseidon_permute_wave(st, tmp, ark, mds); idx = 0; }
    }
    for (;;) {
        const int take = idx + n <= 20 ? n : 20 - idx;
        if (lane < take) out[lane] = st[4 + idx + lane];
        wave_lds_sync();
        if (take == n) { sp.squeezing = 1; sp.idx = idx + n; return; }
        if (n != 20) poseidon_permute_wave(st, tmp, ark, mds);
        out += take; n -= take; idx = 0;
    }
}
// test / ABI kernel (lf_device_sponge): run a script of absorb / squeeze operations on a fresh sponge.  ops[i] = (kind << 24) | count,
// kind 0 absorb (consumes `count` words of `words`), 1 squeeze (`count` words appended to out).  state_out: 24 words + idx + mode.
__global__ void __launch_bounds__(64) k_sponge_script(const u64 *ark, const u64 *mds, const u32 *ops, u32 nops, const u64 *words, u64 *out, u64 *state_out) {
    __shared__ u64 st[24], tmp[24];
    const int lane = threadIdx.x;
    if (lane < 24) st[lane] = 0;
    wave_lds_sync();
    SpongeDev sp;
    sp.idx = 0; sp.squeezing = 0;
    for (u32 i = 0; i < nops; i++) {
        const u32 kind = ops[i] >> 24;
        const int cnt = (int)(ops[i] & 0xffffff);
        if (kind == 0) { sponge_absorb_wave(sp, st, tmp, ark, mds, words, cnt); words += cnt; }
        else { sponge_squeeze_wave(sp, st, tmp, ark, mds, out, cnt); out += cnt; }
    }
    if (lane < 24) state_out[lane] = st[lane];
    if (lane == 0) { state_out[24] = (u64)sp.idx; state_out[25] = (u64)sp.squeezing; }
}
void launch_sponge_script(const u64 *ark, const u64 *mds, const u32 *ops, u32 nops, const u64 *words, u64 *out, u64 *state_out, hipStream_t s) {
    hipLaunchKernelGGL(k_sponge_script, dim3(1), dim3(64), 0, s, ark, mds, ops, nops, words, out, state_out);
}

// ---------------------------------------------------------------------------------------------------------
// Persistent tail of the folding sumcheck (no host hop per round).  Once the tables are small the per-round cost
// is launches + stream synchronisation, not arithmetic (a round >= 11 at 2^20 rows: ~100 us of wall clock for ~5 us of wave
// time).  k_fold_tail runs ALL remaining rounds in one launch: per round it fixes the previous tables with the challenge (fused
// into the pair loads, like MODE 1 above, here for the five special tables too), evaluates the round polynomial, the last
// workgroup to finish reduces the partial sums and writes the message into host-mapped memory; the host -- which still owns the
// Poseidon transcript (a permutation is a serial chain of ~900 dependent 64-bit modmuls: 1.5 us on a host core, >8 us on a GPU
// wave) -- polls that mailbox, absorbs, squeezes and writes the challenge back; workgroup 0 polls it over PCIe and republishes it
// in device memory for the others.
//
// Data flow is workgroup-local by construction: workgroup (slot, z) owns the F_{p^3} rows (table kd, slot) of its table chunk and,
// for z = 0, the G rows of its slot, for ALL pairs and all rounds; the slot-constant eq tables are kept as private copies per
// workgroup (eqpriv).  So tables never travel between workgroups -- on this GPU that would mean between the eight XCDs' L2
// caches, and every agent-scope release/acquire fence writes back / invalidates a whole L2 (measured: ~300 us per round with
// fences in 256 workgroups).  What does cross workgroups -- 15 partial sums each, the round counter, the republished challenge --
// moves through agent-scope atomics only (memory-side, coherent without fences); the host mailbox through system-scope atomics.
// All workgroups must be co-resident (launch_fold_tail sizes the grid from the occupancy query); every wait is bounded by a wall
// clock timeout that aborts the whole kernel (mail->err), so a lost host cannot hang the GPU.
__device__ __forceinline__ u32 ld_sys_u32(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_sys_u64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u64 ld_dev_u64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev_u64(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_mem() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }   // all of this wave's memory operations have completed
constexpr u64 TAIL_TIMEOUT_TICKS = 400000000ull;   // wall_clock64 runs at 100 MHz: 4 s
constexpr u64 TAIL_ABORT_BIT = 1ull << 40;
#ifdef LF_TAIL_DEBUG
#define TAIL_STAMP(mail, rd, k) __hip_atomic_store((u64 *)&(mail)->dbg[rd][k], (u64)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#else
#define TAIL_STAMP(mail, rd, k) do { } while (0)
#endif

// wait for challenge `idx` of this launch (epoch): returns false on abort/timeout.  Called by thread 0 of every workgroup.
__device__ bool tail_wait_challenge(TailMail *mail, u64 *dev_chal, u32 idx, u32 epoch, bool leader, u64 *r_out) {
    u64 *slot = dev_chal + (size_t)idx * 4;
    const u64 t0 = wall_clock64();
    if (leader) {
        u32 st = 0;
        for (u32 it = 0;; it++) {
            if (ld_sys_u32((const u32 *)&mail->chal_seq[idx]) == epoch) { st = 1; TAIL_STAMP(mail, idx + 1, 0); break; }
            if ((it & 63) == 63) {
                if (ld_sys_u32((const u32 *)&mail->abort_seq) == epoch) break;
                if (wall_clock64() - t0 > TAIL_TIMEOUT_TICKS) { __hip_atomic_store((u32 *)&mail->err, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
            }
        }
        if (st) {
            u64 v[3];
            for (int q = 0; q < 3; q++) v[q] = ld_sys_u64((const u64 *)&mail->chal[idx][q]);   // issued after the flag was seen; the host wrote them before it
            for (int q = 0; q < 3; q++) st_dev_u64(slot + q, v[q]);
        }
        wait_mem();
        st_dev_u64(slot + 3, st ? (u64)epoch : ((u64)epoch | TAIL_ABORT_BIT));
        TAIL_STAMP(mail, idx + 1, 1);
    }
    for (u32 it = 0;; it++) {
        u64 v = ld_dev_u64(slot + 3);
        if (v == (u64)epoch) break;
        if (v == ((u64)epoch | TAIL_ABORT_BIT)) return false;
        if ((it & 255) == 255 && wall_clock64() - t0 > 2 * TAIL_TIMEOUT_TICKS) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    for (int q = 0; q < 3; q++) r_out[q] = ld_dev_u64(slot + q);
    return true;
}

template <bool NU>
__global__ void __launch_bounds__(256) k_fold_tail(DevCrt t, FoldTailArgs A) {
    const u32 slot = blockIdx.y;
    const u64 nu = t.nu;
    const u32 nkd = 2 * A.K * 3, per = (nkd + gridDim.z - 1) / gridDim.z;
    const u32 kd0 = blockIdx.z * per, kd1 = kd0 + per < nkd ? kd0 + per : nkd;
    const u32 nblocks = gridDim.y * gridDim.z, bid = blockIdx.z * gridDim.y + blockIdx.y;
    const bool leader = bid == 0;
    __shared__ u64 s_r[3];
    __shared__ u32 s_flag;
    __shared__ u64 red[15];
    // Private working set of this workgroup, 128-byte aligned so that no line is shared with another workgroup: rows
    // 0 eqL, 1 eqR, 2 eqB, 3 G1[slot], 4 G2[slot], 5.. the tables kd0..kd1 (each row = 3 planes), two buffers (ping-pong).
    const size_t half0 = A.n0 / 2, row_words = 3 * half0, nrow = 5 + per;
    const size_t buf_words = (nrow * row_words + 15) & ~(size_t)15;
    u64 *const priv = A.eqpriv + (size_t)bid * 2 * buf_words;
    size_t n_prev = A.n0;
    Fq3 r = fq3_make(A.r_first.c[0], A.r_first.c[1], A.r_first.c[2]);
    for (u32 rd = 0; rd < A.rounds; rd++) {
        if (rd > 0) {
            if (threadIdx.x == 0) {
                u64 rr[3] = {0, 0, 0};
                bool ok = tail_wait_challenge(A.mail, A.dev_chal, rd - 1, A.epoch, leader && !A.dev_transcript, rr);
                s_r[0] = rr[0]; s_r[1] = rr[1]; s_r[2] = rr[2];
                s_flag = ok ? 1u : 0u;
            }
            __syncthreads();   // also orders this workgroup's table stores of the previous round before the loads below
            if (!s_flag) return;
            r = fq3_make(s_r[0], s_r[1], s_r[2]);
            __syncthreads();   // s_flag / s_r are rewritten below only after every wave has read them
            if (leader && threadIdx.x == 0) TAIL_STAMP(A.mail, rd, 2);
        }
        const size_t n = n_prev / 2, pairs = n / 2, ldp = n_prev;
        const bool first = rd == 0, last = rd + 1 == A.rounds;
        const u64 *Pp = priv + (size_t)((rd + 1) & 1) * buf_words;   // previous round's private buffer (rows of leading dimension ldp)
        u64 *Pn = priv + (size_t)(rd & 1) * buf_words;               // this round's (leading dimension n)
        // source row q of the previous tables: the shared layout in the first tail round, the private buffer afterwards
        auto src_row = [&](u32 q) -> const u64 * {
            if (!first) return Pp + (size_t)q * 3 * ldp;
            if (q < 3) return A.T[0] + (size_t)q * 3 * ldp;
            if (q < 5) return A.T[0] + (size_t)(3 + 8 * (q - 3) + slot) * 3 * ldp;
            return A.F[0] + ((size_t)(kd0 + (q - 5)) * 24 + 3 * slot) * ldp;
        };
        // fix entries 4p..4p+3 of an F_{p^3} row (three planes of leading dimension ldp) -> pair (2p, 2p+1)
        auto fix_pair = [&](const u64 *row, size_t p, Fq3 &f0, Fq3 &f1) {
            const u64 *fp = row + 4 * p;
            ulonglong2 a0 = *(const ulonglong2 *)(fp), a1 = *(const ulonglong2 *)(fp + ldp), a2 = *(const ulonglong2 *)(fp + 2 * ldp);
            ulonglong2 b0 = *(const ulonglong2 *)(fp + 2), b1 = *(const ulonglong2 *)(fp + ldp + 2), b2 = *(const ulonglong2 *)(fp + 2 * ldp + 2);
            Fq3 lo = fq3_make(a0.x, a1.x, a2.x), hi = fq3_make(b0.x, b1.x, b2.x);
            f0 = fq3_add(lo, M3<NU>(fq3_sub(fq3_make(a0.y, a1.y, a2.y), lo), r, nu));
            f1 = fq3_add(hi, M3<NU>(fq3_sub(fq3_make(b0.y, b1.y, b2.y), hi), r, nu));
        };
        auto store_pair = [&](u32 q, size_t p, const Fq3 &f0, const Fq3 &f1) {   // private row q: plain 16-byte stores (stay in this XCD's L2)
            u64 *op = Pn + (size_t)q * 3 * n + 2 * p;
            *(ulonglong2 *)(op) = make_ulonglong2(f0.c[0], f1.c[0]);
            *(ulonglong2 *)(op + n) = make_ulonglong2(f0.c[1], f1.c[1]);
            *(ulonglong2 *)(op + 2 * n) = make_ulonglong2(f0.c[2], f1.c[2]);
        };
        Fq3 acc[5];
#pragma unroll
        for (int i = 0; i < 5; i++) acc[i] = fq3_zero();
        // work items = (pair, task): task 0/1 (z = 0 only) = the eq_L G_L / eq_R G_R products, the others one table each.  The round
        // polynomial is linear in the per-table sums, so every item adds its own contribution to acc and the items of a pair can sit
        // in different waves: late rounds (a handful of pairs) are a latency chain, and this cuts it to one table per thread.
        const u32 ntab = kd1 > kd0 ? kd1 - kd0 : 0u, gt = blockIdx.z == 0 ? 2u : 0u;   // (2K*3 need not be a multiple of the chunk count)
        const size_t items = pairs * (size_t)(ntab + gt);
        for (size_t it = threadIdx.x; it < items; it += 256) {
            const size_t p = it % pairs;
            const u32 task = (u32)(it / pairs);
            if (task < gt) {
                const u32 h = task;
                Fq3 ea, eb, ga, gb;
                fix_pair(src_row(h), p, ea, eb);
                fix_pair(src_row(3 + h), p, ga, gb);
                store_pair(h, p, ea, eb);
                store_pair(3 + h, p, ga, gb);
                Fq3 co[3];
                co[0] = M3<NU>(ea, ga, nu);
                co[2] = M3<NU>(fq3_sub(eb, ea), fq3_sub(gb, ga), nu);
                co[1] = fq3_sub(fq3_sub(M3<NU>(eb, gb, nu), co[0]), co[2]);
                add_poly_evals<5>(acc, co, 3);
                continue;
            }
            const u32 q = task - gt, kd = kd0 + q;
            Fq3 bq0, bq1;
            fix_pair(src_row(2), p, bq0, bq1);
            if (q == 0) store_pair(2, p, bq0, bq1);
            // mu_kd ((f0 + X df)^3 - (f0 + X df)) of this table (same algebra as k_fold_round)
            Fq3 f0, f1;
            fix_pair(src_row(5 + q), p, f0, f1);
            store_pair(5 + q, p, f0, f1);
            if (last) {
                // the fully fixed tables (2 entries) go back to the shared layout for the theta kernel: WRITE-THROUGH stores.  Rows of
                // different workgroups share 128-byte lines there, the eight L2s are not coherent with each other, and a line that
                // was read and then partly written with plain stores is written back as a whole at kernel end -- stale neighbour
                // bytes included (seen: theta wrong in random slots).
                u64 *op = A.F[1] + ((size_t)kd * 24 + 3 * slot) * n + 2 * p;
                st_dev_u64(op, f0.c[0]); st_dev_u64(op + 1, f1.c[0]);
                st_dev_u64(op + n, f0.c[1]); st_dev_u64(op + n + 1, f1.c[1]);
                st_dev_u64(op + 2 * n, f0.c[2]); st_dev_u64(op + 2 * n + 1, f1.c[2]);
            }
            const Fq3 df = fq3_sub(f1, f0);
            const Fq3Const mc = A.mu_pow[kd];
            const Fq3 mu = fq3_make(mc.c[0], mc.c[1], mc.c[2]);
            const Fq3 f0s = S3<NU>(f0, nu), dfs = S3<NU>(df, nu);
            const Fq3 c0 = fq3_sub(M3<NU>(f0s, f0, nu), f0);
            const Fq3 c3 = M3<NU>(dfs, df, nu);
            const Fq3 t1 = M3<NU>(f0s, df, nu), t2 = M3<NU>(dfs, f0, nu);
            const Fq3 c1 = fq3_sub(fq3_add(fq3_add(t1, t1), t1), df);
            const Fq3 c2 = fq3_add(fq3_add(t2, t2), t2);
            const Fq3 Q[4] = {M3<NU>(c0, mu, nu), M3<NU>(c1, mu, nu), M3<NU>(c2, mu, nu), M3<NU>(c3, mu, nu)};
            Fq3 ea = bq0, es = fq3_sub(bq1, bq0);
#pragma unroll
            for (int X = 0; X < 5; X++) {
                Fq3 v = Q[3];
                for (int e = 2; e >= 0; e--) v = fq3_add(fq3_mul_small(v, X), Q[e]);
                acc[X] = fq3_add(acc[X], M3<NU>(v, ea, nu));
                ea = fq3_add(ea, es);
            }
        }
        if (leader && threadIdx.x == 0) TAIL_STAMP(A.mail, rd, 3);
        // workgroup partial -> row z (columns of this slot); the last workgroup of the round reduces the rows and mails the message
        u64 vv[15];
#pragma unroll
        for (int i = 0; i < 5; i++) { vv[3 * i] = acc[i].c[0]; vv[3 * i + 1] = acc[i].c[1]; vv[3 * i + 2] = acc[i].c[2]; }
        __syncthreads();            // red[] of the previous round is no longer read
        block_sum_store<15>(vv, red);
        __syncthreads();
        if (threadIdx.x < 15) st_dev_u64(A.partial + (size_t)blockIdx.z * 120 + (threadIdx.x / 3) * 24 + 3 * slot + threadIdx.x % 3, red[threadIdx.x]);
        wait_mem();                 // the partial sums (and, in the last round, the tables) have reached memory ...
        __syncthreads();
        if (threadIdx.x == 0) s_flag = __hip_atomic_fetch_add(&A.counters[rd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1 ? 1u : 0u;   // ... before the count
        __syncthreads();
        if (leader && threadIdx.x == 0) TAIL_STAMP(A.mail, rd, 4);
        if (s_flag) {
            if (threadIdx.x == 0) TAIL_STAMP(A.mail, rd, 5);
            // rows are read with memory-side loads (~2 us each): keep eight in flight per thread, two threads per column
            __shared__ u64 s_half[128];
            {
                const u32 col = threadIdx.x & 127, hf = threadIdx.x >> 7, nz = gridDim.z, per_h = (nz + 1) / 2;
                const u32 b0 = hf * per_h, b1 = b0 + per_h < nz ? b0 + per_h : nz;
                u64 sum = 0;
                if (col < 120) {
                    for (u32 b = b0; b < b1; b += 8) {
                        u64 v[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) v[q] = b + q < b1 ? ld_dev_u64(A.partial + (size_t)(b + q) * 120 + col) : 0;
#pragma unroll
                        for (int q = 0; q < 8; q++) sum = fq_add(sum, v[q]);
                    }
                }
                if (hf == 1) s_half[col] = sum;
                __syncthreads();
                if (hf == 0 && col < 120) {
                    const u64 tot = fq_canon(fq_add(sum, s_half[col]));
                    s_half[col] = tot;   // kept for the device transcript
                    __hip_atomic_store((u64 *)&A.mail->msg[rd][col], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            if (A.dev_transcript) {
                // MLSumcheck round on the device sponge (utils/sumcheck.rs:66-76): absorb the message (5 ring elements), r <- get_challenge
                // (squeeze tau words, absorb them back), absorb R::from(r).  Wave 0 of this (last) workgroup; the sponge lives in device
                // memory between rounds because a different workgroup may be last next time.
                __shared__ u64 sp_st[24], sp_tmp[24], sp_c[24];
                __syncthreads();
                if (threadIdx.x < 64) {
                    const int lane = threadIdx.x;
                    if (lane < 24) sp_st[lane] = ld_dev_u64(A.sponge_state + lane);
                    SpongeDev sp;
                    sp.idx = (int)ld_dev_u64(A.sponge_state + 24);
                    sp.squeezing = (int)ld_dev_u64(A.sponge_state + 25);
                    wave_lds_sync();
                    for (int e = 0; e < 5; e++) sponge_absorb_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, s_half + 24 * e, 24);
                    sponge_squeeze_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, sp_c, 3);
                    sponge_absorb_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, sp_c, 3);
                    const u64 c0 = sp_c[0], c1 = sp_c[1], c2 = sp_c[2];
                    wave_lds_sync();
                    if (lane < 24) sp_c[lane] = lane % 3 == 0 ? c0 : (lane % 3 == 1 ? c1 : c2);   // R::from(r): the challenge in every slot
                    wave_lds_sync();
                    sponge_absorb_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, sp_c, 24);
                    if (lane < 24) st_dev_u64(A.sponge_state + lane, sp_st[lane]);
                    if (lane == 0) { st_dev_u64(A.sponge_state + 24, (u64)sp.idx); st_dev_u64(A.sponge_state + 25, (u64)sp.squeezing); }
                    if (lane < 3) {
                        const u64 cv = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
                        __hip_atomic_store((u64 *)&A.mail->chal_out[rd][lane], cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        st_dev_u64(A.dev_chal + (size_t)rd * 4 + lane, cv);
                    }
                    if (rd + 1 == A.rounds) {
                        if (lane < 24) __hip_atomic_store((u64 *)&A.mail->sponge[lane], sp_st[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (lane == 0) {
                            __hip_atomic_store((u64 *)&A.mail->sponge[24], (u64)sp.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store((u64 *)&A.mail->sponge[25], (u64)sp.squeezing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                    wait_mem();
                    if (lane == 0) st_dev_u64(A.dev_chal + (size_t)rd * 4 + 3, (u64)A.epoch);   // releases the waiting workgroups into the next round
                }
            }
            if (threadIdx.x == 0) TAIL_STAMP(A.mail, rd, 6);
            wait_mem();
            __syncthreads();
            if (threadIdx.x == 0) {
                TAIL_STAMP(A.mail, rd, 7);
                __hip_atomic_store(&A.counters[rd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // self-resetting for the next launch
                __hip_atomic_store((u32 *)&A.mail->msg_seq[rd], A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        n_prev = n;
    }
}
// The same for the linearization sumcheck (comb = linearization/utils.rs:90-107): one workgroup per slot owns the t Mz rows of its slot and
// a private copy of the eq table; a work item is a pair.  T = the Mz tables [t][24][n0], E = the eq table [3][n0] of the round before the
// tail; the fully fixed Mz tables (2 entries per row) are left in Tout ([t][24][2], write-through) for u = Mz(r) (k_fix_final).
template <bool NU>
__global__ void __launch_bounds__(256) k_lin_tail(DevCrt t, LinCombDesc desc, LinTailArgs A) {
    const u32 slot = blockIdx.y;
    const u64 nu = t.nu;
    const u32 nblocks = gridDim.y, bid = blockIdx.y, nt = desc.t, npts = A.deg + 1;
    const bool leader = bid == 0;
    __shared__ u64 s_r[3];
    __shared__ u32 s_flag;
    __shared__ u64 red[15];
    const size_t half0 = A.n0 / 2, row_words = 3 * half0, nrow = 1 + nt;   // rows: 0 eq, 1.. the Mz tables of this slot
    const size_t buf_words = (nrow * row_words + 15) & ~(size_t)15;
    u64 *const priv = A.priv + (size_t)bid * 2 * buf_words;
    size_t n_prev = A.n0;
    Fq3 r = fq3_make(A.r_first.c[0], A.r_first.c[1], A.r_first.c[2]);
    for (u32 rd = 0; rd < A.rounds; rd++) {
        if (rd > 0) {
            if (threadIdx.x == 0) {
                u64 rr[3] = {0, 0, 0};
                bool ok = tail_wait_challenge(A.mail, A.dev_chal, rd - 1, A.epoch, leader && !A.dev_transcript, rr);
                s_r[0] = rr[0]; s_r[1] = rr[1]; s_r[2] = rr[2];
                s_flag = ok ? 1u : 0u;
            }
            __syncthreads();
            if (!s_flag) return;
            r = fq3_make(s_r[0], s_r[1], s_r[2]);
            __syncthreads();
        }
        const size_t n = n_prev / 2, pairs = n / 2, ldp = n_prev;
        const bool first = rd == 0, last = rd + 1 == A.rounds;
        const u64 *Pp = priv + (size_t)((rd + 1) & 1) * buf_words;
        u64 *Pn = priv + (size_t)(rd & 1) * buf_words;
        auto src_row = [&](u32 q) -> const u64 * {
            if (!first) return Pp + (size_t)q * 3 * ldp;
            if (q == 0) return A.E;
            return A.T + ((size_t)(q - 1) * 24 + 3 * slot) * ldp;
        };
        auto fix_pair = [&](const u64 *row, size_t p, Fq3 &f0, Fq3 &f1) {
            const u64 *fp = row + 4 * p;
            ulonglong2 a0 = *(const ulonglong2 *)(fp), a1 = *(const ulonglong2 *)(fp + ldp), a2 = *(const ulonglong2 *)(fp + 2 * ldp);
            ulonglong2 b0 = *(const ulonglong2 *)(fp + 2), b1 = *(const ulonglong2 *)(fp + ldp + 2), b2 = *(const ulonglong2 *)(fp + 2 * ldp + 2);
            Fq3 lo = fq3_make(a0.x, a1.x, a2.x), hi = fq3_make(b0.x, b1.x, b2.x);
            f0 = fq3_add(lo, M3<NU>(fq3_sub(fq3_make(a0.y, a1.y, a2.y), lo), r, nu));
            f1 = fq3_add(hi, M3<NU>(fq3_sub(fq3_make(b0.y, b1.y, b2.y), hi), r, nu));
        };
        auto store_pair = [&](u32 q, size_t p, const Fq3 &f0, const Fq3 &f1) {
            u64 *op = Pn + (size_t)q * 3 * n + 2 * p;
            *(ulonglong2 *)(op) = make_ulonglong2(f0.c[0], f1.c[0]);
            *(ulonglong2 *)(op + n) = make_ulonglong2(f0.c[1], f1.c[1]);
            *(ulonglong2 *)(op + 2 * n) = make_ulonglong2(f0.c[2], f1.c[2]);
        };
        Fq3 acc[5];
#pragma unroll
        for (int i = 0; i < 5; i++) acc[i] = fq3_zero();
        for (size_t p = threadIdx.x; p < pairs; p += 256) {
            Fq3 v[4], st[4], ev, e1;
            fix_pair(src_row(0), p, ev, e1);
            store_pair(0, p, ev, e1);
            Fq3 es = fq3_sub(e1, ev);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if ((u32)j < nt) {
                    Fq3 f0, f1;
                    fix_pair(src_row(1 + j), p, f0, f1);
                    store_pair(1 + j, p, f0, f1);
                    if (last) {   // write-through: the rows of the eight workgroups share cache lines in the shared layout (see k_fold_tail)
                        u64 *op = A.Tout + ((size_t)j * 24 + 3 * slot) * n + 2 * p;
                        st_dev_u64(op, f0.c[0]); st_dev_u64(op + 1, f1.c[0]);
                        st_dev_u64(op + n, f0.c[1]); st_dev_u64(op + n + 1, f1.c[1]);
                        st_dev_u64(op + 2 * n, f0.c[2]); st_dev_u64(op + 2 * n + 1, f1.c[2]);
                    }
                    v[j] = f0; st[j] = fq3_sub(f1, f0);
                } else { v[j] = fq3_zero(); st[j] = fq3_zero(); }
            }
#pragma unroll
            for (int X = 0; X < 5; X++) {
                if ((u32)X < npts) {
                    Fq3 res = fq3_zero(), term = fq3_zero();
                    int sgn = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if ((u32)j < nt) {
                            if (desc.first[j]) {
                                if (sgn) res = sgn < 0 ? fq3_sub(res, term) : fq3_add(res, term);
                                u32 i = desc.ms[j];
                                if (desc.c_unit[i]) { term = v[j]; sgn = desc.c_unit[i]; }
                                else { term = M3<NU>(fq3_make(desc.c[i][3 * slot], desc.c[i][3 * slot + 1], desc.c[i][3 * slot + 2]), v[j], nu); sgn = 1; }
                            } else term = M3<NU>(term, v[j], nu);
                        }
                    }
                    if (sgn) res = sgn < 0 ? fq3_sub(res, term) : fq3_add(res, term);
                    const Fq3 gx = M3<NU>(res, ev, nu);      // (no acc[X]: the rolled loop would index the array dynamically -> scratch, as in k_lin_round)
                    if (X == 0) acc[0] = fq3_add(acc[0], gx);
                    else if (X == 1) acc[1] = fq3_add(acc[1], gx);
                    else if (X == 2) acc[2] = fq3_add(acc[2], gx);
                    else if (X == 3) acc[3] = fq3_add(acc[3], gx);
                    else acc[4] = fq3_add(acc[4], gx);
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = fq3_add(v[j], st[j]);
                    ev = fq3_add(ev, es);
                }
            }
        }
        u64 vv[15];
#pragma unroll
        for (int i = 0; i < 5; i++) { vv[3 * i] = acc[i].c[0]; vv[3 * i + 1] = acc[i].c[1]; vv[3 * i + 2] = acc[i].c[2]; }
        __syncthreads();
        block_sum_store<15>(vv, red);
        __syncthreads();
        if (threadIdx.x < 15) st_dev_u64(A.partial + (threadIdx.x / 3) * 24 + 3 * slot + threadIdx.x % 3, red[threadIdx.x]);
        wait_mem();
        __syncthreads();
        if (threadIdx.x == 0) s_flag = __hip_atomic_fetch_add(&A.counters[rd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1 ? 1u : 0u;
        __syncthreads();
        if (s_flag) {
            __shared__ u64 s_msg[128];
            if (threadIdx.x < 120) {
                const u64 tot = threadIdx.x < npts * 24 ? fq_canon(ld_dev_u64(A.partial + threadIdx.x)) : 0;
                s_msg[threadIdx.x] = tot;
                __hip_atomic_store((u64 *)&A.mail->msg[rd][threadIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (A.dev_transcript) {
                __shared__ u64 sp_st[24], sp_tmp[24], sp_c[24];
                __syncthreads();
                if (threadIdx.x < 64) {
                    const int lane = threadIdx.x;
                    if (lane < 24) sp_st[lane] = ld_dev_u64(A.sponge_state + lane);
                    SpongeDev sp;
                    sp.idx = (int)ld_dev_u64(A.sponge_state + 24);
                    sp.squeezing = (int)ld_dev_u64(A.sponge_state + 25);
                    wave_lds_sync();
                    for (u32 e = 0; e < npts; e++) sponge_absorb_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, s_msg + 24 * e, 24);
                    sponge_squeeze_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, sp_c, 3);
                    sponge_absorb_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, sp_c, 3);
                    const u64 c0 = sp_c[0], c1 = sp_c[1], c2 = sp_c[2];
                    wave_lds_sync();
                    if (lane < 24) sp_c[lane] = lane % 3 == 0 ? c0 : (lane % 3 == 1 ? c1 : c2);
                    wave_lds_sync();
                    sponge_absorb_wave(sp, sp_st, sp_tmp, A.pos_ark, A.pos_mds, sp_c, 24);
                    if (lane < 24) st_dev_u64(A.sponge_state + lane, sp_st[lane]);
                    if (lane == 0) { st_dev_u64(A.sponge_state + 24, (u64)sp.idx); st_dev_u64(A.sponge_state + 25, (u64)sp.squeezing); }
                    if (lane < 3) {
                        const u64 cv = lane == 0 ? c0 : (lane == 1 ? c1 : c2);
                        __hip_atomic_store((u64 *)&A.mail->chal_out[rd][lane], cv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        st_dev_u64(A.dev_chal + (size_t)rd * 4 + lane, cv);
                    }
                    if (rd + 1 == A.rounds) {
                        if (lane < 24) __hip_atomic_store((u64 *)&A.mail->sponge[lane], sp_st[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (lane == 0) {
                            __hip_atomic_store((u64 *)&A.mail->sponge[24], (u64)sp.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store((u64 *)&A.mail->sponge[25], (u64)sp.squeezing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
                    wait_mem();
                    if (lane == 0) st_dev_u64(A.dev_chal + (size_t)rd * 4 + 3, (u64)A.epoch);
                }
            }
            wait_mem();
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_store(&A.counters[rd], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store((u32 *)&A.mail->msg_seq[rd], A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        n_prev = n;
    }
}
size_t lin_tail_priv_words(size_t n0, u32 t) { return (size_t)8 * 2 * ((((1 + (size_t)t) * 3 * (n0 / 2)) + 15) & ~(size_t)15) + 16; }
// eight workgroups (one per slot); returns 0 when the tail cannot run (the caller keeps per-round launches)
u32 launch_lin_tail(const DevCrt &t, const LinCombDesc &desc, const LinTailArgs &A, hipStream_t s) {
    if (A.n0 < 4 || A.rounds < 1 || A.rounds > TAIL_MAX_ROUNDS || A.deg + 1 > 5) return 0;
    if (t.nu2p40) hipLaunchKernelGGL((k_lin_tail<true>), dim3(1, 8), dim3(256), 0, s, t, desc, A);
    else hipLaunchKernelGGL((k_lin_tail<false>), dim3(1, 8), dim3(256), 0, s, t, desc, A);
    return 8;
}

// private working sets: per workgroup two buffers of (5 + tables per workgroup) rows x 3 planes x n0/2 entries; sized for the smallest
// chunk count the launcher may pick (8 workgroups per chunk), which needs the most rows
size_t fold_tail_eqpriv_words(size_t n0, u32 K) {
    size_t worst = 0;
    static const u32 cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 96};
    for (u32 cc : cand) {
        if (8 * cc > FOLD_TAIL_MAX_BLOCKS) break;
        size_t per = (2 * (size_t)K * 3 + cc - 1) / cc, buf = (((5 + per) * 3 * (n0 / 2)) + 15) & ~(size_t)15;
        size_t tot = (size_t)8 * cc * 2 * buf;
        if (tot > worst) worst = tot;
    }
    return worst + 16;
}
// grid = (1, 8 slots, table chunks); returns the number of workgroups (0: the tail cannot run here, the caller falls back to
// per-round launches).  A.eqpriv must hold fold_tail_eqpriv_words(n0, K) words, 128-byte aligned; the fully fixed tables
// (2 entries per row) are left in A.F[1] whatever the number of rounds.
u32 launch_fold_tail(const DevCrt &t, const FoldTailArgs &A, int num_cus, hipStream_t s) {
    static int occ_nu = -1, occ_g = -1;
    if (occ_nu < 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_nu, (const void *)k_fold_tail<true>, 256, 0) != hipSuccess) occ_nu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_g, (const void *)k_fold_tail<false>, 256, 0) != hipSuccess) occ_g = 0;
    }
    const int occ = t.nu2p40 ? occ_nu : occ_g;
    if (occ < 1 || num_cus < 1 || A.n0 < 4 || A.rounds < 1 || A.rounds > TAIL_MAX_ROUNDS) return 0;
    size_t max_blocks = (size_t)occ * (size_t)num_cus / 2;   // half of what could be resident: other streams keep running
    if (max_blocks > FOLD_TAIL_MAX_BLOCKS) max_blocks = FOLD_TAIL_MAX_BLOCKS;
    if (max_blocks < 8) return 0;
    const u32 nkd = 2 * A.K * 3;
    u32 chunks = 1;
    static const u32 cand[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 96};
    for (u32 cc : cand) {
        if (cc > nkd || (size_t)8 * cc > max_blocks) break;
        chunks = cc;
    }
    if (t.nu2p40) hipLaunchKernelGGL((k_fold_tail<true>), dim3(1, 8, chunks), dim3(256), 0, s, t, A);
    else hipLaunchKernelGGL((k_fold_tail<false>), dim3(1, 8, chunks), dim3(256), 0, s, t, A);
    return 8 * chunks;
}

// rounds 3 and 4 straight from the coefficient planes through the 81-entry digit look-up table (lut_dev: [81][3], see FoldSrc):
// round 3 touches no table at all, round 4 fixes with r and writes the first materialised tables Fout [2K*3][24][ldout]
void launch_fold_round_lut(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                           const u64 *lut_dev, u32 K, const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s) {
    FoldSrc src = {};
    src.planesL = planesL; src.planesR = planesR; src.n_planes = n_planes; src.lut = lut_dev;
    launch_fold_round_mode<3>(t, a, nullptr, 0, K, mu_pow_dev, src, partial, out, s);
}
// round 3 with the per-table mu products (mutab_dev: 3 * 2K*3 * 81 * 4 words, filled here from lut_dev and mu_pow_dev); NU = 2^40 only
void launch_fold_round_lut_mu(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                              const u64 *lut_dev, u64 *mutab_dev, u32 K, const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s) {
    LF_LAUNCH(k_fold_mutab, t.nu2p40, dim3(2 * K * 3), dim3(128), s, t, lut_dev, mu_pow_dev, 2 * K * 3, mutab_dev);
    FoldSrc src = {};
    src.planesL = planesL; src.planesR = planesR; src.n_planes = n_planes; src.lut = lut_dev; src.mutab = mutab_dev;
    launch_fold_round_mode<5>(t, a, nullptr, 0, K, mu_pow_dev, src, partial, out, s);
}
void launch_fold_round_lut_fix(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                               const u64 *lut_dev, Fq3Const r, u64 *Fout, size_t ldout, u32 K, const Fq3Const *mu_pow_dev, u64 *partial,
                               u64 *out, hipStream_t s) {
    FoldSrc src = {};
    src.planesL = planesL; src.planesR = planesR; src.n_planes = n_planes; src.lut = lut_dev;
    src.out = Fout; src.ldo = ldout; src.r = r;
    launch_fold_round_mode<4>(t, a, nullptr, 0, K, mu_pow_dev, src, partial, out, s);
}
// the same through the product-free tables of mode 6 (sq_dev 6561*4 words, mt_dev 2K*3*2*81*4 words, filled here); NU = 2^40 only
void launch_fold_round_lut_fix_tab(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                                   const u64 *lut_dev, Fq3Const r, u64 *sq_dev, u64 *mt_dev, u64 *Fout, size_t ldout, u32 K, const Fq3Const *mu_pow_dev,
                                   u64 *partial, u64 *out, hipStream_t s, const u64 *E, size_t ldE) {
    const u32 nkd = 2 * K * 3;
    LF_LAUNCH(k_fold_r4tab, t.nu2p40, dim3((6561 + nkd * 162 + 255) / 256), dim3(256), s, t, lut_dev, r, mu_pow_dev, nkd, sq_dev, mt_dev);
    FoldSrc src = {};
    src.planesL = planesL; src.planesR = planesR; src.n_planes = n_planes; src.lut = lut_dev;
    src.out = Fout; src.ldo = ldout; src.r = r; src.sq4 = sq_dev; src.mt4 = mt_dev; src.E = E; src.ldE = ldE;
    launch_fold_round_mode<6>(t, a, nullptr, 0, K, mu_pow_dev, src, partial, out, s);
}
// round 5 from the planes (mode 7): xx_dev / yy_dev 6561*4 words each, mt_dev 2K*3*4*81*4 words, filled here; r3 / r4 = the challenges of rounds 3 / 4
void launch_fold_round_lut_fix5(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                                const u64 *lut_dev, Fq3Const r3, Fq3Const r4, u64 *xx_dev, u64 *yy_dev, u64 *mt_dev, u64 *Fout, size_t ldout, u32 K,
                                const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s, const u64 *E, size_t ldE) {
    const u32 nkd = 2 * K * 3;
    LF_LAUNCH(k_fold_r5tab, t.nu2p40, dim3((2 * 6561 + nkd * 324 + 255) / 256), dim3(256), s, t, lut_dev, r3, r4, mu_pow_dev, nkd, xx_dev, yy_dev, mt_dev);
    FoldSrc src = {};
    src.planesL = planesL; src.planesR = planesR; src.n_planes = n_planes; src.lut = lut_dev;
    src.out = Fout; src.ldo = ldout; src.r = r4; src.r_prev = r3; src.xx5 = xx_dev; src.yy5 = yy_dev; src.mt5 = mt_dev; src.E = E; src.ldE = ldE;
    launch_fold_round_mode<7>(t, a, nullptr, 0, K, mu_pow_dev, src, partial, out, s);
}
// round message + fused fix_variables: Fprev [2K*3][24][ldprev] (entries 4p..4p+3 of every pair p) -> Fout [..][ldout]
void launch_fold_round_fix(const DevCrt &t, const FoldRoundArgs &a, const u64 *Fprev, size_t ldprev, Fq3Const r, u64 *Fout, size_t ldout, u32 K,
                           const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s, const u64 *E, size_t ldE) {
    FoldSrc src = {};
    src.out = Fout; src.ldo = ldout; src.r = r; src.E = E; src.ldE = ldE;
    launch_fold_round_mode<1>(t, a, Fprev, ldprev, K, mu_pow_dev, src, partial, out, s);
}

// ---------------------------------------------------------------------------------------------------------
// compute_f_0 (folding.rs:258-268) in the coefficient domain: ICRT(sum_i rho_i (.) f_i) = sum_i rho_i * f_i mod
// Phi_72 exactly, with rho_i in [-32,32)^24 and f_i the bit-planes -> plain int32 convolutions.
// Nibble tables.  For one side, sum_k rho_k[a] * digit_k(v_c) = sign(v_c) * sum_nibbles R[nibble][value][a] with
// R[q][val][a] = sum_{b<4} bit_b(val) rho_{4q+b}[a]: four look-ups of a 24-vector and 24 additions per coefficient c replace the
// 16 x 24 multiply-adds over the bit-planes.  The tables (both signs, both sides: 2*2*4*16*24 int32 = 24 KB) are built in LDS per block.
// Sliding window: coefficient c only touches positions c..c+23, so with both sides handled per group of 8 coefficients the positions
// C0..C0+7 are final after the group; they are stored (before the X^24 wrap) and leave the registers -- 31 live accumulators, not 47.
template <int C0, int NQ>
__device__ __forceinline__ void fw_group8(int32_t (&win)[31], const int32_t *pL, const int32_t *pR, size_t n, size_t j,
                                          const int32_t (*R)[2][NQ][16][28], int32_t *out) {
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const int32_t *pl = side ? pR : pL;
        int32_t vv[8];
#pragma unroll
        for (int i = 0; i < 8; i++) vv[i] = pl[(size_t)(C0 + i) * n + j];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int32_t v = vv[i];
            u32 mg = (u32)(v < 0 ? -v : v), sg = v < 0;
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                const int4 *t = (const int4 *)R[side][sg][q][(mg >> (4 * q)) & 15];
#pragma unroll
                for (int w = 0; w < 6; w++) {
                    int4 x = t[w];
                    win[i + 4 * w] += x.x; win[i + 4 * w + 1] += x.y; win[i + 4 * w + 2] += x.z; win[i + 4 * w + 3] += x.w;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[(size_t)(C0 + i) * n + j] = win[i];
#pragma unroll
    for (int i = 0; i < 23; i++) win[i] = win[i + 8];
#pragma unroll
    for (int i = 23; i < 31; i++) win[i] = 0;
}
template <int NQ>   // nibbles of |v|: 4 for K <= 16 bit-planes, 8 for K <= 32
__global__ void __launch_bounds__(256) k_fold_witness(const int32_t *planesL, const int32_t *planesR, size_t n, u32 K, const int8_t *rho,
                                                      int32_t *out) {
    __shared__ __align__(16) int32_t R[2][2][NQ][16][28];   // [side][sign][nibble][value][a]; rows padded to 28 words (bank spread)
    for (u32 idx = threadIdx.x; idx < 2 * NQ * 16 * 24; idx += 256) {
        u32 a = idx % 24, val = (idx / 24) % 16, q = (idx / (24 * 16)) % NQ, side = idx / (24 * 16 * NQ);
        int sum = 0;
#pragma unroll
        for (u32 b = 0; b < 4; b++)
            if (4 * q + b < K && ((val >> b) & 1)) sum += rho[(size_t)(side * K + 4 * q + b) * 24 + a];
        R[side][0][q][val][a] = sum;
        R[side][1][q][val][a] = -sum;
    }
    __syncthreads();
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int32_t win[31];
#pragma unroll
    for (int i = 0; i < 31; i++) win[i] = 0;
    fw_group8<0, NQ>(win, planesL, planesR, n, j, R, out);
    fw_group8<8, NQ>(win, planesL, planesR, n, j, R, out);
    fw_group8<16, NQ>(win, planesL, planesR, n, j, R, out);
    // win[i] = position 24 + i;  X^24 = X^12 - 1, applied top down (positions >= 36 land on positions >= 24 first)
    int32_t delta[24];
#pragma unroll
    for (int i = 0; i < 24; i++) delta[i] = 0;
#pragma unroll
    for (int i = 22; i >= 12; i--) { win[i - 12] += win[i]; delta[i] -= win[i]; }
#pragma unroll
    for (int i = 11; i >= 0; i--) { delta[12 + i] += win[i]; delta[i] -= win[i]; }
#pragma unroll
    for (int c = 0; c < 24; c++) out[(size_t)c * n + j] += delta[c];
}
void launch_fold_witness(const int32_t *planesL, const int32_t *planesR, size_t n, u32 K, const int8_t *rho_dev, int32_t *out, hipStream_t s) {
    if (K <= 16) hipLaunchKernelGGL(k_fold_witness<4>, dim3(cdiv(n, 256)), dim3(256), 0, s, planesL, planesR, n, K, rho_dev, out);
    else hipLaunchKernelGGL(k_fold_witness<8>, dim3(cdiv(n, 256)), dim3(256), 0, s, planesL, planesR, n, K, rho_dev, out);
}

}  // namespace lf

// lf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the LatticeFold prover hot path: layouts, CRT / ICRT, decomposition, witness plumbing, eq
// tables, SpMV, batched inner products, fix_variables, compute_f_0.  The sumcheck round kernels live in lf_rounds.hip; both include lf_kernels_dev.cuh.
//
// Every kernel works on plane-major (SoA) tables so that lane <-> consecutive element index gives coalesced
// 8/16-byte accesses; cross-lane reductions use wave64 shuffles + one LDS hop per 256-thread block; the Ajtai
// mat-vec stages (A, witness) tiles through LDS.  No MFMA: the arithmetic is 64-bit modular (four
// v_mad_u64_u32 per product, see lf_field.cuh).  Reference semantics each kernel replaces are cited inline.
#include "lf_kernels.h"

#include <stdlib.h>

#include "lf_kernels_dev.cuh"

namespace lf {


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


__device__ __forceinline__ u64 splitmix_fq(u64 seed, u64 index);


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
// thread = (element i, residue class u of the coefficient index), coefficients c = u + 3 q: the 8 x 4 plane entries it needs are loaded
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
    if (kg < 0) kg = 8;   // planes per thread: 8 halves the HBM re-reads of the 4-plane version at the same speed (16 spills into the latency chain: slower)
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

// bb_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the BabyBearRingNTT backend (BASELINE configs[2]:
// "alt-prime 31-bit Montgomery path").  Same data-flow as the Goldilocks kernels (lf_kernels.hip): plane-major (SoA)
// tables, lane <-> consecutive element index, cross-lane reductions by wave64 shuffles + one LDS hop, the Ajtai mat-vec
// staged through LDS.  Arithmetic: centred Montgomery int32 words; an F_{p^9} product is 81 v_mad_i64_i32 into nine
// signed 64-bit column sums (9 * H^2 < 2^63) + nine Montgomery reductions (bb_field.cuh).  No MFMA.
#include "bb_kernels.h"

#include <stdlib.h>

#include "bb_kernels_dev.cuh"

namespace lfbb {


DevBb make_dev_bb(const BbTables &T) {
    DevBb d;
    d.nu = from_canon(T.nu);
    d.w4 = from_canon(T.w4); d.w2 = from_canon(T.w2); d.w10 = from_canon(T.w10); d.w1 = from_canon(T.w1);
    d.w7 = from_canon(T.w7); d.w5 = from_canon(T.w5); d.w11 = from_canon(T.w11);
    for (int p = 0; p < 8; p++) {
        d.slot_of_pos[p] = T.slot_of_pos[p];
        for (int r = 0; r < TAU; r++) { d.pos[r][p] = T.pos[r][p]; d.tw[r][p] = from_canon(T.tw[r][p]); }
    }
    return d;
}
E9C e9c_from_h9(const H9 &h) { E9C r; for (int i = 0; i < TAU; i++) r.c[i] = from_canon(h.c[i]); return r; }
E9PreC e9pre_from_h9(const H9 &h, u64 nu) {
    E9PreC r;
    for (int i = 0; i < TAU; i++) { r.v[i] = from_canon(h.c[i]); r.vn[i] = from_canon(hmul(h.c[i] % BB_P, nu)); }
    return r;
}
// out[i] = canonical( sum_b partial[b*nv + i] ); values are Montgomery words
__global__ void __launch_bounds__(256) k_reduce_rows(const i64 *partial, u32 nblocks, u32 nv, u64 *out) {
    u32 i = blockIdx.x;
    i64 acc[1] = {0};
    for (u32 b = threadIdx.x; b < nblocks; b += 256) acc[0] += partial[(size_t)b * nv + i] % (i64)BB_P;
    __shared__ i64 res[1];
    block_sum_store<1>(acc, res);
    __syncthreads();
    if (threadIdx.x == 0) out[i] = to_canon(fred(res[0]));
}
// v = sum_k 2^k v_s[k] (canonical words): the MLE evaluation of the witness coefficients from the evaluations of their K binary digit planes
__global__ void __launch_bounds__(256) k_vs_combine(const u64 *vs, u32 K, u32 nv, u64 *v) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    u64 acc = 0, pw = 1;
    for (u32 k = 0; k < K; k++) {
        acc = (acc + vs[(size_t)k * nv + i] % BB_P * pw) % BB_P;
        pw = pw * 2 % BB_P;
    }
    v[i] = acc;
}
void launch_vs_combine(const u64 *vs, u32 K, u32 nv, u64 *v, hipStream_t s) { hipLaunchKernelGGL(k_vs_combine, dim3((nv + 255) / 256), dim3(256), 0, s, vs, K, nv, v); }
void launch_reduce_rows(const i64 *partial, u32 nblocks, u32 nv, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce_rows, dim3(nv), dim3(256), 0, s, partial, nblocks, nv, out);
}
size_t red_partial_words(u32 nv) { return (size_t)RED_BLOCKS * nv; }

// ---------------------------------------------------------------------------------------------------------
// layout + Montgomery conversion at the ABI
__global__ void __launch_bounds__(256) k_aos_to_soa(const u64 *aos, fe *soa, size_t n) {
    __shared__ fe tile[64][RE + 1];
    size_t base = (size_t)blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * RE; idx += 256) {
        size_t e = base + idx / RE;
        tile[idx / RE][idx % RE] = e < n ? from_canon(aos[e * RE + idx % RE]) : 0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * RE; idx += 256) {
        int w = idx / 64, j = idx % 64;
        if (base + j < n) soa[(size_t)w * n + base + j] = tile[j][w];
    }
}
__global__ void __launch_bounds__(256) k_soa_to_aos(const fe *soa, u64 *aos, size_t n) {
    __shared__ fe tile[64][RE + 1];
    size_t base = (size_t)blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * RE; idx += 256) {
        int w = idx / 64, j = idx % 64;
        tile[j][w] = base + j < n ? soa[(size_t)w * n + base + j] : 0;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * RE; idx += 256) {
        size_t e = base + idx / RE;
        if (e < n) aos[e * RE + idx % RE] = to_canon(tile[idx / RE][idx % RE]);
    }
}
void launch_aos_to_soa(const u64 *aos, fe *soa, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_aos_to_soa, dim3(cdiv(n, 64)), dim3(256), 0, s, aos, soa, n);
}
void launch_soa_to_aos(const fe *soa, u64 *aos, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_soa_to_aos, dim3(cdiv(n, 64)), dim3(256), 0, s, soa, aos, n);
}
// workload.py splitmix_fq(ring="babybear"): top 32 bits of SplitMix64 word (index+1), mod p
__device__ __forceinline__ u64 splitmix_bb(u64 seed, u64 index) {
    u64 z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (z >> 32) % BB_P;
}
__global__ void __launch_bounds__(256) k_fill_ajtai(fe *A, u32 kappa, size_t n, size_t n_total, size_t col0, u64 seed) {
    size_t total = (size_t)kappa * RE * n;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < total; i += st) {
        size_t j = i % n, w = (i / n) % RE, row = i / (RE * n);
        A[i] = from_canon(splitmix_bb(seed, (row * n_total + col0 + j) * RE + w));
    }
}
void launch_fill_ajtai(fe *A, u32 kappa, size_t n, size_t n_total, size_t col0, u64 seed, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_ajtai, dim3(4096), dim3(256), 0, s, A, kappa, n, n_total, col0, seed);
}
// arithmetic self-test: in[i] = (a[9], b[9]) canonical; out[i] = (a*b [9], a0+b0, a0-b0, a0*b0) canonical.
// The host recomputes with plain % arithmetic (bb_host.cpp) and compares.
__global__ void __launch_bounds__(256) k_selftest(const u64 *in, u64 *out, u32 n, fe nu) {
    u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    E9 a, b;
    for (int c = 0; c < TAU; c++) { a.c[c] = from_canon(in[(size_t)i * 18 + c]); b.c[c] = from_canon(in[(size_t)i * 18 + 9 + c]); }
    E9 p = e9_mul(a, b, nu);
    for (int c = 0; c < TAU; c++) out[(size_t)i * 12 + c] = to_canon(p.c[c]);
    out[(size_t)i * 12 + 9] = to_canon(fadd(a.c[0], b.c[0]));
    out[(size_t)i * 12 + 10] = to_canon(fsub(a.c[0], b.c[0]));
    out[(size_t)i * 12 + 11] = to_canon(fmul(a.c[0], b.c[0]));
}
void launch_selftest(const u64 *in, u64 *out, u32 n, fe nu, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest, dim3(cdiv(n, 256)), dim3(256), 0, s, in, out, n, nu);
}

// ---------------------------------------------------------------------------------------------------------
// CRT: a(X) = sum_{r<9} X^r A_r(X^9); A_r is evaluated at the 8 primitive 24th roots by three radix-2 layers over
// U^8 - U^4 + 1 = (U^4 - w^4)(U^4 - w^20); the per-slot monomial twist X^r -> tw[r] Y^pos[r] maps F_p[X]/(X^9 - zeta_k)
// onto F_p[Y]/(Y^9 - nu).  (stark-rings CRT; call sites arith.rs:238,327.)
__device__ __forceinline__ void crt8(const fe x[8], fe o[8], const DevBb &t) {
    fe lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        fe tt = fmul(t.w4, x[i + 4]);
        lo[i] = fadd(x[i], tt);
        hi[i] = fsub(fadd(x[i], x[i + 4]), tt);
    }
    fe l0[2], l1[2], h0[2], h1[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        fe tt = fmul(t.w2, lo[i + 2]);
        l0[i] = fadd(lo[i], tt); l1[i] = fsub(lo[i], tt);
        fe uu = fmul(t.w10, hi[i + 2]);
        h0[i] = fadd(hi[i], uu); h1[i] = fsub(hi[i], uu);
    }
    fe a = fmul(t.w1, l0[1]);  o[0] = fadd(l0[0], a); o[1] = fsub(l0[0], a);
    fe b = fmul(t.w7, l1[1]);  o[2] = fadd(l1[0], b); o[3] = fsub(l1[0], b);
    fe c = fmul(t.w5, h0[1]);  o[4] = fadd(h0[0], c); o[5] = fsub(h0[0], c);
    fe d = fmul(t.w11, h1[1]); o[6] = fadd(h1[0], d); o[7] = fsub(h1[0], d);
}
// coefficients a[72] (Montgomery) -> the 72 NTT words of element j of plane table `out`
__device__ __forceinline__ void crt_store(const fe a[RE], fe *out, size_t ld, size_t j, const DevBb &t) {
#pragma unroll
    for (int r = 0; r < TAU; r++) {
        fe x[8], A[8];
#pragma unroll
        for (int v = 0; v < 8; v++) x[v] = a[r + TAU * v];
        crt8(x, A, t);
#pragma unroll
        for (int p = 0; p < 8; p++) {
            int plane = TAU * t.slot_of_pos[p] + t.pos[r][p];
            out[(size_t)plane * ld + j] = r == 0 ? A[p] : fmul(t.tw[r][p], A[p]);
        }
    }
}
__global__ void __launch_bounds__(256) k_crt_fwd(DevBb t, const fe *coef, fe *ntt, size_t n) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    fe a[RE];
#pragma unroll
    for (int c = 0; c < RE; c++) a[c] = coef[(size_t)c * n + j];
    crt_store(a, ntt, n, j, t);
}
void launch_crt_fwd(const DevBb &t, const fe *coef, fe *ntt, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_crt_fwd, dim3(cdiv(n, 256)), dim3(256), 0, s, t, coef, ntt, n);
}
// ICRT as the dense 72x72 F_p matrix (rare: ingest / export only)
__global__ void __launch_bounds__(256) k_icrt_dense(const fe *mat, const fe *ntt, fe *coef, size_t n) {
    __shared__ fe M[RE * RE];
    for (int i = threadIdx.x; i < RE * RE; i += 256) M[i] = mat[i];
    __syncthreads();
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    fe x[RE];
#pragma unroll
    for (int c = 0; c < RE; c++) x[c] = ntt[(size_t)c * n + j];
    for (int i = 0; i < RE; i++) {
        i64 tot = 0;
#pragma unroll
        for (int g = 0; g < 8; g++) {   // 9 products per exact 64-bit group
            i64 acc = 0;
#pragma unroll
            for (int c = 0; c < 9; c++) acc += (i64)M[i * RE + 9 * g + c] * (i64)x[9 * g + c];
            tot += mred(acc);
        }
        coef[(size_t)i * n + j] = fred(tot);
    }
}
void launch_icrt_dense(const fe *mat, const fe *ntt, fe *coef, size_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_icrt_dense, dim3(cdiv(n, 256)), dim3(256), 0, s, mat, ntt, coef, n);
}
// The digit pass of a general commitment (lf_ajtai_i8g.hip) straight from the NTT form: f [72][ld] -> coefficients (inverse CRT map through LDS) -> centred
// residues -> NP balanced base-128 digit words per (coefficient, 8 columns): pre [NP][72][ldw], byte = 64 + digit.  Block = 32 columns (4 tiles).
// The inverse map is data: `mat` is its dense 72 x 72 matrix; when every row has at most 8 non-zero entries (the shipped tables: one per slot) the
// compressed rows sp_val / sp_col [72][8] make an output 8 products instead of 72.
__global__ void __launch_bounds__(256) k_i8g_cut_ntt(const fe *mat, const fe *sp_val, const u32 *sp_col, const fe *ntt, size_t ld, size_t n, u32 NP, size_t ntiles,
                                                     unsigned long long *pre, size_t ldw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cut[];
    fe *M = (fe *)smem_cut;                                      // dense: [72][72]; compressed: [72][8]
    u32 *MC = (u32 *)(M + (sp_val ? RE * 8 : RE * RE));          // compressed: [72][8] columns
    fe *X = (fe *)(MC + (sp_val ? RE * 8 : 0));                  // [73][32] (row 72: zeros, the operand of a missing entry)
    int32_t *Cf = (int32_t *)(X + 73 * 32);                      // [72][33] centred residues
    if (sp_val) {
        for (int t = threadIdx.x; t < RE * 8; t += 256) { M[t] = sp_val[t]; MC[t] = sp_col[t] < (u32)RE ? sp_col[t] : RE; }
    } else
        for (int t = threadIdx.x; t < RE * RE; t += 256) M[t] = mat[t];
    if (threadIdx.x < 32) X[RE * 32 + threadIdx.x] = 0;
    const size_t j0 = (size_t)blockIdx.x * 32;
    for (int t = threadIdx.x; t < RE * 32; t += 256) {
        const u32 c = t >> 5, jj = t & 31;
        X[c * 32 + jj] = j0 + jj < n ? ntt[(size_t)c * ld + j0 + jj] : 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < RE * 32; t += 256) {
        const u32 r = t >> 5, jj = t & 31;
        fe v;
        if (sp_val) {
            i64 acc = 0;                                         // eight products: exact in 64 bits
#pragma unroll
            for (int q = 0; q < 8; q++) acc += (i64)M[r * 8 + q] * (i64)X[MC[r * 8 + q] * 32 + jj];
            v = mred(acc);
        } else {
            i64 tot = 0;
            for (int g = 0; g < 8; g++) {
                i64 acc = 0;
#pragma unroll
                for (int c = 0; c < 9; c++) acc += (i64)M[r * RE + 9 * g + c] * (i64)X[(9 * g + c) * 32 + jj];
                tot += mred(acc);
            }
            v = fred(tot);
        }
        const u32 cv = to_canon(v);
        Cf[r * 33 + jj] = cv > (BB_P - 1) / 2 ? (int32_t)cv - (int32_t)BB_P : (int32_t)cv;
    }
    __syncthreads();
    for (u32 item = threadIdx.x; item < (u32)RE * 4; item += 256) {
        const u32 c = item >> 2, tl = item & 3;
        const size_t T = (size_t)blockIdx.x * 4 + tl;
        if (T >= ntiles) continue;
        int32_t x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = Cf[c * 33 + tl * 8 + q];
        for (u32 k = 0; k < NP; k++) {
            unsigned long long w = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int32_t t = x[q] + 64;
                w |= (unsigned long long)(t & 127) << (8 * q);
                x[q] = t >> 7;
            }
            pre[((size_t)k * RE + c) * ldw + T] = w;
        }
    }
}
void launch_i8g_cut_ntt(const fe *icrt_mat, const fe *sp_val, const u32 *sp_col, const fe *ntt, size_t ld, size_t n, u32 NP, unsigned long long *pre, size_t ldw,
                        hipStream_t s) {
    const size_t ntiles = (n + 7) / 8;
    const size_t lds = (sp_val ? (size_t)RE * 8 * 8 : (size_t)RE * RE * 4) + 73 * 32 * 4 + 72 * 33 * 4;
    if (n) hipLaunchKernelGGL(k_i8g_cut_ntt, dim3((unsigned)cdiv(ntiles, 4)), dim3(256), lds, s, icrt_mat, sp_val, sp_col, ntt, ld, n, NP, ntiles, pre, ldw);
}

// ---------------------------------------------------------------------------------------------------------
// balanced decomposition on coefficient tables, power-of-two base (stark_rings::balanced_decomposition; call sites
// arith.rs:235, decomposition/utils.rs:23-31,48).  Sign-magnitude, |digit| <= base/2, ties kept.
__global__ void __launch_bounds__(256) k_decompose(const fe *coef, size_t n, u32 log_base, u32 digits, int layout, fe *out, int mode) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * RE) return;
    size_t c = idx / n, i = idx % n;
    u32 v = to_canon(coef[idx]);
    bool neg = v > (BB_P - 1) / 2;
    u64 mag = neg ? BB_P - v : v;
    u64 half = 1ULL << (log_base - 1), mask = (1ULL << log_base) - 1;
    size_t n_out = layout == 0 ? n * digits : n;
    int64_t cur = neg ? -(int64_t)mag : (int64_t)mag;
    for (u32 k = 0; k < digits; k++) {
        int64_t dg;
        if (mode == 1 && log_base > 1) {   // floor rule (lf_set_digit_mode 1): digits in [-base/2, base/2)
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
        size_t o = layout == 0 ? (c * n_out + i * digits + k) : ((size_t)k * RE * n + c * n + i);
        out[o] = from_small((int32_t)dg);
    }
}
void launch_decompose(const fe *coef, size_t n, u64 base, u32 digits, int layout, fe *out, hipStream_t s, int mode) {
    u32 lb = 0;
    while ((1ULL << lb) < base) lb++;
    if (n) hipLaunchKernelGGL(k_decompose, dim3(cdiv(n * RE, 256)), dim3(256), 0, s, coef, n, lb, digits, layout, out, mode);
}
__global__ void __launch_bounds__(256) k_recompose(const fe *in, size_t n_out, fe base, u32 digits, fe *out) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out * RE) return;
    size_t w = idx / n_out, i = idx % n_out;
    size_t n_in = n_out * digits;
    fe acc = 0, pw = BB_ONE;
    for (u32 j = 0; j < digits; j++) {
        acc = fadd(acc, fmul(in[w * n_in + i * digits + j], pw));
        pw = fmul(pw, base);
    }
    out[idx] = acc;
}
void launch_recompose(const fe *in, size_t n_out, u64 base, u32 digits, fe *out, hipStream_t s) {
    if (n_out) hipLaunchKernelGGL(k_recompose, dim3(cdiv(n_out * RE, 256)), dim3(256), 0, s, in, n_out, from_canon(base % BB_P), digits, out);
}
__global__ void __launch_bounds__(256) k_coef_to_i32(const fe *coef, int32_t *planes, size_t total, u32 bound, int *viol) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    int bad = 0;
    for (; i < total; i += st) {
        u32 v = to_canon(coef[i]);
        bool neg = v > (BB_P - 1) / 2;
        u32 mag = neg ? BB_P - v : v;
        if (mag > bound) { bad = 1; mag = 0; }
        planes[i] = neg ? -(int32_t)mag : (int32_t)mag;
    }
    if (bad) atomicOr(viol, 1);
}
void launch_coef_to_i32(const fe *coef, int32_t *planes, size_t n, u32 bound, int *viol, hipStream_t s) {
    hipLaunchKernelGGL(k_coef_to_i32, dim3(grid_for(n * RE, 4096)), dim3(256), 0, s, coef, planes, n * RE, bound, viol);
}
__global__ void __launch_bounds__(256) k_i32_to_coef(const int32_t *planes, fe *coef, size_t total) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i < total; i += st) coef[i] = from_small(planes[i]);
}
void launch_i32_to_coef(const int32_t *planes, fe *coef, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_i32_to_coef, dim3(grid_for(n * RE, 4096)), dim3(256), 0, s, planes, coef, n * RE);
}
__global__ void __launch_bounds__(256) k_linf(const fe *coef, size_t total, unsigned long long *out_max) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    u64 mx = 0;
    for (; i < total; i += st) {
        u32 v = to_canon(coef[i]);
        u64 mag = v > (BB_P - 1) / 2 ? BB_P - v : v;
        mx = mag > mx ? mag : mx;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        u64 o = __shfl_down((unsigned long long)mx, off, 64);
        mx = o > mx ? o : mx;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out_max, (unsigned long long)mx);
}
void launch_linf(const fe *coef, size_t n, u64 *out_max, hipStream_t s) {
    (void)hipMemsetAsync(out_max, 0, 8, s);
    hipLaunchKernelGGL(k_linf, dim3(grid_for(n * RE, 4096)), dim3(256), 0, s, coef, n * RE, (unsigned long long *)out_max);
}



struct BPow { fe v[8]; };
// thread = (element i, residue class r of the coefficient index); in bit-plane mode the 8 x L plane entries are loaded once and all
// K bit-planes are produced from registers (planes read once per launch, not K times)
template <int LL>
__global__ void __launch_bounds__(256) k_recompose_crt_bits(DevBb t, const int32_t *planes, size_t n_planes, u32 wit_len, BPow bp, u32 K,
                                                             fe *out, size_t ldz, size_t off) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 r = blockIdx.y;
    if (i >= wit_len) return;
    int32_t v[8][LL];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int32_t *pc = planes + (size_t)(r + TAU * q) * n_planes + i * LL;
#pragma unroll
        for (int l = 0; l < LL; l++) v[q][l] = pc[l];
    }
    int plane[8];
#pragma unroll
    for (int p = 0; p < 8; p++) plane[p] = TAU * t.slot_of_pos[p] + t.pos[r][p];
    const size_t jj = off + i;
    for (u32 k = 0; k < K; k++) {
        fe x[8], A[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            fe acc = 0;
#pragma unroll
            for (int l = 0; l < LL; l++) {
                int d = digit2(v[q][l], k);
                if (d > 0) acc = fadd(acc, bp.v[l]);
                else if (d < 0) acc = fsub(acc, bp.v[l]);
            }
            x[q] = acc;
        }
        crt8(x, A, t);
        fe *o = out + (size_t)k * RE * ldz;
#pragma unroll
        for (int p = 0; p < 8; p++) o[(size_t)plane[p] * ldz + jj] = r == 0 ? A[p] : fmul(t.tw[r][p], A[p]);
    }
}
__global__ void __launch_bounds__(256) k_recompose_crt(DevBb t, const int32_t *planes, size_t n_planes, u32 wit_len, u32 L, BPow bp, u32 K,
                                                        int mode_bits, fe *out, size_t ldz, size_t off) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 r = blockIdx.y, k = blockIdx.z;
    if (i >= wit_len) return;
    fe x[8], A[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int32_t *pc = planes + (size_t)(r + TAU * q) * n_planes + i * L;
        fe acc = 0;
        for (u32 l = 0; l < L; l++) {
            int32_t v = pc[l];
            if (mode_bits) {
                int d = digit2(v, k);
                if (d > 0) acc = fadd(acc, bp.v[l]);
                else if (d < 0) acc = fsub(acc, bp.v[l]);
            } else {
                acc = fadd(acc, fmul(bp.v[l], from_small(v)));
            }
        }
        x[q] = acc;
    }
    crt8(x, A, t);
    fe *o = out + (size_t)k * RE * ldz;
    const size_t jj = off + i;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        int plane = TAU * t.slot_of_pos[p] + t.pos[r][p];
        o[(size_t)plane * ldz + jj] = r == 0 ? A[p] : fmul(t.tw[r][p], A[p]);
    }
}
void launch_recompose_crt(const DevBb &t, const int32_t *planes, size_t n_planes, u32 wit_len, u32 L, u64 B, u32 K, int mode_bits, fe *out,
                          size_t ldz, size_t off, hipStream_t s) {
    BPow bp;
    u64 pw = 1;
    for (int l = 0; l < 8; l++) { bp.v[l] = from_canon(pw); pw = hmul(pw, B % BB_P); }
    if (mode_bits && (L == 2 || L == 4)) {
        if (L == 2) hipLaunchKernelGGL(k_recompose_crt_bits<2>, dim3(cdiv(wit_len, 256), TAU), dim3(256), 0, s, t, planes, n_planes, wit_len, bp, K, out, ldz, off);
        else hipLaunchKernelGGL(k_recompose_crt_bits<4>, dim3(cdiv(wit_len, 256), TAU), dim3(256), 0, s, t, planes, n_planes, wit_len, bp, K, out, ldz, off);
        return;
    }
    hipLaunchKernelGGL(k_recompose_crt, dim3(cdiv(wit_len, 256), TAU, K), dim3(256), 0, s, t, planes, n_planes, wit_len, L, bp, K, mode_bits, out,
                       ldz, off);
}


// ---------------------------------------------------------------------------------------------------------
// eq(x, r) table over {0,1}^nv, LSB-first (build_eq_x_r, utils/sumcheck/utils.rs:100-170): entry i =
// prod_j (bit_j(i) ? r_j : 1 - r_j); one thread per entry, nv products by pre-multiplied constants.
__global__ void __launch_bounds__(256) k_build_eq(DevBb t, const E9PreC *r, const E9PreC *omr, u32 nv, fe *eq) {
    size_t n = (size_t)1 << nv;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    E9 acc = e9_from_fe(BB_ONE);
    for (u32 j = 0; j < nv; j++) {
        const E9PreC &f = ((i >> j) & 1) ? r[j] : omr[j];
        acc = e9_mul(acc, e9p(f));
    }
#pragma unroll
    for (int c = 0; c < TAU; c++) eq[(size_t)c * n + i] = acc.c[c];
}
void launch_build_eq(const DevBb &t, const E9PreC *r, const E9PreC *omr, u32 nv, fe *eq, hipStream_t s) {
    hipLaunchKernelGGL(k_build_eq, dim3(cdiv((size_t)1 << nv, 256)), dim3(256), 0, s, t, r, omr, nv, eq);
}

// sparse mat-vec (mat_vec_mul, arith/utils.rs:52-65): out[row] = sum_k val_k (.) z[col_k]
__global__ void __launch_bounds__(256) k_spmv(DevBb t, const u32 *rowptr, const u32 *col, const fe *val, const fe *z, size_t ldz, fe *out,
                                              size_t m, int accumulate) {
    size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 slot = blockIdx.y;
    if (row >= m) return;
    E9 acc = accumulate ? ld9(out, m, slot, row) : e9_zero();
    for (u32 k = rowptr[row]; k < rowptr[row + 1]; k++) {
        E9 v;
#pragma unroll
        for (int c = 0; c < TAU; c++) v.c[c] = val[(size_t)k * RE + TAU * slot + c];
        acc = e9_add(acc, e9_mul(v, ld9(z, ldz, slot, col[k]), t.nu));
    }
    st9(out, m, slot, row, acc);
}
void launch_spmv(const DevBb &t, const u32 *rowptr, const u32 *col, const fe *val, const fe *z, size_t ldz, fe *out, size_t m,
                 int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(k_spmv, dim3(cdiv(m, 256), 8), dim3(256), 0, s, t, rowptr, col, val, z, ldz, out, m, accumulate);
}
// out = sum_{j<nm} M_j z_j in one pass (fold prepare): one launch and one write of the output instead of nm read-modify-write passes
struct SpmvSet { const u32 *rowptr[4]; const u32 *col[4]; const fe *val[4]; const fe *z[4]; u32 nm; };
__global__ void __launch_bounds__(256) k_spmv_sum(DevBb t, SpmvSet ms, size_t ldz, fe *out, size_t m) {
    size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 slot = blockIdx.y;
    if (row >= m) return;
    E9 acc = e9_zero();
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        if (j < ms.nm) {
            const u32 *rp = ms.rowptr[j], *cl = ms.col[j];
            for (u32 k = rp[row]; k < rp[row + 1]; k++) {
                E9 v;
#pragma unroll
                for (int c = 0; c < TAU; c++) v.c[c] = ms.val[j][(size_t)k * RE + TAU * slot + c];
                acc = e9_add(acc, e9_mul(v, ld9(ms.z[j], ldz, slot, cl[k]), t.nu));
            }
        }
    }
    st9(out, m, slot, row, acc);
}
void launch_spmv_sum(const DevBb &t, u32 nm, const u32 *const *rowptr, const u32 *const *col, const fe *const *val, const fe *z, size_t z_stride,
                     size_t ldz, fe *out, size_t m, hipStream_t s) {
    SpmvSet ms = {};
    ms.nm = nm;
    for (u32 j = 0; j < nm && j < 4; j++) { ms.rowptr[j] = rowptr[j]; ms.col[j] = col[j]; ms.val[j] = val[j]; ms.z[j] = z + (size_t)j * z_stride; }
    hipLaunchKernelGGL(k_spmv_sum, dim3(cdiv(m, 256), 8), dim3(256), 0, s, t, ms, ldz, out, m);
}
// q[col] = sum_{rows} eq[row] * val  (CSC).  block = 32 columns x 8 slots, the slots of a column side by side (contiguous reads of the 72 coefficient words of a
// non-zero, the eq words of a row shared by its eight slots); the sums cross LDS so that the stores are runs of 32 columns per output row (as lf::k_spmv_t_eq)
__global__ void __launch_bounds__(256) k_spmv_t_eq(DevBb t, const u32 *colptr, const u32 *rowidx, const fe *val, const fe *eq, size_t m,
                                                   fe *q, size_t n) {
    const u32 cl = threadIdx.x >> 3, slot = threadIdx.x & 7;
    const size_t cb = (size_t)blockIdx.x * 32, c0 = cb + cl;
    __shared__ fe sm[RE][33];
    E9 acc = e9_zero();
    if (c0 < n)
        for (u32 k = colptr[c0]; k < colptr[c0 + 1]; k++) {
            E9 v, e;
            size_t r = rowidx[k];
#pragma unroll
            for (int c = 0; c < TAU; c++) { v.c[c] = val[(size_t)k * RE + TAU * slot + c]; e.c[c] = eq[(size_t)c * m + r]; }
            acc = e9_add(acc, e9_mul(v, e, t.nu));
        }
#pragma unroll
    for (int c = 0; c < TAU; c++) sm[TAU * slot + c][cl] = acc.c[c];
    __syncthreads();
    for (u32 o = threadIdx.x; o < RE * 32; o += 256) {
        const u32 row = o >> 5, cc = o & 31;
        if (cb + cc < n) q[(size_t)row * n + cb + cc] = sm[row][cc];
    }
}
void launch_spmv_t_eq(const DevBb &t, const u32 *colptr, const u32 *rowidx, const fe *val, const fe *eq, size_t m, fe *q, size_t n,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_spmv_t_eq, dim3(cdiv(n, 32)), dim3(256), 0, s, t, colptr, rowidx, val, eq, m, q, n);
}

// batched inner products (evaluate_mles, utils/mle_helpers.rs:65-88, restructured as dot products)
template <int NB, bool NU2>
__global__ void __launch_bounds__(256) k_dot_batch(DevBb t, const fe *X, size_t ldx, u32 na, const fe *Y, size_t ldy, size_t n, i64 *partial) {
    // grid (8 slots, na, column blocks); partial[block][(a*NB + b)*72 + 9*slot + c].  Linear workgroup id = slot + 8 * (a + na * block):
    // the na blocks reading the same slice of Y run back to back on one XCD and share it in that L2 (with the column block as the fastest
    // index every a fetched Y from HBM again: 2.6 GB for 0.72 GB of tables)
    const u32 slot = blockIdx.x, a = blockIdx.y, bx = blockIdx.z, nbx = gridDim.z;
    i64 acc[NB * TAU];
#pragma unroll
    for (int i = 0; i < NB * TAU; i++) acc[i] = 0;
    const fe *Xa = X + (size_t)a * RE * ldx;
    for (size_t i = (size_t)bx * 256 + threadIdx.x; i < n; i += (size_t)nbx * 256) {
        E9 x = ld9(Xa, ldx, slot, i);
        E9 xn = e9_times_nu_t<NU2>(x, t.nu);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            E9 y = ld9(Y + (size_t)b * RE * ldy, ldy, slot, i);
            E9 p = e9_mul_pre(y, x, xn);
#pragma unroll
            for (int c = 0; c < TAU; c++) acc[b * TAU + c] += p.c[c];
        }
    }
    __shared__ i64 red[NB * TAU];
    block_sum_store<NB * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < NB * TAU) {
        u32 b = threadIdx.x / TAU, c = threadIdx.x % TAU;
        partial[(size_t)bx * ((size_t)na * NB * RE) + ((size_t)a * NB + b) * RE + TAU * slot + c] = red[threadIdx.x];
    }
}
void launch_dot_batch(const DevBb &t, const fe *X, size_t ldx, u32 na, const fe *Y, size_t ldy, u32 nb, size_t n, i64 *partial, u64 *out,
                      hipStream_t s) {
    u32 gb = (u32)((n + 256 * 16 - 1) / (256 * 16));   // >= 16 elements per thread: the 27-value block reduction is a fixed cost
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    dim3 g(8, na, gb);
#define BB_DOT(NBV)                                                                                                    \
    do {                                                                                                               \
        if (t.nu == BB_TWO) hipLaunchKernelGGL((k_dot_batch<NBV, true>), g, dim3(256), 0, s, t, X, ldx, na, Y, ldy, n, partial);  \
        else hipLaunchKernelGGL((k_dot_batch<NBV, false>), g, dim3(256), 0, s, t, X, ldx, na, Y, ldy, n, partial);        \
    } while (0)
    switch (nb) {
        case 1: BB_DOT(1); break;
        case 2: BB_DOT(2); break;
        case 3: BB_DOT(3); break;
        default: BB_DOT(4); break;
    }
#undef BB_DOT
    launch_reduce_rows(partial, gb, na * nb * RE, out, s);
}
__global__ void __launch_bounds__(256) k_dot_eq(DevBb t, const fe *X, size_t ldx, const fe *eq, size_t ldeq, size_t n, i64 *partial) {
    u32 slot = blockIdx.y, a = blockIdx.z, na = gridDim.z;
    i64 acc[TAU];
#pragma unroll
    for (int i = 0; i < TAU; i++) acc[i] = 0;
    const fe *Xa = X + (size_t)a * RE * ldx;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        E9 x = ld9(Xa, ldx, slot, i), e;
#pragma unroll
        for (int c = 0; c < TAU; c++) e.c[c] = eq[(size_t)c * ldeq + i];
        E9 p = e9_mul(x, e, t.nu);
#pragma unroll
        for (int c = 0; c < TAU; c++) acc[c] += p.c[c];
    }
    __shared__ i64 red[TAU];
    block_sum_store<TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < TAU) partial[(size_t)blockIdx.x * ((size_t)na * RE) + (size_t)a * RE + TAU * slot + threadIdx.x] = red[threadIdx.x];
}
void launch_dot_eq(const DevBb &t, const fe *X, size_t ldx, u32 na, const fe *eq, size_t ldeq, size_t n, i64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((n + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    hipLaunchKernelGGL(k_dot_eq, dim3(gb, 8, na), dim3(256), 0, s, t, X, ldx, eq, ldeq, n, partial);
    launch_reduce_rows(partial, gb, na * RE, out, s);
}

// f-hat evaluations without materialising f-hat (Witness::get_fhat, arith.rs:273-297, is a re-layout of f_coeff):
// T[k][c] = sum_i eq[i] * digit_k(f[i][c]);  v_d slot s = T[k][8d+s].
__global__ void __launch_bounds__(256) k_coef_eval(const int32_t *planes, size_t n, const fe *eq, size_t ldeq, u32 K, int mode_bits,
                                                   i64 *partial) {
    // grid (blocks, 72 coefficients, K-groups of 4 bit-planes); partial[block][(k*72 + c)*9 + q]
    u32 c = blockIdx.y, kg = blockIdx.z * 4;
    i64 acc[4 * TAU];
#pragma unroll
    for (int i = 0; i < 4 * TAU; i++) acc[i] = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int32_t v = planes[(size_t)c * n + i];
        fe e[TAU];
#pragma unroll
        for (int q = 0; q < TAU; q++) e[q] = eq[(size_t)q * ldeq + i];
        bool neg = v < 0;
        u32 mg = (u32)(neg ? -v : v);
        if (mode_bits) {
#pragma unroll
            for (int q = 0; q < TAU; q++) e[q] = neg ? -e[q] : e[q];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int32_t mask = -(int32_t)((mg >> (kg + k)) & 1);
#pragma unroll
                for (int q = 0; q < TAU; q++) acc[TAU * k + q] += (i64)(e[q] & mask);
            }
        } else {
#pragma unroll
            for (int q = 0; q < TAU; q++) acc[q] += (i64)e[q] * (i64)v;   // integer scaling keeps the Montgomery form
        }
    }
    // per-thread sums are < 2^50 (a few elements each): no reduction needed before the block sum
    __shared__ i64 red[4 * TAU];
    block_sum_store<4 * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < 4 * TAU) {
        u32 k = kg + threadIdx.x / TAU, q = threadIdx.x % TAU;
        if (k < K) partial[(size_t)blockIdx.x * ((size_t)K * RE * TAU) + ((size_t)k * RE + c) * TAU + q] = red[threadIdx.x];
    }
}
void launch_coef_eval(const DevBb &t, const int32_t *planes, size_t n, const fe *eq, size_t ldeq, u32 K, int mode_bits, i64 *partial, u64 *out,
                      hipStream_t s) {
    u32 gb = (u32)((n + 256 * 16 - 1) / (256 * 16));   // >= 16 elements per thread: the 36-value block reduction is the fixed cost
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    hipLaunchKernelGGL(k_coef_eval, dim3(gb, RE, (K + 3) / 4), dim3(256), 0, s, planes, n, eq, ldeq, K, mode_bits, partial);
    launch_reduce_rows(partial, gb, K * RE * TAU, out, s);
}

// zz_j = sum_k coef[k][j] * z_k   (Mz restructuring: G += sum_j M_j zz_j)
__global__ void __launch_bounds__(256) k_lincomb_z(DevBb t, const fe *z, size_t ldz, u32 K, const E9PreC *coef, u32 tt, size_t n, fe *out, u32 per_slot) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 slot = blockIdx.y;
    if (i >= n) return;
    // (the K products of an output stay lazy: their nine un-reduced column sums are added as 96-bit integers and reduced once -- nine Montgomery reductions per
    // output instead of nine per product)
    HL acc[4 * TAU];
#pragma unroll
    for (int q = 0; q < 4 * TAU; q++) hl_zero(acc[q]);
#pragma unroll 1
    for (u32 k = 0; k < K; k++) {
        E9 zk = ld9(z + (size_t)k * RE * ldz, ldz, slot, i);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((u32)j < tt) {
                const E9Pre cf = e9p(coef[per_slot ? (size_t)(k * tt + j) * 8 + slot : (size_t)(k * tt + j)]);
                i64 T[TAU];
                e9_mul_cols(zk, cf.v, cf.vn, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) hl_add(acc[j * TAU + c], T[c]);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        if ((u32)j < tt) {
#pragma unroll
            for (int c = 0; c < TAU; c++) out[((size_t)j * RE + TAU * slot + c) * ldz + i] = hl_finish(acc[j * TAU + c]);
        }
}
void launch_lincomb_z(const DevBb &t, const fe *z, size_t ldz, u32 K, const E9PreC *coef, u32 tt, size_t n, fe *out, hipStream_t s, u32 per_slot) {
    hipLaunchKernelGGL(k_lincomb_z, dim3(cdiv(n, 256), 8), dim3(256), 0, s, t, z, ldz, K, coef, tt, n, out, per_slot);
}
// G[row][slot] += sum_{k<K} sum_{d<9} apow[k][d] * digit_k(planes[8d+slot][row])   (rows < n_planes)
__global__ void __launch_bounds__(256) k_add_fhat_comb(const int32_t *planes, size_t n_planes, u32 K, const E9C *apow, fe *G, size_t m) {
    __shared__ fe sc[32 * TAU * TAU];
    for (u32 i = threadIdx.x; i < K * TAU * TAU; i += 256) sc[i] = apow[i / TAU].c[i % TAU];
    __syncthreads();
    size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 slot = blockIdx.y;
    if (row >= n_planes) return;
    i64 acc[TAU];
#pragma unroll
    for (int c = 0; c < TAU; c++) acc[c] = 0;
#pragma unroll 1
    for (int d = 0; d < TAU; d++) {
        int32_t v = planes[(size_t)(8 * d + slot) * n_planes + row];
        bool neg = v < 0;
        u32 mg = (u32)(neg ? -v : v);
        i64 loc[TAU];
#pragma unroll
        for (int c = 0; c < TAU; c++) loc[c] = 0;
        for (u32 k = 0; k < K; k++) {
            int32_t mask = -(int32_t)((mg >> k) & 1);
            const fe *ap = sc + ((size_t)k * TAU + d) * TAU;
#pragma unroll
            for (int c = 0; c < TAU; c++) loc[c] += (i64)(ap[c] & mask);
        }
#pragma unroll
        for (int c = 0; c < TAU; c++) acc[c] += neg ? -loc[c] : loc[c];
    }
#pragma unroll
    for (int c = 0; c < TAU; c++) {
        size_t o = (size_t)(TAU * slot + c) * m + row;
        G[o] = fred(acc[c] + (i64)G[o]);
    }
}
void launch_add_fhat_comb(const DevBb &t, const int32_t *planes, size_t n_planes, u32 K, const E9C *apow, fe *G, size_t m, hipStream_t s) {
    hipLaunchKernelGGL(k_add_fhat_comb, dim3(cdiv(n_planes, 256), 8), dim3(256), 0, s, planes, n_planes, K, apow, G, m);
}

// ---------------------------------------------------------------------------------------------------------
// DenseMultilinearExtension::fix_variables (sumcheck/prover.rs:70-72,112-123): new[j] = old[2j] + r (old[2j+1] - old[2j])
__global__ void __launch_bounds__(256) k_fix(DevBb t, const fe *in, size_t ld_in, fe *out, size_t ld_out, size_t n_in, E9PreC r) {
    size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t g = blockIdx.y;
    if (j >= n_in / 2) return;
    const fe *pi = in + g * TAU * ld_in;
    E9 a, b;
#pragma unroll
    for (int c = 0; c < TAU; c++) {
        int2 v = *reinterpret_cast<const int2 *>(pi + (size_t)c * ld_in + 2 * j);
        a.c[c] = v.x; b.c[c] = v.y;
    }
    E9 d = e9_mul(e9_sub(b, a), e9p(r));
    E9 o = e9_add(a, d);
    fe *po = out + g * TAU * ld_out;
#pragma unroll
    for (int c = 0; c < TAU; c++) po[(size_t)c * ld_out + j] = o.c[c];
}
void launch_fix(const DevBb &t, const fe *in, size_t ld_in, fe *out, size_t ld_out, size_t n_in, u32 rows9, const E9PreC &r, hipStream_t s) {
    if (n_in < 2 || !rows9) return;
    u32 gy = rows9;
    // grid.y is limited to 65535: split very tall tables
    for (u32 g0 = 0; g0 < gy; g0 += 32768) {
        u32 cnt = gy - g0 < 32768 ? gy - g0 : 32768;
        hipLaunchKernelGGL(k_fix, dim3(cdiv(n_in / 2, 256), cnt), dim3(256), 0, s, t, in + (size_t)g0 * TAU * ld_in, ld_in,
                           out + (size_t)g0 * TAU * ld_out, ld_out, n_in, r);
    }
}

// last fix of the folding sumcheck's f-hat tables (two entries per row): the fully fixed tables are theta = f-hat(r_o)
// (folding.rs:236-242); canonical words in theta's flat order [row][9]
__global__ void __launch_bounds__(256) k_fix_final(DevBb t, const fe *in, size_t ld_in, u32 rows9, E9PreC r, u64 *out) {
    u32 g = blockIdx.x * 256 + threadIdx.x;
    if (g >= rows9) return;
    const fe *pi = in + (size_t)g * TAU * ld_in;
    E9 a, b;
#pragma unroll
    for (int c = 0; c < TAU; c++) { a.c[c] = pi[(size_t)c * ld_in]; b.c[c] = pi[(size_t)c * ld_in + 1]; }
    E9 o = e9_add(a, e9_mul(e9_sub(b, a), e9p(r)));
#pragma unroll
    for (int c = 0; c < TAU; c++) out[(size_t)g * TAU + c] = to_canon(o.c[c]);
}
void launch_fix_final(const DevBb &t, const fe *in, size_t ld_in, u32 rows9, const E9PreC &r, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_fix_final, dim3(cdiv(rows9, 256)), dim3(256), 0, s, t, in, ld_in, rows9, r, out);
}

// ---------------------------------------------------------------------------------------------------------
// compute_f_0 (folding.rs:258-268) in the coefficient domain: f_0[j] = sum_i rho_i * f_i[j] with rho_i a short challenge
// (24 coefficients in [-32,32), rings/babybear.rs:36-68) and f_i the i-th bit-plane; X^72 = X^36 - 1.
// Nibble tables (as in the Goldilocks kernel): for one side, sum_k rho_k[a] * digit_k(v_c) = sign(v_c) * sum_q R[q][nibble_q(|v_c|)][a] with
// R[q][val][a] = sum_{b<4} bit_b(val) rho_{4q+b}[a] -- four look-ups of a 24-vector per coefficient c instead of 16 x 24 multiply-adds.
// Tables for both signs and sides (2*2*4*16*24 int32 = 24 KB) are built in LDS per block; one thread per element.  The coefficient loop
// runs in groups of 8 with a scheduling barrier between groups, so only 8 plane loads are in flight next to the 95 accumulators.
// Sliding window: coefficient c only touches output positions c..c+23, so with both sides handled per group of 8 coefficients the
// positions C0..C0+7 are final after the group -- they are stored (before the X^72 wrap) and leave the registers; 31 live accumulators
// instead of 95.  The wrap X^72 = X^36 - 1 of positions 72..94 is applied to the stored values at the end.
template <int C0>
__device__ __forceinline__ void fw_group8(int32_t (&win)[31], const int32_t *pL, const int32_t *pR, size_t n, size_t j,
                                          const int32_t (*R)[2][4][16][28], int32_t *out) {
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
            for (int q = 0; q < 4; q++) {
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
__global__ void __launch_bounds__(256) k_fold_witness(const int32_t *planesL, const int32_t *planesR, size_t n, u32 K, const int8_t *rho, int32_t *out) {
    __shared__ __align__(16) int32_t R[2][2][4][16][28];   // [side][sign][nibble][value][a]; rows padded to 28 words (bank spread)
    for (u32 idx = threadIdx.x; idx < 2 * 4 * 16 * 24; idx += 256) {
        u32 a = idx % 24, val = (idx / 24) % 16, q = (idx / (24 * 16)) % 4, side = idx / (24 * 16 * 4);
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
    fw_group8<0>(win, planesL, planesR, n, j, R, out);  fw_group8<8>(win, planesL, planesR, n, j, R, out);
    fw_group8<16>(win, planesL, planesR, n, j, R, out); fw_group8<24>(win, planesL, planesR, n, j, R, out);
    fw_group8<32>(win, planesL, planesR, n, j, R, out); fw_group8<40>(win, planesL, planesR, n, j, R, out);
    fw_group8<48>(win, planesL, planesR, n, j, R, out); fw_group8<56>(win, planesL, planesR, n, j, R, out);
    fw_group8<64>(win, planesL, planesR, n, j, R, out);
    // win[i] = position 72 + i;  X^72 = X^36 - 1
#pragma unroll
    for (int i = 0; i < 23; i++) {
        out[(size_t)(36 + i) * n + j] += win[i];
        out[(size_t)i * n + j] -= win[i];
    }
}
void launch_fold_witness(const int32_t *planesL, const int32_t *planesR, size_t n, u32 K, const int8_t *rho, int32_t *out, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_witness, dim3(cdiv(n, 256)), dim3(256), 0, s, planesL, planesR, n, K, rho, out);
}

}  // namespace lfbb


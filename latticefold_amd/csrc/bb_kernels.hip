// bb_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the BabyBearRingNTT backend (BASELINE configs[2]:
// "alt-prime 31-bit Montgomery path").  Same data-flow as the Goldilocks kernels (lf_kernels.hip): plane-major (SoA)
// tables, lane <-> consecutive element index, cross-lane reductions by wave64 shuffles + one LDS hop, the Ajtai mat-vec
// staged through LDS.  Arithmetic: centred Montgomery int32 words; an F_{p^9} product is 81 v_mad_i64_i32 into nine
// signed 64-bit column sums (9 * H^2 < 2^63) + nine Montgomery reductions (bb_field.cuh).  No MFMA.
#include "bb_kernels.h"

#include <stdlib.h>

namespace lfbb {

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
static inline unsigned grid_for(size_t n, unsigned cap = 2048) {
    size_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

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
constexpr u32 RED_BLOCKS = 256;
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

// bit-plane k of a centred small value: sign(v) * bit_k(|v|)   (base-2 balanced digits, decomposition.rs:159-167)
__device__ __forceinline__ int digit2(int32_t v, u32 k) {
    int32_t m = v < 0 ? -v : v;
    int d = (m >> k) & 1;
    return v < 0 ? -d : d;
}
__device__ __forceinline__ fe fe_from_digit(int d) { return d == 0 ? 0 : (d > 0 ? BB_ONE : -BB_ONE); }


struct BPow { fe v[8]; };
// thread = (element i, residue class r of the coefficient index, table k): 8 coefficients c = r + 9 q, one crt8 (see k_bitplane_crt)
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
// linearization sumcheck round (comb fn nifs/linearization/utils.rs:90-107): g(X) = eq(X) * sum_i c_i prod_{j in S_i} Mz_j(X),
// evaluated at X = 0..deg on every index pair by stepping vals += (v1 - v0)  (sumcheck/prover.rs:111-160)
__device__ __forceinline__ E9 pick4(const E9 (&v)[4], u32 idx) {
    E9 r;
#pragma unroll
    for (int c = 0; c < TAU; c++) r.c[c] = idx == 0 ? v[0].c[c] : (idx == 1 ? v[1].c[c] : (idx == 2 ? v[2].c[c] : v[3].c[c]));
    return r;
}
__device__ __forceinline__ E9 ldq(const fe *eq, size_t ld, size_t i) {
    E9 r;
#pragma unroll
    for (int c = 0; c < TAU; c++) r.c[c] = eq[(size_t)c * ld + i];
    return r;
}
// one pair of one slot: v[q] = Mz_q at the pair's first entry, st[q] = the step to its second, e / es likewise for eq; adds g(X) into acc[X][9]
__device__ __forceinline__ void lin_pair_eval(const DevBb &t, const LinDesc &desc, E9 (&v)[4], const E9 (&st)[4], E9 e, const E9 &es, u32 slot, u32 deg, i64 (&acc)[5 * TAU]) {
    for (u32 X = 0; X <= deg; X++) {
        if (X) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if ((u32)q < desc.t) v[q] = e9_add(v[q], st[q]);
            e = e9_add(e, es);
        }
        E9 sum = e9_zero();
        for (u32 i = 0; i < desc.q; i++) {
            E9 term;
            u32 k0 = desc.S_off[i], k1 = desc.S_off[i + 1];
            term = pick4(v, desc.S_idx[k0]);
            for (u32 k = k0 + 1; k < k1; k++) term = e9_mul(term, pick4(v, desc.S_idx[k]), t.nu);
            if (desc.c_unit[i] == 1) sum = e9_add(sum, term);
            else if (desc.c_unit[i] == -1) sum = e9_sub(sum, term);
            else {
                E9 cc;
#pragma unroll
                for (int c = 0; c < TAU; c++) cc.c[c] = desc.c[i][TAU * slot + c];
                sum = e9_add(sum, e9_mul(term, cc, t.nu));
            }
        }
        E9 g = e9_mul(sum, e, t.nu);
#pragma unroll
        for (int c = 0; c < TAU; c++)
            if (X == 0) acc[c] += g.c[c];
            else if (X == 1) acc[TAU + c] += g.c[c];
            else if (X == 2) acc[2 * TAU + c] += g.c[c];
            else if (X == 3) acc[3 * TAU + c] += g.c[c];
            else acc[4 * TAU + c] += g.c[c];
    }
}
__global__ void __launch_bounds__(256) k_lin_round(DevBb t, LinDesc desc, const fe *mz, size_t ld, const fe *eq, size_t ldeq, size_t n, u32 deg,
                                                   i64 *partial) {
    u32 slot = blockIdx.y;
    i64 acc[5 * TAU];
#pragma unroll
    for (int i = 0; i < 5 * TAU; i++) acc[i] = 0;
    size_t pairs = n / 2;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < pairs; j += (size_t)gridDim.x * 256) {
        E9 v[4], st[4], e, es;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if ((u32)q < desc.t) {
                E9 a = ld9(mz + (size_t)q * RE * ld, ld, slot, 2 * j), b = ld9(mz + (size_t)q * RE * ld, ld, slot, 2 * j + 1);
                v[q] = a; st[q] = e9_sub(b, a);
            } else { v[q] = e9_zero(); st[q] = e9_zero(); }
        }
        {
            E9 a, b;
#pragma unroll
            for (int c = 0; c < TAU; c++) { a.c[c] = eq[(size_t)c * ldeq + 2 * j]; b.c[c] = eq[(size_t)c * ldeq + 2 * j + 1]; }
            e = a; es = e9_sub(b, a);
        }
        // (the loop of lin_pair_eval, in place: as a call it costs this kernel its second wave per SIMD -- 256 instead of 231 registers)
        for (u32 X = 0; X <= deg; X++) {
            if (X) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if ((u32)q < desc.t) v[q] = e9_add(v[q], st[q]);
                e = e9_add(e, es);
            }
            E9 sum = e9_zero();
            for (u32 i = 0; i < desc.q; i++) {
                E9 term;
                u32 k0 = desc.S_off[i], k1 = desc.S_off[i + 1];
                term = pick4(v, desc.S_idx[k0]);
                for (u32 k = k0 + 1; k < k1; k++) term = e9_mul(term, pick4(v, desc.S_idx[k]), t.nu);
                if (desc.c_unit[i] == 1) sum = e9_add(sum, term);
                else if (desc.c_unit[i] == -1) sum = e9_sub(sum, term);
                else {
                    E9 cc;
#pragma unroll
                    for (int c = 0; c < TAU; c++) cc.c[c] = desc.c[i][TAU * slot + c];
                    sum = e9_add(sum, e9_mul(term, cc, t.nu));
                }
            }
            E9 g = e9_mul(sum, e, t.nu);
#pragma unroll
            for (int c = 0; c < TAU; c++)
                if (X == 0) acc[c] += g.c[c];
                else if (X == 1) acc[TAU + c] += g.c[c];
                else if (X == 2) acc[2 * TAU + c] += g.c[c];
                else if (X == 3) acc[3 * TAU + c] += g.c[c];
                else acc[4 * TAU + c] += g.c[c];
        }
    }
    __shared__ i64 red[5 * TAU];
    block_sum_store<5 * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < 5 * TAU) {
        u32 X = threadIdx.x / TAU, c = threadIdx.x % TAU;
        partial[(size_t)blockIdx.x * (5 * RE) + X * RE + TAU * slot + c] = red[threadIdx.x];
    }
}
// The R1CS shape (t = 3, q = 2: + Mz_0 Mz_1 - Mz_2 -- every row of the benchmark configurations): g(X) = eq(X) (a b - c)(X) at X = 0..3, eight F_{p^9} products per
// pair and slot with nothing interpreted -- the generic kernel above carries the multiset descriptor, a 4-way select per factor and 231 registers (415 us for the
// 226 MB of round 1 at 2^18 rows; this one is bound by that traffic).  FIX: fix_variables of the previous round fused in -- the pair is read as four entries of the
// previous tables, fixed with r and stored for the next round (mz_out / eq_out; the eq rows by the blocks of slot 0), so the tables make one trip per round.
// DIRECT (one block per slot, at most 256 pairs): the block's sums are the message -- written canonical to `out` (mapped host memory), no reduction launch.
template <bool FIX, bool DIRECT>
__global__ void __launch_bounds__(256) k_lin_r1cs(DevBb t, const fe *mz, size_t ld, const fe *eq, size_t ldeq, size_t pairs, E9PreC rfix, fe *mz_out, size_t ld_out, fe *eq_out,
                                                  size_t ldeq_out, i64 *partial, u64 *out) {
    const u32 slot = blockIdx.y;
    i64 acc[4 * TAU];
#pragma unroll
    for (int i = 0; i < 4 * TAU; i++) acc[i] = 0;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < pairs; j += (size_t)gridDim.x * 256) {
        E9 v[3], st[3], e, es;
        if (FIX) {
            const E9Pre R = e9p(rfix);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const fe *src = mz + ((size_t)q * RE + TAU * slot) * ld + 4 * j;
                E9 p0, p1, p2, p3;
#pragma unroll
                for (int c = 0; c < TAU; c++) {
                    const int4 w = *reinterpret_cast<const int4 *>(src + (size_t)c * ld);
                    p0.c[c] = w.x; p1.c[c] = w.y; p2.c[c] = w.z; p3.c[c] = w.w;
                }
                const E9 a = e9_add(p0, e9_mul(e9_sub(p1, p0), R)), b = e9_add(p2, e9_mul(e9_sub(p3, p2), R));
                fe *dst = mz_out + ((size_t)q * RE + TAU * slot) * ld_out + 2 * j;
#pragma unroll
                for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(dst + (size_t)c * ld_out) = make_int2(a.c[c], b.c[c]);
                v[q] = a; st[q] = e9_sub(b, a);
            }
            E9 p0, p1, p2, p3;
#pragma unroll
            for (int c = 0; c < TAU; c++) {
                const int4 w = *reinterpret_cast<const int4 *>(eq + (size_t)c * ldeq + 4 * j);
                p0.c[c] = w.x; p1.c[c] = w.y; p2.c[c] = w.z; p3.c[c] = w.w;
            }
            const E9 a = e9_add(p0, e9_mul(e9_sub(p1, p0), R)), b = e9_add(p2, e9_mul(e9_sub(p3, p2), R));
            if (slot == 0) {
#pragma unroll
                for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(eq_out + (size_t)c * ldeq_out + 2 * j) = make_int2(a.c[c], b.c[c]);
            }
            e = a; es = e9_sub(b, a);
        } else {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const fe *src = mz + ((size_t)q * RE + TAU * slot) * ld + 2 * j;
                E9 a, b;
#pragma unroll
                for (int c = 0; c < TAU; c++) {
                    const int2 w = *reinterpret_cast<const int2 *>(src + (size_t)c * ld);
                    a.c[c] = w.x; b.c[c] = w.y;
                }
                v[q] = a; st[q] = e9_sub(b, a);
            }
            E9 a, b;
#pragma unroll
            for (int c = 0; c < TAU; c++) {
                const int2 w = *reinterpret_cast<const int2 *>(eq + (size_t)c * ldeq + 2 * j);
                a.c[c] = w.x; b.c[c] = w.y;
            }
            e = a; es = e9_sub(b, a);
        }
#pragma unroll
        for (int X = 0; X < 4; X++) {
            if (X) {
#pragma unroll
                for (int q = 0; q < 3; q++) v[q] = e9_add(v[q], st[q]);
                e = e9_add(e, es);
            }
            const E9 g = e9_mul(e9_sub(e9_mul(v[0], v[1], t.nu), v[2]), e, t.nu);
#pragma unroll
            for (int c = 0; c < TAU; c++) acc[X * TAU + c] += g.c[c];
        }
    }
    __shared__ i64 red[4 * TAU];
    block_sum_store<4 * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < 5 * TAU) {
        const u32 X = threadIdx.x / TAU, c = threadIdx.x % TAU;
        const i64 v = X < 4 ? red[threadIdx.x] : 0;
        if (DIRECT) out[X * RE + TAU * slot + c] = to_canon(fred(v));
        else partial[(size_t)blockIdx.x * (5 * RE) + X * RE + TAU * slot + c] = v;
    }
}
// ---- persistent tail of the linearization sumcheck ----------------------------------------------------------------------------------------------------------
// Once a round has at most 256 pairs its cost is the launch, the stream synchronisation and the wake-up of the host thread (~85 us of wall clock for ~25 us of
// kernel), not arithmetic.  k_lin_tail runs ALL remaining rounds in one launch: workgroup = slot; per round it waits for the previous challenge in the host-mapped
// mailbox, fixes its slot's rows of the three tables and its private copy of the (slot-independent) eq table, evaluates the round polynomial and writes its rows
// of the message straight into the mailbox; the host -- which keeps the Poseidon transcript -- polls the eight flags, absorbs, squeezes and writes the challenge
// back.  The eight workgroups never exchange data (tables are slot-local, eq is recomputed privately), so no device-wide synchronisation exists at all.
constexpr u64 BB_TAIL_TIMEOUT_TICKS = 300000000ull;   // wall_clock64 runs at 100 MHz: 3 s
__device__ __forceinline__ void bb_wait_mem() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ u32 bb_ld_sys_u32(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// thread 0 of a workgroup: wait for challenge idx of this launch.  false on abort / timeout
__device__ bool bb_tail_wait_challenge(BbTailMail *mail, u32 idx, u32 epoch) {
    const u64 t0 = wall_clock64();
    for (u32 it = 0;; it++) {
        if (bb_ld_sys_u32(&mail->chal_seq[idx]) == epoch) break;
        if ((it & 63) == 63) {
            if (bb_ld_sys_u32(&mail->abort_seq) == epoch) return false;
            if (wall_clock64() - t0 > BB_TAIL_TIMEOUT_TICKS) {
                __hip_atomic_store(&mail->err, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
    return true;   // (the challenge words -- written by the host before the flag -- are then read by 18 lanes at once: one PCIe read instead of 18 in a row)
}
__global__ void __launch_bounds__(256) k_lin_tail(DevBb t, BbLinTailArgs A) {
    const u32 slot = blockIdx.x;
    __shared__ int32_t s_r[2 * TAU + 2];
    __shared__ u32 s_ok;
    __shared__ i64 red[4 * TAU];
    const fe *src = A.mz, *esrc = A.eq;
    size_t lds = A.ld, ldes = A.ldeq, n = A.n0;
    const size_t ldw = A.n0 / 2 < 2 ? 2 : A.n0 / 2;
    E9PreC rc = A.r_first;
    for (u32 rd = 0; rd < A.rounds; rd++) {
        if (rd > 0) {
            if (threadIdx.x == 0) s_ok = bb_tail_wait_challenge(A.mail, rd - 1, A.epoch) ? 1u : 0u;
            __syncthreads();
            if (!s_ok) return;
            if (threadIdx.x < 2 * TAU) s_r[threadIdx.x] = __hip_atomic_load(&A.mail->chal[rd - 1][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __syncthreads();
#pragma unroll
            for (int q = 0; q < TAU; q++) { rc.v[q] = s_r[q]; rc.vn[q] = s_r[TAU + q]; }
        }
        const E9Pre R = e9p(rc);
        fe *dst = A.work[rd & 1], *edst = A.eqw[rd & 1] + (size_t)slot * TAU * ldw;
        i64 acc[4 * TAU];
#pragma unroll
        for (int i = 0; i < 4 * TAU; i++) acc[i] = 0;
        const size_t pairs = n / 4;
        for (size_t j = threadIdx.x; j < pairs; j += 256) {
            E9 v[3], st[3], e, es;
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const fe *sp = src + ((size_t)q * RE + TAU * slot) * lds + 4 * j;
                E9 p0, p1, p2, p3;
#pragma unroll
                for (int c = 0; c < TAU; c++) {
                    const int4 w = *reinterpret_cast<const int4 *>(sp + (size_t)c * lds);
                    p0.c[c] = w.x; p1.c[c] = w.y; p2.c[c] = w.z; p3.c[c] = w.w;
                }
                const E9 a = e9_add(p0, e9_mul(e9_sub(p1, p0), R)), b = e9_add(p2, e9_mul(e9_sub(p3, p2), R));
                fe *dp = dst + ((size_t)q * RE + TAU * slot) * ldw + 2 * j;
#pragma unroll
                for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(dp + (size_t)c * ldw) = make_int2(a.c[c], b.c[c]);
                v[q] = a; st[q] = e9_sub(b, a);
            }
            {
                E9 p0, p1, p2, p3;
#pragma unroll
                for (int c = 0; c < TAU; c++) {
                    const int4 w = *reinterpret_cast<const int4 *>(esrc + (size_t)c * ldes + 4 * j);
                    p0.c[c] = w.x; p1.c[c] = w.y; p2.c[c] = w.z; p3.c[c] = w.w;
                }
                const E9 a = e9_add(p0, e9_mul(e9_sub(p1, p0), R)), b = e9_add(p2, e9_mul(e9_sub(p3, p2), R));
#pragma unroll
                for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(edst + (size_t)c * ldw + 2 * j) = make_int2(a.c[c], b.c[c]);
                e = a; es = e9_sub(b, a);
            }
#pragma unroll
            for (int X = 0; X < 4; X++) {
                if (X) {
#pragma unroll
                    for (int q = 0; q < 3; q++) v[q] = e9_add(v[q], st[q]);
                    e = e9_add(e, es);
                }
                const E9 g = e9_mul(e9_sub(e9_mul(v[0], v[1], t.nu), v[2]), e, t.nu);
#pragma unroll
                for (int c = 0; c < TAU; c++) acc[X * TAU + c] += g.c[c];
            }
        }
        block_sum_store<4 * TAU>(acc, red);
        __syncthreads();
        if (threadIdx.x < 5 * TAU) {
            const u32 X = threadIdx.x / TAU, c = threadIdx.x % TAU;
            const u64 w = X < 4 ? (u64)to_canon(fred(red[threadIdx.x])) : 0ull;
            __hip_atomic_store(&A.mail->msg[rd][X * RE + TAU * slot + c], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        bb_wait_mem();            // this wave's stores have completed: the message rows (write-through to host memory) before the flag, this round's table rows
        __syncthreads();          // before the next round's loads by the other waves of the workgroup (same CU: no fence, as in the Goldilocks tail)
        if (threadIdx.x == 0) __hip_atomic_store(&A.mail->msg_seq[rd][slot], A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        src = dst; lds = ldw; esrc = edst; ldes = ldw; n /= 2;
    }
}
void launch_lin_tail(const DevBb &t, const BbLinTailArgs &A, hipStream_t s) { hipLaunchKernelGGL(k_lin_tail, dim3(8), dim3(256), 0, s, t, A); }

bool lin_desc_is_r1cs(const LinDesc &d) {
    return d.t == 3 && d.q == 2 && d.S_off[0] == 0 && d.S_off[1] == 2 && d.S_off[2] == 3 && d.S_idx[0] == 0 && d.S_idx[1] == 1 && d.S_idx[2] == 2 && d.c_unit[0] == 1 && d.c_unit[1] == -1;
}
// one round of the R1CS shape: r == nullptr: the tables as they are (mz / eq hold 2 * pairs entries); else they are the previous round's (4 * pairs entries), fixed on the way
void launch_lin_r1cs(const DevBb &t, const fe *mz, size_t ld, const fe *eq, size_t ldeq, size_t pairs, const E9PreC *r, fe *mz_out, size_t ld_out, fe *eq_out, size_t ldeq_out,
                     i64 *partial, u64 *out, hipStream_t s, u32 max_blocks) {
    const E9PreC none = {};
    u32 gb = (u32)((pairs + 255) / 256);
    const u32 cap = max_blocks && max_blocks < RED_BLOCKS ? max_blocks : RED_BLOCKS;
    if (gb > cap) gb = cap;
    if (gb < 1) gb = 1;
    if (pairs <= 256) {
        if (r) hipLaunchKernelGGL((k_lin_r1cs<true, true>), dim3(1, 8), dim3(256), 0, s, t, mz, ld, eq, ldeq, pairs, *r, mz_out, ld_out, eq_out, ldeq_out, partial, out);
        else hipLaunchKernelGGL((k_lin_r1cs<false, true>), dim3(1, 8), dim3(256), 0, s, t, mz, ld, eq, ldeq, pairs, none, mz_out, ld_out, eq_out, ldeq_out, partial, out);
        return;
    }
    if (r) hipLaunchKernelGGL((k_lin_r1cs<true, false>), dim3(gb, 8), dim3(256), 0, s, t, mz, ld, eq, ldeq, pairs, *r, mz_out, ld_out, eq_out, ldeq_out, partial, out);
    else hipLaunchKernelGGL((k_lin_r1cs<false, false>), dim3(gb, 8), dim3(256), 0, s, t, mz, ld, eq, ldeq, pairs, none, mz_out, ld_out, eq_out, ldeq_out, partial, out);
    launch_reduce_rows(partial, gb, 5 * RE, out, s);
}
// Small rounds (at most 256 pairs): fix_variables of the previous round's tables with its challenge, the round evaluation and the reduction in ONE launch -- block = slot,
// thread = pair; the fixed tables go to mz_out / eq_out for the next round, the message straight to `out` (mapped host memory, rows X > deg zero).  A round
// of this size is launch- and latency-bound: four launches (two k_fix, the round, the reduction) become one.
__global__ void __launch_bounds__(256) k_lin_small(DevBb t, LinDesc desc, const fe *mz_prev, size_t ld_prev, const fe *eq_prev, size_t ldeq_prev, size_t n_prev, E9PreC rfix,
                                                   fe *mz_out, size_t ld_out, fe *eq_out, size_t ldeq_out, u32 deg, u64 *out) {
    const u32 slot = blockIdx.x;
    i64 acc[5 * TAU];
#pragma unroll
    for (int i = 0; i < 5 * TAU; i++) acc[i] = 0;
    const size_t pairs = n_prev / 4, j = threadIdx.x;
    if (j < pairs) {
        const E9Pre R = e9p(rfix);
        E9 v[4], st[4], e, es;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if ((u32)q < desc.t) {
                const fe *src = mz_prev + (size_t)q * RE * ld_prev;
                const E9 p0 = ld9(src, ld_prev, slot, 4 * j), p1 = ld9(src, ld_prev, slot, 4 * j + 1), p2 = ld9(src, ld_prev, slot, 4 * j + 2), p3 = ld9(src, ld_prev, slot, 4 * j + 3);
                const E9 a = e9_add(p0, e9_mul(e9_sub(p1, p0), R)), b = e9_add(p2, e9_mul(e9_sub(p3, p2), R));
                fe *dst = mz_out + ((size_t)q * RE + TAU * slot) * ld_out;
#pragma unroll
                for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(dst + (size_t)c * ld_out + 2 * j) = make_int2(a.c[c], b.c[c]);
                v[q] = a; st[q] = e9_sub(b, a);
            } else { v[q] = e9_zero(); st[q] = e9_zero(); }
        }
        {
            const E9 p0 = ldq(eq_prev, ldeq_prev, 4 * j), p1 = ldq(eq_prev, ldeq_prev, 4 * j + 1), p2 = ldq(eq_prev, ldeq_prev, 4 * j + 2), p3 = ldq(eq_prev, ldeq_prev, 4 * j + 3);
            const E9 a = e9_add(p0, e9_mul(e9_sub(p1, p0), R)), b = e9_add(p2, e9_mul(e9_sub(p3, p2), R));
            if (slot == 0) {
#pragma unroll
                for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(eq_out + (size_t)c * ldeq_out + 2 * j) = make_int2(a.c[c], b.c[c]);
            }
            e = a; es = e9_sub(b, a);
        }
        lin_pair_eval(t, desc, v, st, e, es, slot, deg, acc);
    }
    __shared__ i64 red[5 * TAU];
    block_sum_store<5 * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < 5 * TAU) {
        const u32 X = threadIdx.x / TAU, c = threadIdx.x % TAU;
        out[X * RE + TAU * slot + c] = to_canon(fred(red[threadIdx.x]));
    }
}
void launch_lin_small(const DevBb &t, const LinDesc &desc, const fe *mz_prev, size_t ld_prev, const fe *eq_prev, size_t ldeq_prev, size_t n_prev, const E9PreC &r, fe *mz_out, size_t ld_out,
                      fe *eq_out, size_t ldeq_out, u32 deg, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_lin_small, dim3(8), dim3(256), 0, s, t, desc, mz_prev, ld_prev, eq_prev, ldeq_prev, n_prev, r, mz_out, ld_out, eq_out, ldeq_out, deg, out);
}
void launch_lin_round(const DevBb &t, const LinDesc &desc, const fe *mz, size_t ld, const fe *eq, size_t ldeq, size_t n, u32 deg, i64 *partial,
                      u64 *out, hipStream_t s, u32 max_blocks) {
    u32 gb = (u32)((n / 2 + 255) / 256);
    const u32 cap = max_blocks && max_blocks < RED_BLOCKS ? max_blocks : RED_BLOCKS;
    if (gb > cap) gb = cap;
    if (gb < 1) gb = 1;
    hipLaunchKernelGGL(k_lin_round, dim3(gb, 8), dim3(256), 0, s, t, desc, mz, ld, eq, ldeq, n, deg, partial);
    launch_reduce_rows(partial, gb, 5 * RE, out, s);   // X = deg+1.. rows stay zero
}

// ---------------------------------------------------------------------------------------------------------
// folding sumcheck (comb fn nifs/folding/utils.rs:273-325, b = 2):
//   g(X) = eqL G1 + eqR G2 + eqB * sum_{k<2K} sum_{d<9} mu_k^{d+1} h(f_{k,d}),  h(f) = f (f^2 - 1)
// block sum of 32-bit words (a thread holds ONE pair's contribution: centred words, widened one at a time -- 45 registers instead of 90 in the epilogue)
template <int NV>
__device__ __forceinline__ void block_sum_store_fe(const fe (&v)[NV], i64 *dst) {
    __shared__ i64 sm[4][NV];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        i64 s = wave_sum((i64)v[i]);
        if (lane == 0) sm[wave][i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += 256) dst[i] = sm[0][i] + sm[1][i] + sm[2][i] + sm[3][i];
}
// the eqL*G1 + eqR*G2 part at X = 0..4, added into acc[X][9]
__device__ __forceinline__ void fold_linear_part(const DevBb &t, const FoldArgs &a, u32 slot, size_t j, i64 (&acc)[5 * TAU]) {
#pragma unroll 1
    for (int side = 0; side < 2; side++) {
        const fe *eq = side ? a.eqR : a.eqL;
        const fe *G = side ? a.G2 : a.G1;
        E9 e0 = ldq(eq, a.ld, 2 * j), e1 = ldq(eq, a.ld, 2 * j + 1);
        E9 g0 = ld9(G, a.ld, slot, 2 * j), g1 = ld9(G, a.ld, slot, 2 * j + 1);
        E9 es = e9_sub(e1, e0), gs = e9_sub(g1, g0);
        E9 e = e0, g = g0;
#pragma unroll
        for (int X = 0; X < 5; X++) {
            if (X) { e = e9_add(e, es); g = e9_add(g, gs); }
            E9 p = e9_mul(e, g, t.nu);
#pragma unroll
            for (int c = 0; c < TAU; c++) acc[X * TAU + c] += p.c[c];
        }
    }
}
__device__ __forceinline__ void fold_store(i64 (&acc)[5 * TAU], u32 slot, i64 *partial) {
    __shared__ i64 red[5 * TAU];
    block_sum_store<5 * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < 5 * TAU) {
        u32 X = threadIdx.x / TAU, c = threadIdx.x % TAU;
        partial[(size_t)blockIdx.x * (5 * RE) + X * RE + TAU * slot + c] = red[threadIdx.x];
    }
}
// the G part of a round message alone (eqL G1 + eqR G2 at X = 0..4): for the rounds whose norm part comes from elsewhere (int8 GEMM rounds, split table rounds).
// thread = (pair, slot), ten F_{p^9} products; the table kernel run without tables costs five times as much (its accumulators leave it one wave per SIMD)
__global__ void __launch_bounds__(256) k_fold_round_g(DevBb t, FoldArgs a, i64 *partial) {
    const u32 slot = blockIdx.y;
    i64 acc[5 * TAU];
#pragma unroll
    for (int i = 0; i < 5 * TAU; i++) acc[i] = 0;
    const size_t pend = a.p0 + a.pcnt;
    for (size_t j = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; j < pend; j += (size_t)gridDim.x * 256) fold_linear_part(t, a, slot, j, acc);
    fold_store(acc, slot, partial);
}
void launch_fold_round_g(const DevBb &t, const FoldArgs &a, i64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((a.pcnt + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    hipLaunchKernelGGL(k_fold_round_g, dim3(gb, 8), dim3(256), 0, s, t, a, partial);
    launch_reduce_rows(partial, gb, 5 * RE, out, s);
}
// round 1: f-hat entries are the base-2 digits themselves, so h(f0 + X (f1 - f0)) is a small integer (|.| <= 720) and
// vanishes at X = 0, 1; S(X) = sum M[k][d] * h is accumulated as exact integer multiples of the (uniform) constants.
__global__ void __launch_bounds__(256, 2) k_fold_round1(DevBb t, FoldArgs a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                                                     const E9C *Mc, i64 *partial) {
    u32 slot = blockIdx.y;
    i64 acc[5 * TAU];
#pragma unroll
    for (int i = 0; i < 5 * TAU; i++) acc[i] = 0;
    const size_t pend = a.p0 + a.pcnt;
    for (size_t j = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; j < pend; j += (size_t)gridDim.x * 256) {
        fold_linear_part(t, a, slot, j, acc);
        i64 S[3 * TAU];
#pragma unroll
        for (int i = 0; i < 3 * TAU; i++) S[i] = 0;
        bool in = 2 * j + 1 < n_planes || 2 * j < n_planes;
        if (in) {
#pragma unroll 1
            for (int side = 0; side < 2; side++) {
                const int32_t *pl = side ? planesR : planesL;
#pragma unroll 1
                for (int d = 0; d < TAU; d++) {
                    size_t base = (size_t)(8 * d + slot) * n_planes;
                    int32_t v0 = 2 * j < n_planes ? pl[base + 2 * j] : 0;
                    int32_t v1 = 2 * j + 1 < n_planes ? pl[base + 2 * j + 1] : 0;
#pragma unroll 1
                    for (u32 k = 0; k < K; k++) {
                        int f0 = digit2(v0, k), df = digit2(v1, k) - f0;
                        int f2 = f0 + 2 * df, f3 = f2 + df, f4 = f3 + df;
                        int h2 = f2 * (f2 * f2 - 1), h3 = f3 * (f3 * f3 - 1), h4 = f4 * (f4 * f4 - 1);
                        const E9C &M = Mc[(size_t)(side * K + k) * TAU + d];
#pragma unroll
                        for (int c = 0; c < TAU; c++) {
                            i64 mc = (i64)M.c[c];
                            S[c] += mc * h2; S[TAU + c] += mc * h3; S[2 * TAU + c] += mc * h4;
                        }
                    }
                }
            }
            E9 e0 = ldq(a.eqB, a.ld, 2 * j), e1 = ldq(a.eqB, a.ld, 2 * j + 1);
            E9 es = e9_sub(e1, e0);
            E9 e = e9_add(e1, es);   // X = 2
#pragma unroll
            for (int X = 2; X < 5; X++) {
                if (X > 2) e = e9_add(e, es);
                E9 sv;
#pragma unroll
                for (int c = 0; c < TAU; c++) sv.c[c] = fred(S[(X - 2) * TAU + c]);
                E9 p = e9_mul(sv, e, t.nu);
#pragma unroll
                for (int c = 0; c < TAU; c++) acc[X * TAU + c] += p.c[c];
            }
        }
    }
    fold_store(acc, slot, partial);
}
void launch_fold_round1(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                        const E9C *Mc, i64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((a.pcnt + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    hipLaunchKernelGGL(k_fold_round1, dim3(gb, 8), dim3(256), 0, s, t, a, planesL, planesR, n_planes, K, Mc, partial);
    launch_reduce_rows(partial, gb, 5 * RE, out, s);
}
// round 2, still from the planes: after fixing the first variable every f-hat entry is a + b r1 with small integers
// (a = d0, b = d1 - d0), so h(f(X)) = c0 + c1 r1 + c2 r1^2 + c3 r1^3 with small integer c_i and
//   sum_tb M_tb h = T0 + r1 T1 + r1^2 T2 + r1^3 T3,  T_i = sum_tb M_tb c_i(tb)   (exact integer multiples of the constants).
// One thread per (pair, slot, evaluation point X = blockIdx.z); a.* are the once-fixed tables (a.n = m/2).
struct R1Pow { E9PreC r1, r2, r3; };
__global__ void __launch_bounds__(256) k_fold_round2(DevBb t, FoldArgs a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                                                     const E9C *Mc, R1Pow rp, i64 *partial) {
    u32 slot = blockIdx.y;
    const int X = blockIdx.z;
    i64 acc[TAU];
#pragma unroll
    for (int i = 0; i < TAU; i++) acc[i] = 0;
    const size_t pend = a.p0 + a.pcnt;
    for (size_t j = a.p0 + (size_t)blockIdx.x * 256 + threadIdx.x; j < pend; j += (size_t)gridDim.x * 256) {
        // linear part at this X
#pragma unroll 1
        for (int side = 0; side < 2; side++) {
            const fe *eq = side ? a.eqR : a.eqL;
            const fe *G = side ? a.G2 : a.G1;
            E9 e0 = ldq(eq, a.ld, 2 * j), e1 = ldq(eq, a.ld, 2 * j + 1);
            E9 g0 = ld9(G, a.ld, slot, 2 * j), g1 = ld9(G, a.ld, slot, 2 * j + 1);
            E9 e, g;
#pragma unroll
            for (int c = 0; c < TAU; c++) {
                e.c[c] = fred((i64)e0.c[c] + (i64)X * ((i64)e1.c[c] - (i64)e0.c[c]));
                g.c[c] = fred((i64)g0.c[c] + (i64)X * ((i64)g1.c[c] - (i64)g0.c[c]));
            }
            E9 pr = e9_mul(e, g, t.nu);
#pragma unroll
            for (int c = 0; c < TAU; c++) acc[c] += pr.c[c];
        }
        if (4 * j < n_planes) {
            i64 T[4 * TAU];
#pragma unroll
            for (int i = 0; i < 4 * TAU; i++) T[i] = 0;
#pragma unroll 1
            for (int side = 0; side < 2; side++) {
                const int32_t *pl = side ? planesR : planesL;
#pragma unroll 1
                for (int d = 0; d < TAU; d++) {
                    size_t base = (size_t)(8 * d + slot) * n_planes + 4 * j;
                    int32_t v0 = pl[base], v1 = 4 * j + 1 < n_planes ? pl[base + 1] : 0;
                    int32_t v2 = 4 * j + 2 < n_planes ? pl[base + 2] : 0, v3 = 4 * j + 3 < n_planes ? pl[base + 3] : 0;
#pragma unroll 1
                    for (u32 k = 0; k < K; k++) {
                        int d0 = digit2(v0, k), d1 = digit2(v1, k), d2 = digit2(v2, k), d3 = digit2(v3, k);
                        int A0 = d0, B0 = d1 - d0, A1 = d2, B1 = d3 - d2;
                        int A = A0 + X * (A1 - A0), B = B0 + X * (B1 - B0);
                        int c0 = A * (A * A - 1), c1 = B * (3 * A * A - 1), c2 = 3 * A * B * B, c3 = B * B * B;
                        const E9C &M = Mc[(size_t)(side * K + k) * TAU + d];
#pragma unroll
                        for (int c = 0; c < TAU; c++) {
                            i64 mc = (i64)M.c[c];
                            T[c] += mc * c0; T[TAU + c] += mc * c1; T[2 * TAU + c] += mc * c2; T[3 * TAU + c] += mc * c3;
                        }
                    }
                }
            }
            E9 t0, t1, t2, t3;
#pragma unroll
            for (int c = 0; c < TAU; c++) { t0.c[c] = fred(T[c]); t1.c[c] = fred(T[TAU + c]); t2.c[c] = fred(T[2 * TAU + c]); t3.c[c] = fred(T[3 * TAU + c]); }
            E9 sv = e9_add(e9_add(t0, e9_mul(t1, e9p(rp.r1))), e9_add(e9_mul(t2, e9p(rp.r2)), e9_mul(t3, e9p(rp.r3))));
            E9 e0 = ldq(a.eqB, a.ld, 2 * j), e1 = ldq(a.eqB, a.ld, 2 * j + 1), e;
#pragma unroll
            for (int c = 0; c < TAU; c++) e.c[c] = fred((i64)e0.c[c] + (i64)X * ((i64)e1.c[c] - (i64)e0.c[c]));
            E9 pr = e9_mul(sv, e, t.nu);
#pragma unroll
            for (int c = 0; c < TAU; c++) acc[c] += pr.c[c];
        }
    }
    __shared__ i64 red[TAU];
    block_sum_store<TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < TAU) partial[(size_t)blockIdx.x * (5 * RE) + X * RE + TAU * slot + threadIdx.x] = red[threadIdx.x];
}
void launch_fold_round2(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                        const E9C *Mc, const H9 &r1, const BbHostRing &ring, i64 *partial, u64 *out, hipStream_t s) {
    u32 gb = (u32)((a.pcnt + 255) / 256);
    if (gb > RED_BLOCKS) gb = RED_BLOCKS;
    if (gb < 1) gb = 1;
    R1Pow rp;
    H9 r2 = ring.mul9(r1, r1), r3 = ring.mul9(r2, r1);
    rp.r1 = e9pre_from_h9(r1, ring.T.nu); rp.r2 = e9pre_from_h9(r2, ring.T.nu); rp.r3 = e9pre_from_h9(r3, ring.T.nu);
    hipLaunchKernelGGL(k_fold_round2, dim3(gb, 8, 5), dim3(256), 0, s, t, a, planesL, planesR, n_planes, K, Mc, rp, partial);
    launch_reduce_rows(partial, gb, 5 * RE, out, s);
}
// after r_2: F[(side*K+k)*9+d][9*slot+c][j] = sum_{b<4} W_b * digit(f[4j+b]),  W_b = eq((r1,r2), b) (b = b0 + 2 b1, LSB first)
struct W4 { fe v[4][TAU]; };
__global__ void __launch_bounds__(256) k_fold_materialize2(const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t j0, size_t quarter,
                                                           u32 K, W4 w, fe *F) {
    size_t jl = (size_t)blockIdx.x * 256 + threadIdx.x;
    u32 slot = blockIdx.y % 8, d = blockIdx.y / 8, side = blockIdx.z;
    if (jl >= quarter) return;
    const size_t j = j0 + jl;   // global entry; stored at local index jl with leading dimension `quarter`
    const int32_t *pl = side ? planesR : planesL;
    size_t base = (size_t)(8 * d + slot) * n_planes + 4 * j;
    int32_t v[4];
#pragma unroll
    for (int b = 0; b < 4; b++) v[b] = 4 * j + b < n_planes ? pl[base + b] : 0;
    for (u32 k = 0; k < K; k++) {
        fe *o = F + ((size_t)((side * K + k) * TAU + d) * RE + TAU * slot) * quarter + jl;
        int dg[4];
#pragma unroll
        for (int b = 0; b < 4; b++) dg[b] = digit2(v[b], k);
#pragma unroll
        for (int c = 0; c < TAU; c++) {
            fe x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                fe term = dg[b] == 0 ? 0 : (dg[b] > 0 ? w.v[b][c] : -w.v[b][c]);
                x = fadd(x, term);
            }
            o[(size_t)c * quarter] = x;
        }
    }
}
void launch_fold_materialize2(const DevBb &t, const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t j0, size_t q, u32 K,
                              const H9 &r1, const H9 &r2, const BbHostRing &ring, fe *F, hipStream_t s) {
    H9 one;
    for (int i = 0; i < TAU; i++) one.c[i] = i == 0;
    H9 o1, o2;
    for (int i = 0; i < TAU; i++) { o1.c[i] = hsub(one.c[i], r1.c[i]); o2.c[i] = hsub(one.c[i], r2.c[i]); }
    H9 Wb[4] = {ring.mul9(o1, o2), ring.mul9(r1, o2), ring.mul9(o1, r2), ring.mul9(r1, r2)};
    W4 w;
    for (int b = 0; b < 4; b++)
        for (int c = 0; c < TAU; c++) w.v[b][c] = from_canon(Wb[b].c[c]);
    hipLaunchKernelGGL(k_fold_materialize2, dim3(cdiv(q, 256), 8 * TAU, 2), dim3(256), 0, s, planesL, planesR, n_planes, j0, q, K, w, F);
}
// general round on the materialised tables: per table h(f0 + X df) = c0 + c1 X + c2 X^2 + c3 X^3 with
//   M c0 = p (f0^2 - 1), M c1 = q (3 f0^2 - 1), M c2 = 3 p df^2, M c3 = q df^2,   p = M f0, q = M df
// Small rounds are latency-bound if one thread walks all 2K*9 tables, so the table range is split over blockIdx.z
// (every part is linear in the tables, including the final product with eqB).
// One pair per thread (the grid covers all pairs), so nothing but the table-loop state is live inside the loop and the four
// sums of products can be kept as lazy (high, low) column sums: no Montgomery reduction per product, one per sum at the end.
// FIX: fix_variables of the previous round fused in (unsharded large rounds): F holds the PREVIOUS tables, the pair is
//   f0 = F[4j] + r (F[4j+1] - F[4j]),  f1 = F[4j+2] + r (F[4j+3] - F[4j+2])  and is stored to Fout[2j], Fout[2j+1] for the next round;
// the round kernel is ALU-bound, so the table traffic of the separate memory-bound k_fix pass disappears under it.
// MODE 3 / 4 (rounds 3 / 4 of large unsharded instances): no m/4-entry tables at all.  After two rounds an entry of table (side,k,d) is
// sum_b W_b * digit_k(plane[4j+b]) with four ternary digits -- one of 81 values independent of table and slot -- and comes from a look-up
// table in LDS indexed by the digit code (lut: [81][9] words); mode 4 also fixes the four round-3 entries of a pair with r and stores the
// first materialised tables (m/8 entries) like mode 1.  MODE 0: plain tables, MODE 1: fused fix (above).
struct FoldLut { const int32_t *planesL, *planesR; size_t n_planes; const fe *lut; const fe *mutab; const fe *sq4; const fe *mt4; E9PreC rprev; const fe *xx5, *yy5, *mt5; const fe *Esp; size_t ldEsp; };   // mutab: mode 5, [3][2K*9][81][12]; sq4 / mt4: mode 6, [81*81][12] and [2K*9][2][81][12]; rprev / xx5 / yy5 / mt5: mode 7 (r_3; [81*81][12] twice; [2K*9][4][81][12])
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
// mode 5 (round 3): per-table products M_tb * {value, value^2, value^3} of the 81 look-up values; with them a table costs two lazy
// products (the mixed terms), the pure cubes are look-ups
template <bool NU2>
__global__ void __launch_bounds__(128) k_fold_mutab(DevBb t, const fe *lut, const E9PreC *Mpre, u32 ntab, fe *mutab) {
    u32 tb = blockIdx.x, code = threadIdx.x;
    if (code >= 81) return;
    E9 L;
#pragma unroll
    for (int c = 0; c < TAU; c++) L.c[c] = lut[TAU * code + c];
    E9 m1 = e9_mul(L, e9p(Mpre[tb])), m2 = e9_mul_t<NU2>(m1, L, t.nu), m3 = e9_mul_t<NU2>(m2, L, t.nu);
    const E9 v[3] = {m1, m2, m3};
#pragma unroll
    for (int q = 0; q < 3; q++) {
        fe *o = mutab + (((size_t)q * ntab + tb) * 81 + code) * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) o[c] = c < TAU ? v[q].c[c] : 0;
    }
}
// mode 6 (round 4 without a reduced product; the Goldilocks twin is lf::k_fold_r4tab): a fixed entry is f = L[c_lo] + (r L[c_hi] - r L[c_lo]), one of 81^2
// values -- sq[c_lo * 81 + c_hi] = f^2, mt[tb][0][c] = M_tb (L[c] - r L[c]), mt[tb][1][c] = M_tb r L[c], so M_tb f = mt[tb][0][c_lo] + mt[tb][1][c_hi]
template <bool NU2>
__global__ void __launch_bounds__(256) k_fold_r4tab(DevBb t, const fe *lut, E9PreC rfix, const E9PreC *Mpre, u32 ntab, fe *sq, fe *mt) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    auto L = [&](u32 code) { E9 g; for (int c = 0; c < TAU; c++) g.c[c] = lut[TAU * code + c]; return g; };
    if (i < 6561) {
        const u32 c0 = i / 81, c1 = i % 81;
        const E9 g0 = L(c0), r0 = e9_mul(g0, e9p(rfix)), r1 = e9_mul(L(c1), e9p(rfix));
        E9 f;
#pragma unroll
        for (int c = 0; c < TAU; c++) f.c[c] = fadd(g0.c[c], fsub(r1.c[c], r0.c[c]));
        const E9 sv = e9_sqr_t<NU2>(f, t.nu);
        fe *o = sq + (size_t)i * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) o[c] = c < TAU ? sv.c[c] : 0;
    } else if (i < 6561 + ntab * 162) {
        const u32 j = i - 6561, tb = j / 162, w = (j % 162) / 81, code = j % 81;
        const E9 g = L(code), rl = e9_mul(g, e9p(rfix));
        const E9 m = e9_mul(w ? rl : e9_sub(g, rl), e9p(Mpre[tb]));
        fe *o = mt + (((size_t)tb * 2 + w) * 81 + code) * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) o[c] = c < TAU ? m.c[c] : 0;
    }
}
// mode 7 (round 5 still from the planes; twin of lf::k_fold_r5tab): an entry of the m/16-entry tables is X + Y, X = T0[c0] + T1[c1], Y = T2[c2] + T3[c3] with
// T = (1-r4)(1-r3) L, (1-r4) r3 L, r4 (1-r3) L, r4 r3 L;  xx[c0 * 81 + c1] = X^2, yy[c2 * 81 + c3] = Y^2, mt[tb][w][c] = M_tb T_w[c]
__device__ __forceinline__ E9 r5_entry(const fe *lut, u32 w, u32 code, const E9PreC &r3, const E9PreC &r4) {
    E9 g;
#pragma unroll
    for (int c = 0; c < TAU; c++) g.c[c] = lut[TAU * code + c];
    const E9 rl = e9_mul(g, e9p(r3)), a = (w & 1) ? rl : e9_sub(g, rl), ra = e9_mul(a, e9p(r4));
    return (w & 2) ? ra : e9_sub(a, ra);
}
template <bool NU2>
__global__ void __launch_bounds__(256) k_fold_r5tab(DevBb t, const fe *lut, E9PreC r3, E9PreC r4, const E9PreC *Mpre, u32 ntab, fe *xx, fe *yy, fe *mt) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < 2 * 6561) {
        const u32 hi = i / 6561, j = i % 6561, c0 = j / 81, c1 = j % 81;
        const E9 f = e9_add(r5_entry(lut, 2 * hi, c0, r3, r4), r5_entry(lut, 2 * hi + 1, c1, r3, r4)), sv = e9_sqr_t<NU2>(f, t.nu);
        fe *o = (hi ? yy : xx) + (size_t)j * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) o[c] = c < TAU ? sv.c[c] : 0;
    } else if (i < 2 * 6561 + ntab * 324) {
        const u32 j = i - 2 * 6561, tb = j / 324, w = (j % 324) / 81, code = j % 81;
        const E9 m = e9_mul(r5_entry(lut, w, code, r3, r4), e9p(Mpre[tb]));
        fe *o = mt + (((size_t)tb * 4 + w) * 81 + code) * 12;
#pragma unroll
        for (int c = 0; c < 12; c++) o[c] = c < TAU ? m.c[c] : 0;
    }
}
// SPLIT (modes 6 / 7; the Goldilocks twin is lf::k_fold_round SPLIT): eqB fixed at r_1..r_{i-1} is c_i eq(beta_i, b) E_i[p] at entry 2p + b with
// E_i = eq((beta_{i+1}..beta_s), .) one value per pair, so the norm part of the message is c_i eq(beta_i, X) (A0 + A1 X + A2 X^2 + A3 X^3),
// A_e = sum_p E_i[p] C_e(p).  The kernel leaves A0..A2 in rows 0..2 of its message (rows 3, 4 zero) -- C3 is the one coefficient that needs the fourth lazy
// product P3 of a table -- and no G part (that comes from the round kernel run without tables); the host takes A3 from g(0) + g(1) = the previous
// message at its challenge (bb_capi.cpp).  lt.Esp = E_i as [9][lt.ldEsp].
template <bool NU2, int MODE, bool SPLIT = false>
__global__ void __launch_bounds__(256) k_fold_round(DevBb t, FoldArgs a, const fe *F, size_t ldF, u32 K, const E9PreC *Mpre, E9PreC rfix,
                                                    fe *Fout, size_t ldo, FoldLut lt, i64 *partial) {
    static_assert(!SPLIT || MODE == 6 || MODE == 7, "split form: modes 6 and 7");
    constexpr bool FIX = MODE == 1;
    __shared__ fe slut[MODE == 7 ? 4 * 81 * TAU : (MODE >= 3 ? 3 * 81 * TAU : 1)];   // the 81 values, their squares, (modes 4, 6) r times the values; mode 7: T0..T3
    if (MODE == 7) {
        for (u32 i = threadIdx.x; i < 4 * 81; i += 256) {
            const E9 e = r5_entry(lt.lut, i / 81, i % 81, lt.rprev, rfix);
#pragma unroll
            for (int c = 0; c < TAU; c++) slut[TAU * i + c] = e.c[c];
        }
        __syncthreads();
    } else
    if (MODE >= 3) {
        for (u32 i = threadIdx.x; i < 2 * 81 * TAU; i += 256) slut[i] = lt.lut[i];
        __syncthreads();
        if (MODE == 4 || MODE == 6) {   // fix_variables on look-up values needs no product per entry: f = g0 + r g1 - r g0
            if (threadIdx.x < 81) {
                E9 g;
#pragma unroll
                for (int c = 0; c < TAU; c++) g.c[c] = slut[TAU * threadIdx.x + c];
                E9 rv = e9_mul(g, e9p(rfix));
#pragma unroll
                for (int c = 0; c < TAU; c++) slut[TAU * (162 + threadIdx.x) + c] = rv.c[c];
            }
            __syncthreads();
        }
    }
    constexpr bool MONO = MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7;   // the cubic in the monomial basis P0..P3 (binomials after the loop)
    i64 SP[MONO ? TAU : 1], SU[MONO ? TAU : 1];   // sum M f0, sum M f1 (lazy 64-bit sums; as 32-bit words reduced on every add they save 36 registers and cost round 4 0.1 ms)
    i64 P0s[MODE == 5 ? TAU : 1], P3s[MODE == 5 ? TAU : 1];                            // mode 5: sum M f0^3, sum M f1^3 (look-ups)
    if (MONO) {
#pragma unroll
        for (int c = 0; c < TAU; c++) { SP[c] = 0; SU[c] = 0; }
    }
    if (MODE == 5) {
#pragma unroll
        for (int c = 0; c < TAU; c++) { P0s[c] = 0; P3s[c] = 0; }
    }
    u32 slot = blockIdx.y;
    const u32 ntab = 2 * K * TAU, per = (ntab + gridDim.z - 1) / gridDim.z;
    const u32 tb0 = blockIdx.z * per, tb1 = tb0 + per < ntab ? tb0 + per : ntab;
    // small rounds (modes 0 / 1): qsplit threads share a pair and split the block's tables between them -- everything after the table loop is linear in the
    // sums, so the block reduction adds the shares up; a latency-bound thread then walks 1..5 tables instead of 9
    const u32 Q = (MODE <= 1) ? a.qsplit : 1u, tq = (MODE <= 1) ? threadIdx.x % Q : 0u;
    const size_t j = a.p0 + ((MODE <= 1) ? (size_t)blockIdx.x * (256 / Q) + threadIdx.x / Q : (size_t)blockIdx.x * 256 + threadIdx.x);
    const bool live = j < a.p0 + a.pcnt;
    const size_t jj = live ? j - a.pF0 : 0;   // index into the f-hat buffer (it starts at pair a.pF0 when sharded)
    HL C[4 * TAU];
#pragma unroll
    for (int i = 0; i < 4 * TAU; i++) hl_zero(C[i]);
#pragma unroll 1
    for (u32 tb = tb0 + tq; tb < tb1; tb += Q) {
        const fe *Ft = F + ((size_t)tb * RE + TAU * slot) * ldF;
        E9 f0, f1;
        if (MODE >= 3) {
            constexpr int NE = MODE == 7 ? 32 : ((MODE == 4 || MODE == 6) ? 16 : 8);      // plane entries behind one pair
            const u32 side = tb / (TAU * K), k = (tb / TAU) % K, d = tb % TAU;
            const int32_t *pl = (side ? lt.planesR : lt.planesL) + (size_t)(8 * d + slot) * lt.n_planes + (size_t)NE * jj;
            int32_t v[NE];
            if ((size_t)NE * jj + NE <= lt.n_planes && (lt.n_planes & 3) == 0) {
#pragma unroll
                for (int q = 0; q < NE / 4; q++) {
                    int4 w4 = *reinterpret_cast<const int4 *>(pl + 4 * q);
                    v[4 * q] = w4.x; v[4 * q + 1] = w4.y; v[4 * q + 2] = w4.z; v[4 * q + 3] = w4.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < NE; q++) v[q] = (size_t)NE * jj + q < lt.n_planes ? pl[q] : 0;
            }
            if (MODE == 7) {
                auto ld = [&](const fe *base, u32 code, E9 &o) {
                    const int4 *q = reinterpret_cast<const int4 *>(base + 12 * code);
                    int4 a0 = q[0], a1 = q[1], a2 = q[2];
                    o.c[0] = a0.x; o.c[1] = a0.y; o.c[2] = a0.z; o.c[3] = a0.w; o.c[4] = a1.x; o.c[5] = a1.y; o.c[6] = a1.z; o.c[7] = a1.w; o.c[8] = a2.x;
                };
                const fe *mt = lt.mt5 + (size_t)tb * 4 * 81 * 12;
                E9 fv[2], sq[2], mf[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const u32 c0 = digit_code4(v + 16 * e, k), c1 = digit_code4(v + 16 * e + 4, k), c2 = digit_code4(v + 16 * e + 8, k), c3 = digit_code4(v + 16 * e + 12, k);
                    const fe *t0 = slut + TAU * c0, *t1 = slut + TAU * (81 + c1), *t2 = slut + TAU * (162 + c2), *t3 = slut + TAU * (243 + c3);
                    E9 X, Y, qx, qy, m0, m1, m2, m3;
#pragma unroll
                    for (int c = 0; c < TAU; c++) { X.c[c] = fadd(t0[c], t1[c]); Y.c[c] = fadd(t2[c], t3[c]); }
                    ld(lt.xx5, c0 * 81 + c1, qx); ld(lt.yy5, c2 * 81 + c3, qy);
                    ld(mt, c0, m0); ld(mt, 81 + c1, m1); ld(mt, 162 + c2, m2); ld(mt, 243 + c3, m3);
                    const E9 xy = e9_mul_t<NU2>(X, Y, t.nu);
#pragma unroll
                    for (int c = 0; c < TAU; c++) {
                        fv[e].c[c] = fadd(X.c[c], Y.c[c]);
                        sq[e].c[c] = fadd(fadd(qx.c[c], qy.c[c]), fadd(xy.c[c], xy.c[c]));
                        mf[e].c[c] = fadd(fadd(m0.c[c], m1.c[c]), fadd(m2.c[c], m3.c[c]));
                    }
                }
                if (live) {
                    fe *Fo = Fout + ((size_t)tb * RE + TAU * slot) * ldo;
#pragma unroll
                    for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(Fo + (size_t)c * ldo + 2 * jj) = make_int2(fv[0].c[c], fv[1].c[c]);
                }
                E9 s0n = e9_times_nu_t<NU2>(sq[0], t.nu), s1n = e9_times_nu_t<NU2>(sq[1], t.nu);
                i64 T[TAU];
                e9_mul_cols(mf[0], sq[0], s0n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) hl_add(C[c], T[c]);
                e9_mul_cols(mf[1], sq[0], s0n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) hl_add(C[TAU + c], T[c]);
                e9_mul_cols(mf[0], sq[1], s1n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) { hl_add(C[2 * TAU + c], T[c]); SP[c] += mf[0].c[c]; SU[c] += mf[1].c[c]; }
                if (!SPLIT) {
                    e9_mul_cols(mf[1], sq[1], s1n, T);
#pragma unroll
                    for (int c = 0; c < TAU; c++) hl_add(C[3 * TAU + c], T[c]);
                }
                continue;
            }
            if (MODE == 5) {
                // per-table products of the look-up values: T1 = M L, T2 = M L^2, T3 = M L^3 (k_fold_mutab)
                //   P0 = sum T3[c0], P3 = sum T3[c1] (additions), P1 = sum T2[c0] * L[c1], P2 = sum T2[c1] * L[c0] (two lazy products)
                const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k);
                const fe *t1 = lt.mutab + ((size_t)tb * 81) * 12, *t2 = t1 + (size_t)ntab * 81 * 12, *t3 = t2 + (size_t)ntab * 81 * 12;
                auto ld = [&](const fe *base, u32 code, E9 &o) {
                    const int4 *q = reinterpret_cast<const int4 *>(base + 12 * code);
                    int4 a0 = q[0], a1 = q[1], a2 = q[2];
                    o.c[0] = a0.x; o.c[1] = a0.y; o.c[2] = a0.z; o.c[3] = a0.w; o.c[4] = a1.x; o.c[5] = a1.y; o.c[6] = a1.z; o.c[7] = a1.w; o.c[8] = a2.x;
                };
                E9 m10, m11, m20, m21, m30, m31, L0, L1;
                ld(t1, c0, m10); ld(t1, c1, m11); ld(t2, c0, m20); ld(t2, c1, m21); ld(t3, c0, m30); ld(t3, c1, m31);
                const fe *l0 = slut + TAU * c0, *l1 = slut + TAU * c1;
#pragma unroll
                for (int c = 0; c < TAU; c++) { L0.c[c] = l0[c]; L1.c[c] = l1[c]; }
                E9 L0n = e9_times_nu_t<NU2>(L0, t.nu), L1n = e9_times_nu_t<NU2>(L1, t.nu);
                i64 T[TAU];
                e9_mul_cols(m20, L1, L1n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) hl_add(C[TAU + c], T[c]);
                e9_mul_cols(m21, L0, L0n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) {
                    hl_add(C[2 * TAU + c], T[c]);
                    P0s[c] += m30.c[c]; P3s[c] += m31.c[c]; SP[c] += m10.c[c]; SU[c] += m11.c[c];
                }
                continue;
            } else if (MODE == 3) {
                // both ends of the pair and their squares are look-up values: with t = M f0, u = M f1 the lazy sums
                //   P0 = sum t f0^2, P1 = sum u f0^2, P2 = sum t f1^2, P3 = sum u f1^2
                // take two products by M and four lazy products per table (no squarings); C0..C3 follow by binomials after the loop
                const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k);
                const fe *l0 = slut + TAU * c0, *l1 = slut + TAU * c1, *q0 = slut + TAU * (81 + c0), *q1 = slut + TAU * (81 + c1);
                E9 s0, s1;
#pragma unroll
                for (int c = 0; c < TAU; c++) { f0.c[c] = l0[c]; f1.c[c] = l1[c]; s0.c[c] = q0[c]; s1.c[c] = q1[c]; }
                E9Pre M = e9p(Mpre[tb]);
                E9 tt = e9_mul(f0, M), uu = e9_mul(f1, M);
                E9 s0n = e9_times_nu_t<NU2>(s0, t.nu), s1n = e9_times_nu_t<NU2>(s1, t.nu);
                i64 T[TAU];
                e9_mul_cols(tt, s0, s0n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) hl_add(C[c], T[c]);
                e9_mul_cols(uu, s0, s0n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) hl_add(C[TAU + c], T[c]);
                e9_mul_cols(tt, s1, s1n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) hl_add(C[2 * TAU + c], T[c]);
                e9_mul_cols(uu, s1, s1n, T);
#pragma unroll
                for (int c = 0; c < TAU; c++) { hl_add(C[3 * TAU + c], T[c]); SP[c] += tt.c[c]; SU[c] += uu.c[c]; }
                continue;
            } else {
                const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k), c2 = digit_code4(v + 8, k), c3 = digit_code4(v + 12, k);
                const fe *g0 = slut + TAU * c0, *g2 = slut + TAU * c2;
                const fe *r0 = slut + TAU * (162 + c0), *r1 = slut + TAU * (162 + c1), *r2 = slut + TAU * (162 + c2), *r3 = slut + TAU * (162 + c3);
#pragma unroll
                for (int c = 0; c < TAU; c++) {
                    f0.c[c] = fadd(g0[c], fsub(r1[c], r0[c]));
                    f1.c[c] = fadd(g2[c], fsub(r3[c], r2[c]));
                }
                if (live && Fout) {      // (Fout is null when round 5 works from the planes as well: mode 7)
                    fe *Fo = Fout + ((size_t)tb * RE + TAU * slot) * ldo;
#pragma unroll
                    for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(Fo + (size_t)c * ldo + 2 * jj) = make_int2(f0.c[c], f1.c[c]);
                }
                if (MODE == 6) {
                    // the operands of the four lazy products are gathers: M f0 = mt[0][c0] + mt[1][c1], f0^2 = sq[c0, c1] (likewise f1 from c2, c3)
                    auto ld = [&](const fe *base, u32 code, E9 &o) {
                        const int4 *q = reinterpret_cast<const int4 *>(base + 12 * code);
                        int4 a0 = q[0], a1 = q[1], a2 = q[2];
                        o.c[0] = a0.x; o.c[1] = a0.y; o.c[2] = a0.z; o.c[3] = a0.w; o.c[4] = a1.x; o.c[5] = a1.y; o.c[6] = a1.z; o.c[7] = a1.w; o.c[8] = a2.x;
                    };
                    const fe *ma = lt.mt4 + (size_t)tb * 2 * 81 * 12, *mb = ma + 81 * 12;
                    E9 xa, xb, ya, yb, s0, s1, tt, uu;
                    ld(ma, c0, xa); ld(mb, c1, xb); ld(ma, c2, ya); ld(mb, c3, yb);
                    ld(lt.sq4, c0 * 81 + c1, s0); ld(lt.sq4, c2 * 81 + c3, s1);
#pragma unroll
                    for (int c = 0; c < TAU; c++) { tt.c[c] = fadd(xa.c[c], xb.c[c]); uu.c[c] = fadd(ya.c[c], yb.c[c]); }
                    E9 s0n = e9_times_nu_t<NU2>(s0, t.nu), s1n = e9_times_nu_t<NU2>(s1, t.nu);
                    i64 T[TAU];
                    e9_mul_cols(tt, s0, s0n, T);
#pragma unroll
                    for (int c = 0; c < TAU; c++) hl_add(C[c], T[c]);
                    e9_mul_cols(uu, s0, s0n, T);
#pragma unroll
                    for (int c = 0; c < TAU; c++) hl_add(C[TAU + c], T[c]);
                    e9_mul_cols(tt, s1, s1n, T);
#pragma unroll
                    for (int c = 0; c < TAU; c++) { hl_add(C[2 * TAU + c], T[c]); SP[c] += tt.c[c]; SU[c] += uu.c[c]; }
                    if (!SPLIT) {
                        e9_mul_cols(uu, s1, s1n, T);
#pragma unroll
                        for (int c = 0; c < TAU; c++) hl_add(C[3 * TAU + c], T[c]);
                    }
                    continue;
                }
            }
        } else if (FIX) {
            E9 a0, a1, b0, b1;
#pragma unroll
            for (int c = 0; c < TAU; c++) {
                int4 v = *reinterpret_cast<const int4 *>(Ft + (size_t)c * ldF + 4 * jj);
                a0.c[c] = v.x; a1.c[c] = v.y; b0.c[c] = v.z; b1.c[c] = v.w;
            }
            E9Pre R = e9p(rfix);
            f0 = e9_add(a0, e9_mul(e9_sub(a1, a0), R));
            f1 = e9_add(b0, e9_mul(e9_sub(b1, b0), R));
            if (live) {
                fe *Fo = Fout + ((size_t)tb * RE + TAU * slot) * ldo;
#pragma unroll
                for (int c = 0; c < TAU; c++) *reinterpret_cast<int2 *>(Fo + (size_t)c * ldo + 2 * jj) = make_int2(f0.c[c], f1.c[c]);
            }
        } else {
#pragma unroll
            for (int c = 0; c < TAU; c++) {
                int2 v = *reinterpret_cast<const int2 *>(Ft + (size_t)c * ldF + 2 * jj);
                f0.c[c] = v.x; f1.c[c] = v.y;
            }
        }
        E9 df = e9_sub(f1, f0);
        E9Pre M = e9p(Mpre[tb]);
        E9 p = e9_mul(f0, M), q = e9_mul(df, M);
        E9 s0 = e9_sqr_t<NU2>(f0, t.nu), sd = e9_sqr_t<NU2>(df, t.nu);
        E9 u = s0; u.c[0] = fsub(u.c[0], BB_ONE);                    // f0^2 - 1
        E9 un = e9_times_nu_t<NU2>(u, t.nu);
        E9 w = e9_add(e9_add(s0, s0), s0); w.c[0] = fsub(w.c[0], BB_ONE);   // 3 f0^2 - 1 (two centred additions per word)
        E9 wn;
        if (NU2) wn = e9_times_nu_t<true>(w, t.nu);                  // doubling
        else {                                                       // nu * w = 3 (nu u) + 2 nu   (linear: no second pre-multiplication)
#pragma unroll
            for (int c = 0; c < TAU; c++) wn.c[c] = fred(3 * (i64)un.c[c] + (c == 0 ? 2 * (i64)t.nu : 0));
        }
        E9 sdn = e9_times_nu_t<NU2>(sd, t.nu);
        i64 T[TAU];
        e9_mul_cols(p, u, un, T);
#pragma unroll
        for (int c = 0; c < TAU; c++) hl_add(C[c], T[c]);
        e9_mul_cols(q, w, wn, T);
#pragma unroll
        for (int c = 0; c < TAU; c++) hl_add(C[TAU + c], T[c]);
        e9_mul_cols(p, sd, sdn, T);
#pragma unroll
        for (int c = 0; c < TAU; c++) hl_add(C[2 * TAU + c], T[c]);
        e9_mul_cols(q, sd, sdn, T);
#pragma unroll
        for (int c = 0; c < TAU; c++) hl_add(C[3 * TAU + c], T[c]);
    }
    // (a thread holds ONE pair's contribution: 32-bit words, the G part one side at a time -- as 45 64-bit sums next to fold_linear_part's operands this epilogue,
    // not the table loop, set the register count of every mode: 256 + 42..186 -> see the resource table in profiles/r04c_bb_fold_regs.txt)
    fe acc[5 * TAU];
#pragma unroll
    for (int i = 0; i < 5 * TAU; i++) acc[i] = 0;
    if (live && SPLIT) {
        E9 c0, c1, c2;
#pragma unroll
        for (int c = 0; c < TAU; c++) {
            const i64 P0 = hl_finish(C[c]), P1 = hl_finish(C[TAU + c]), P2 = hl_finish(C[2 * TAU + c]), sp = fred(SP[c]), su = fred(SU[c]);
            c0.c[c] = fred(P0 - sp); c1.c[c] = fred(3 * (P1 - P0) - (su - sp)); c2.c[c] = fred(3 * (P2 - 2 * P1 + P0));
        }
        const E9 E = ldq(lt.Esp, lt.ldEsp, j);
        const E9 a0 = e9_mul_t<NU2>(c0, E, t.nu), a1 = e9_mul_t<NU2>(c1, E, t.nu), a2 = e9_mul_t<NU2>(c2, E, t.nu);
#pragma unroll
        for (int c = 0; c < TAU; c++) { acc[c] = a0.c[c]; acc[TAU + c] = a1.c[c]; acc[2 * TAU + c] = a2.c[c]; }
    } else
    if (live) {
        // S(X) = C0 + C1 X + 3 C2 X^2 + C3 X^3
        {
            E9 c0, c1, c2, c3;
#pragma unroll
            for (int c = 0; c < TAU; c++) {
                if (MONO) {
                    // C0 = P0 - sp, C1 = 3 (P1 - P0) - (su - sp), 3 C2 = 3 (P2 - 2 P1 + P0), C3 = P3 - 3 P2 + 3 P1 - P0   (values of a few p: one reduction)
                    i64 P0 = MODE == 5 ? (i64)fred(P0s[c]) : (i64)hl_finish(C[c]), P1 = hl_finish(C[TAU + c]), P2 = hl_finish(C[2 * TAU + c]);
                    i64 P3 = MODE == 5 ? (i64)fred(P3s[c]) : (i64)hl_finish(C[3 * TAU + c]);
                    i64 sp = fred(SP[c]), su = fred(SU[c]);
                    c0.c[c] = fred(P0 - sp); c1.c[c] = fred(3 * (P1 - P0) - (su - sp));
                    c2.c[c] = fred(3 * (P2 - 2 * P1 + P0)); c3.c[c] = fred(P3 - 3 * P2 + 3 * P1 - P0);
                } else {
                    c0.c[c] = hl_finish(C[c]); c1.c[c] = hl_finish(C[TAU + c]);
                    c2.c[c] = fred(3 * (i64)hl_finish(C[2 * TAU + c])); c3.c[c] = hl_finish(C[3 * TAU + c]);
                }
            }
            E9 e0 = ldq(a.eqB, a.ld, 2 * j), e1 = ldq(a.eqB, a.ld, 2 * j + 1);
            E9 es = e9_sub(e1, e0), e = e0;
#pragma unroll
            for (int X = 0; X < 5; X++) {
                if (X) e = e9_add(e, es);
                E9 sv;
#pragma unroll
                for (int c = 0; c < TAU; c++)
                    sv.c[c] = fred((i64)c0.c[c] + (i64)c1.c[c] * X + (i64)c2.c[c] * (X * X) + (i64)c3.c[c] * (X * X * X));
                E9 pr = e9_mul_t<NU2>(sv, e, t.nu);
#pragma unroll
                for (int c = 0; c < TAU; c++) acc[X * TAU + c] = pr.c[c];
            }
        }
        if (blockIdx.z == 0 && tq == 0) {
#pragma unroll 1
            for (int side = 0; side < 2; side++) {       // the G part: eqL G1 + eqR G2 at X = 0..4
                const fe *eq = side ? a.eqR : a.eqL;
                const fe *G = side ? a.G2 : a.G1;
                E9 q0 = ldq(eq, a.ld, 2 * j), q1 = ldq(eq, a.ld, 2 * j + 1);
                E9 g0 = ld9(G, a.ld, slot, 2 * j), g1 = ld9(G, a.ld, slot, 2 * j + 1);
                E9 qs = e9_sub(q1, q0), gs = e9_sub(g1, g0);
#pragma unroll
                for (int X = 0; X < 5; X++) {
                    if (X) { q0 = e9_add(q0, qs); g0 = e9_add(g0, gs); }
                    E9 p = e9_mul(q0, g0, t.nu);
#pragma unroll
                    for (int c = 0; c < TAU; c++) acc[X * TAU + c] = fadd(acc[X * TAU + c], p.c[c]);
                }
            }
        }
    }
    __shared__ i64 red[5 * TAU];
    block_sum_store_fe<5 * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < 5 * TAU) {
        u32 X = threadIdx.x / TAU, c = threadIdx.x % TAU;
        partial[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * (5 * RE) + X * RE + TAU * slot + c] = red[threadIdx.x];
    }
}
// Round 5 from the planes on TWO lanes per pair (mode 7's arithmetic, another thread mapping).  k_fold_round<., 7> keeps a whole pair in one thread: three or four
// lazy sums (27 - 36 x 96 bits), both entries' look-ups and their products -- 298 registers, one wave per SIMD next to twelve L2 gathers per pair and table:
// 2.2 ms for 8 192 pairs at C3 when round 4 (mode 6, two waves) takes 1.25 ms for twice as many.  Here lane e of a pair builds entry e only (X, Y, X Y, f^2, M f:
// half the gathers and one reduced product per lane), gets the other entry's M f from its neighbour (nine lane swaps) and forms the two lazy products that use ITS
// f^2: (M f_e)(f_e^2) and (M f_{1-e})(f_e^2) -- P0, P1 on lane 0, P3, P2 on lane 1: all four products of the cubic, balanced, so the round runs unsplit with the
// G part inside.  Lane 0 collects the four sums at the end and finishes the pair as the MONO epilogue of k_fold_round does.
template <bool NU2>
__global__ void __launch_bounds__(256) k_fold_round5_2l(DevBb t, FoldArgs a, u32 K, E9PreC rfix, fe *Fout, size_t ldo, FoldLut lt, i64 *partial) {
    __shared__ fe slut[4 * 81 * TAU];
    for (u32 i = threadIdx.x; i < 4 * 81; i += 256) {
        const E9 e = r5_entry(lt.lut, i / 81, i % 81, lt.rprev, rfix);
#pragma unroll
        for (int c = 0; c < TAU; c++) slut[TAU * i + c] = e.c[c];
    }
    __syncthreads();
    const u32 slot = blockIdx.y, en = threadIdx.x & 1;
    const u32 ntab = 2 * K * TAU;
    const size_t j = a.p0 + (size_t)blockIdx.x * 128 + (threadIdx.x >> 1);
    const bool live = j < a.p0 + a.pcnt;
    const size_t jj = live ? j - a.pF0 : 0;
    HL CA[TAU], CB[TAU];
    i64 SM[TAU];
#pragma unroll
    for (int c = 0; c < TAU; c++) { hl_zero(CA[c]); hl_zero(CB[c]); SM[c] = 0; }
    auto ld = [&](const fe *base, u32 code, E9 &o) {
        const int4 *q = reinterpret_cast<const int4 *>(base + 12 * code);
        int4 a0 = q[0], a1 = q[1], a2 = q[2];
        o.c[0] = a0.x; o.c[1] = a0.y; o.c[2] = a0.z; o.c[3] = a0.w; o.c[4] = a1.x; o.c[5] = a1.y; o.c[6] = a1.z; o.c[7] = a1.w; o.c[8] = a2.x;
    };
#pragma unroll 1
    for (u32 tb = 0; tb < ntab; tb++) {
        const u32 side = tb / (TAU * K), k = (tb / TAU) % K, d = tb % TAU;
        const size_t pos = (size_t)32 * jj + 16 * en;
        const int32_t *pl = (side ? lt.planesR : lt.planesL) + (size_t)(8 * d + slot) * lt.n_planes + pos;
        int32_t v[16];
        if (pos + 16 <= lt.n_planes && (lt.n_planes & 3) == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int4 w4 = *reinterpret_cast<const int4 *>(pl + 4 * q);
                v[4 * q] = w4.x; v[4 * q + 1] = w4.y; v[4 * q + 2] = w4.z; v[4 * q + 3] = w4.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = pos + q < lt.n_planes ? pl[q] : 0;
        }
        const u32 c0 = digit_code4(v, k), c1 = digit_code4(v + 4, k), c2 = digit_code4(v + 8, k), c3 = digit_code4(v + 12, k);
        const fe *t0 = slut + TAU * c0, *t1 = slut + TAU * (81 + c1), *t2 = slut + TAU * (162 + c2), *t3 = slut + TAU * (243 + c3);
        const fe *mt = lt.mt5 + (size_t)tb * 4 * 81 * 12;
        // (in stages with scheduling barriers between them: hoisting all twelve loads of the body to its top costs the second wave per SIMD)
        E9 sq, mf, mfo;
        {
            E9 m0, m1;
            ld(mt, c0, m0); ld(mt, 81 + c1, m1);
#pragma unroll
            for (int c = 0; c < TAU; c++) mf.c[c] = fadd(m0.c[c], m1.c[c]);
            ld(mt, 162 + c2, m0); ld(mt, 243 + c3, m1);
#pragma unroll
            for (int c = 0; c < TAU; c++) mf.c[c] = fadd(mf.c[c], fadd(m0.c[c], m1.c[c]));
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            E9 X, Y;
#pragma unroll
            for (int c = 0; c < TAU; c++) { X.c[c] = fadd(t0[c], t1[c]); Y.c[c] = fadd(t2[c], t3[c]); }
            if (live) {
                fe *Fo = Fout + ((size_t)tb * RE + TAU * slot) * ldo + 2 * jj + en;
#pragma unroll
                for (int c = 0; c < TAU; c++) Fo[(size_t)c * ldo] = fadd(X.c[c], Y.c[c]);
            }
            sq = e9_mul_t<NU2>(X, Y, t.nu);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            E9 qx, qy;
            ld(lt.xx5, c0 * 81 + c1, qx); ld(lt.yy5, c2 * 81 + c3, qy);
#pragma unroll
            for (int c = 0; c < TAU; c++) sq.c[c] = fadd(fadd(qx.c[c], qy.c[c]), fadd(sq.c[c], sq.c[c]));
        }
#pragma unroll
        for (int c = 0; c < TAU; c++) {
            mfo.c[c] = __shfl_xor(mf.c[c], 1);
            SM[c] += mf.c[c];
        }
        __builtin_amdgcn_sched_barrier(0);
        const E9 sn = e9_times_nu_t<NU2>(sq, t.nu);
        i64 T[TAU];
        e9_mul_cols(mf, sq, sn, T);
#pragma unroll
        for (int c = 0; c < TAU; c++) hl_add(CA[c], T[c]);
        e9_mul_cols(mfo, sq, sn, T);
#pragma unroll
        for (int c = 0; c < TAU; c++) hl_add(CB[c], T[c]);
    }
    // lane 0: P0 = CA, P1 = CB, sp = SM; lane 1: P3 = CA, P2 = CB, su = SM
    fe acc[5 * TAU];      // (32-bit words and one side of the G part at a time: this epilogue, not the table loop, set the kernel's register count)
#pragma unroll
    for (int i = 0; i < 5 * TAU; i++) acc[i] = 0;
    E9 c0, c1, c2, c3;
#pragma unroll
    for (int c = 0; c < TAU; c++) {
        const fe pa = hl_finish(CA[c]), pb = hl_finish(CB[c]), sm = fred(SM[c]);
        const fe oa = __shfl_xor(pa, 1), ob = __shfl_xor(pb, 1), om = __shfl_xor(sm, 1);
        const i64 P0 = pa, P1 = pb, P2 = ob, P3 = oa, sp = sm, su = om;     // (meaningful on lane 0)
        c0.c[c] = fred(P0 - sp); c1.c[c] = fred(3 * (P1 - P0) - (su - sp));
        c2.c[c] = fred(3 * (P2 - 2 * P1 + P0)); c3.c[c] = fred(P3 - 3 * P2 + 3 * P1 - P0);
    }
    if (live && en == 0) {
        E9 e0 = ldq(a.eqB, a.ld, 2 * j), e1 = ldq(a.eqB, a.ld, 2 * j + 1);
        E9 es = e9_sub(e1, e0), e = e0;
#pragma unroll
        for (int X = 0; X < 5; X++) {
            if (X) e = e9_add(e, es);
            E9 sv;
#pragma unroll
            for (int c = 0; c < TAU; c++)
                sv.c[c] = fred((i64)c0.c[c] + (i64)c1.c[c] * X + (i64)c2.c[c] * (X * X) + (i64)c3.c[c] * (X * X * X));
            E9 pr = e9_mul_t<NU2>(sv, e, t.nu);
#pragma unroll
            for (int c = 0; c < TAU; c++) acc[X * TAU + c] = pr.c[c];
        }
#pragma unroll 1
        for (int side = 0; side < 2; side++) {       // the G part: eqL G1 + eqR G2 at X = 0..4
            const fe *eq = side ? a.eqR : a.eqL;
            const fe *G = side ? a.G2 : a.G1;
            E9 q0 = ldq(eq, a.ld, 2 * j), q1 = ldq(eq, a.ld, 2 * j + 1);
            E9 g0 = ld9(G, a.ld, slot, 2 * j), g1 = ld9(G, a.ld, slot, 2 * j + 1);
            E9 qs = e9_sub(q1, q0), gs = e9_sub(g1, g0);
#pragma unroll
            for (int X = 0; X < 5; X++) {
                if (X) { q0 = e9_add(q0, qs); g0 = e9_add(g0, gs); }
                E9 p = e9_mul(q0, g0, t.nu);
#pragma unroll
                for (int c = 0; c < TAU; c++) acc[X * TAU + c] = fadd(acc[X * TAU + c], p.c[c]);
            }
        }
    }
    __shared__ i64 red[5 * TAU];
    block_sum_store_fe<5 * TAU>(acc, red);
    __syncthreads();
    if (threadIdx.x < 5 * TAU) {
        u32 X = threadIdx.x / TAU, c = threadIdx.x % TAU;
        partial[(size_t)blockIdx.x * (5 * RE) + X * RE + TAU * slot + c] = red[threadIdx.x];
    }
}
// rows of `partial` a general round may write (one block per 256 pairs, times the table chunks)
size_t fold_partial_words(size_t m) {
    size_t rows = m / 8 / 256;          // the largest general round (round 3) has m/8 pairs
    if (rows < RED_BLOCKS) rows = RED_BLOCKS;
    return rows * 5 * RE;
}
static void launch_fold_round_impl(const DevBb &t, const FoldArgs &a, const fe *F, size_t ldF, u32 K, const E9PreC *Mpre, int mode, E9PreC rfix,
                                   fe *Fout, size_t ldo, FoldLut lt, i64 *partial, u64 *out, hipStream_t s) {
    size_t pairs = a.pcnt;
    u32 gb = (u32)((pairs + 255) / 256);
    if (gb < 1) gb = 1;
    FoldArgs a2 = a;
    a2.qsplit = 1;
    static const bool no_small = getenv("LF_FOLD_NO_SMALL") != nullptr;
    if (mode <= 1 && pairs <= 128 && !no_small)
        while (a2.qsplit < 16 && pairs * a2.qsplit * 2 <= 256) a2.qsplit *= 2;
    // enough threads to fill the chip (~128k): split the 2K*9 tables when there are few pairs
    u32 tch = 1;
    static const size_t chunk_threads = [] { const char *e = getenv("LF_FOLD_CHUNK_THREADS"); return e ? (size_t)atoll(e) : ((size_t)1 << 17); }();
    while (tch < 32 && pairs * 8 * tch < chunk_threads) tch *= 2;
    while (tch > 1 && (size_t)gb * tch > RED_BLOCKS) tch /= 2;
    const bool nu2 = t.nu == BB_TWO;
    if (mode >= 3) tch = 1;   // the planes of one (side, d) serve all K tables: no table split (large rounds only)
    const bool split = lt.Esp != nullptr && (mode == 6 || mode == 7);
#define BB_FR(N2, MD) hipLaunchKernelGGL((k_fold_round<N2, MD>), dim3(gb, 8, tch), dim3(256), 0, s, t, a2, F, ldF, K, Mpre, rfix, Fout, ldo, lt, partial)
#define BB_FRS(N2, MD) hipLaunchKernelGGL((k_fold_round<N2, MD, true>), dim3(gb, 8, tch), dim3(256), 0, s, t, a2, F, ldF, K, Mpre, rfix, Fout, ldo, lt, partial)
#define BB_FRM(N2)                                                              \
    do {                                                                        \
        if (split && mode == 6) BB_FRS(N2, 6); else if (split) BB_FRS(N2, 7);   \
        else if (mode == 1) BB_FR(N2, 1); else if (mode == 3) BB_FR(N2, 3);     \
        else if (mode == 4) BB_FR(N2, 4); else if (mode == 5) BB_FR(N2, 5);     \
        else if (mode == 6) BB_FR(N2, 6); else if (mode == 7) BB_FR(N2, 7);     \
        else BB_FR(N2, 0);                                                      \
    } while (0)
    if (nu2) BB_FRM(true); else BB_FRM(false);
#undef BB_FRM
#undef BB_FRS
#undef BB_FR
    launch_reduce_rows(partial, gb * tch, 5 * RE, out, s);
}
void launch_fold_round(const DevBb &t, const FoldArgs &a, const fe *F, size_t ldF, u32 K, const E9PreC *Mpre, i64 *partial, u64 *out,
                       hipStream_t s) {
    E9PreC none = {};
    FoldLut nl = {};
    launch_fold_round_impl(t, a, F, ldF, K, Mpre, 0, none, nullptr, 0, nl, partial, out, s);
}
void launch_fold_round_lut(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                           u32 K, const E9PreC *Mpre, i64 *partial, u64 *out, hipStream_t s) {
    E9PreC none = {};
    FoldLut lt = {planesL, planesR, n_planes, lut_dev, nullptr, nullptr, nullptr, {}, nullptr, nullptr, nullptr};
    launch_fold_round_impl(t, a, nullptr, 0, K, Mpre, 3, none, nullptr, 0, lt, partial, out, s);
}
// round 3 with per-table products of the look-up values (mutab_dev: 3 * 2K*9 * 81 * 12 words, filled by this call)
void launch_fold_round_lut_mu(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                              fe *mutab_dev, u32 K, const E9PreC *Mpre, i64 *partial, u64 *out, hipStream_t s) {
    const u32 ntab = 2 * K * TAU;
    if (t.nu == BB_TWO) hipLaunchKernelGGL((k_fold_mutab<true>), dim3(ntab), dim3(128), 0, s, t, lut_dev, Mpre, ntab, mutab_dev);
    else hipLaunchKernelGGL((k_fold_mutab<false>), dim3(ntab), dim3(128), 0, s, t, lut_dev, Mpre, ntab, mutab_dev);
    E9PreC none = {};
    FoldLut lt = {planesL, planesR, n_planes, lut_dev, mutab_dev, nullptr, nullptr, {}, nullptr, nullptr, nullptr};
    launch_fold_round_impl(t, a, nullptr, 0, K, Mpre, 5, none, nullptr, 0, lt, partial, out, s);
}
void launch_fold_round_lut_fix(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                               const H9 &r, const BbHostRing &ring, fe *Fout, size_t ldout, u32 K, const E9PreC *Mpre, i64 *partial, u64 *out,
                               hipStream_t s) {
    FoldLut lt = {planesL, planesR, n_planes, lut_dev, nullptr, nullptr, nullptr, {}, nullptr, nullptr, nullptr};
    launch_fold_round_impl(t, a, nullptr, 0, K, Mpre, 4, e9pre_from_h9(r, ring.T.nu), Fout, ldout, lt, partial, out, s);
}
// round 4 through the product-free tables of mode 6 (sq_dev 6561*12 words, mt_dev 2K*9*2*81*12 words, filled by this call)
void launch_fold_round_lut_fix_tab(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                                   const H9 &r, const BbHostRing &ring, fe *sq_dev, fe *mt_dev, fe *Fout, size_t ldout, u32 K, const E9PreC *Mpre, i64 *partial,
                                   u64 *out, hipStream_t s, const fe *Esp, size_t ldEsp) {
    const u32 ntab = 2 * K * TAU;
    const E9PreC rp = e9pre_from_h9(r, ring.T.nu);
    const u32 grid = (6561 + ntab * 162 + 255) / 256;
    if (t.nu == BB_TWO) hipLaunchKernelGGL((k_fold_r4tab<true>), dim3(grid), dim3(256), 0, s, t, lut_dev, rp, Mpre, ntab, sq_dev, mt_dev);
    else hipLaunchKernelGGL((k_fold_r4tab<false>), dim3(grid), dim3(256), 0, s, t, lut_dev, rp, Mpre, ntab, sq_dev, mt_dev);
    FoldLut lt = {planesL, planesR, n_planes, lut_dev, nullptr, sq_dev, mt_dev, {}, nullptr, nullptr, nullptr, Esp, ldEsp};
    launch_fold_round_impl(t, a, nullptr, 0, K, Mpre, 6, rp, Fout, ldout, lt, partial, out, s);
}
// round 5 from the planes (mode 7): xx_dev / yy_dev 6561*12 words each, mt_dev 2K*9*4*81*12 words, filled by this call; r3 / r4: the challenges of rounds 3 / 4
void launch_fold_round_lut_fix5(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                                const H9 &r3, const H9 &r4, const BbHostRing &ring, fe *xx_dev, fe *yy_dev, fe *mt_dev, fe *Fout, size_t ldout, u32 K,
                                const E9PreC *Mpre, i64 *partial, u64 *out, hipStream_t s, const fe *Esp, size_t ldEsp) {
    const u32 ntab = 2 * K * TAU;
    const E9PreC r3p = e9pre_from_h9(r3, ring.T.nu), r4p = e9pre_from_h9(r4, ring.T.nu);
    const u32 grid = (2 * 6561 + ntab * 324 + 255) / 256;
    if (t.nu == BB_TWO) hipLaunchKernelGGL((k_fold_r5tab<true>), dim3(grid), dim3(256), 0, s, t, lut_dev, r3p, r4p, Mpre, ntab, xx_dev, yy_dev, mt_dev);
    else hipLaunchKernelGGL((k_fold_r5tab<false>), dim3(grid), dim3(256), 0, s, t, lut_dev, r3p, r4p, Mpre, ntab, xx_dev, yy_dev, mt_dev);
    FoldLut lt = {planesL, planesR, n_planes, lut_dev, nullptr, nullptr, nullptr, r3p, xx_dev, yy_dev, mt_dev, Esp, ldEsp};
    static const bool one_lane = getenv("LF_FOLD_R5_ONE_LANE") != nullptr;
    if (!Esp && !one_lane && Fout) {     // (unsplit callers: two lanes per pair, k_fold_round5_2l)
        const u32 gb = (u32)((a.pcnt + 127) / 128);
        if (t.nu == BB_TWO) hipLaunchKernelGGL((k_fold_round5_2l<true>), dim3(gb, 8), dim3(256), 0, s, t, a, K, r4p, Fout, ldout, lt, partial);
        else hipLaunchKernelGGL((k_fold_round5_2l<false>), dim3(gb, 8), dim3(256), 0, s, t, a, K, r4p, Fout, ldout, lt, partial);
        launch_reduce_rows(partial, gb, 5 * RE, out, s);
        return;
    }
    launch_fold_round_impl(t, a, nullptr, 0, K, Mpre, 7, r4p, Fout, ldout, lt, partial, out, s);
}
// the 81-entry table of modes 3 / 4: lut[code][c] = sum_b (t_b - 1) W_b[c], code = sum_b t_b 3^b, W = eq((r1, r2), .)
void build_fold_lut(const H9 &r1, const H9 &r2, const BbHostRing &ring, fe *lut_host /* 2 * 81 * 9: values, then squares */) {
    H9 one;
    for (int i = 0; i < TAU; i++) one.c[i] = i == 0;
    H9 o1, o2;
    for (int i = 0; i < TAU; i++) { o1.c[i] = hsub(one.c[i], r1.c[i]); o2.c[i] = hsub(one.c[i], r2.c[i]); }
    H9 Wb[4] = {ring.mul9(o1, o2), ring.mul9(r1, o2), ring.mul9(o1, r2), ring.mul9(r1, r2)};
    for (int code = 0; code < 81; code++) {
        H9 val;
        for (int c = 0; c < TAU; c++) {
            u64 v = 0;
            int cc = code;
            for (int b = 0; b < 4; b++, cc /= 3) {
                if (cc % 3 == 2) v = hadd(v, Wb[b].c[c]);
                else if (cc % 3 == 0) v = hsub(v, Wb[b].c[c]);
            }
            val.c[c] = v;
            lut_host[code * TAU + c] = from_canon(v);
        }
        H9 sq = ring.mul9(val, val);
        for (int c = 0; c < TAU; c++) lut_host[(81 + code) * TAU + c] = from_canon(sq.c[c]);
    }
}
void launch_fold_round_fix(const DevBb &t, const FoldArgs &a, const fe *Fprev, size_t ldprev, const H9 &r, const BbHostRing &ring, fe *Fout, size_t ldout,
                           u32 K, const E9PreC *Mpre, i64 *partial, u64 *out, hipStream_t s) {
    FoldLut nl = {};
    launch_fold_round_impl(t, a, Fprev, ldprev, K, Mpre, 1, e9pre_from_h9(r, ring.T.nu), Fout, ldout, nl, partial, out, s);
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

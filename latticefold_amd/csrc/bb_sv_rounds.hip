// bb_sv_rounds.hip -- rounds 1..3 of the folding sumcheck of the BabyBear backend as exact int8 GEMMs (gfx950 v_mfma_i32_16x16x64_i8): the BabyBear
// form of lf_sv_rounds.hip (algebra: lf_sv_rounds.h; reference comb function nifs/folding/utils.rs:273-325, b = 2).  Before round i an f-hat entry is
// sum_b W_b y_b with ternary digits y_b, so the norm part of the round message is
//     sum_T mu_T sum_pi C_pi(X) (M0_pi[T] + X (M1_pi[T] - M0_pi[T])),    M^h_pi[T] = w_h M'_pi[T],    M'_pi[T] = sum_p E_i[p] sigma_pi(T, p) beta_pi(T, p)
// in the split eq form (eqB fixed at r_1..r_{i-1} is c_i eq(beta_i, h) E_i[p] at entry 2p + h, w_h = c_i eq(beta_i, h)).  The sums over the pairs p -- all the
// work -- are the GEMM: rows = the 16 digit planes of a (side, coefficient) group (2 x 72 groups), inner dimension = pairs, columns = the 9 x 4 balanced
// base-256 digits of the centred Montgomery words of E_i[p] (three column tiles, the last twelve columns zero).  The GEMM kernel itself is the Goldilocks
// one (lf_sv_rounds.hip: k_sv_gemm with 72 coefficient groups per side, launch_sv_gemm_tiles); this file holds what depends on the field:
//   k_bbsv_pack     E_i[9][.] -> digit bytes EB[48][padded pairs] in the slot order of the A operand
//   k_bbsv_finish1  per table T: M' from the int32 tiles (sum_u 256^u C_u mod p per word -- the words stay Montgomery words: the sums are linear),
//                   M0 / M1 = w_h M', the table's degree-4 polynomial sum_pi C_pi(X) (M0 + X (M1 - M0)), times mu_T (F_{p^9} products, bb_field.cuh)
//   k_bbsv_finish2  sum over the tables of a slot, evaluation at X = 0..4, plus the G part of the message (canonical words into mapped host memory)
//   k_bb_eq_pairsum E_{i+1}[p] = E_i[2p] + E_i[2p+1]
#include <hip/hip_runtime.h>
#include "bb_field.cuh"
#include "bb_kernels.h"
#include "lf_sv_rounds.h"

namespace lfbb {
static inline size_t cdiv(size_t a, size_t b) { return (a + b - 1) / b; }

// EB[(4 q + u)][slot] = balanced base-256 digit u of the centred Montgomery word E[q][pair(slot)]: the bytes of (x + 0x80808080) ^ 0x80808080 (|x| < 2^30)
__global__ void __launch_bounds__(256) k_bbsv_pack(const fe *E, size_t ld, size_t npairs, size_t ldeb, int V, unsigned char *EB) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, groups = ldeb / 16;
    if (gid >= groups * TAU) return;
    const u32 q = (u32)(gid / groups);
    const size_t p0 = (gid % groups) * 16;
    u32 w[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const size_t pr = p0 + (size_t)lf::sv_slot_pair_pub(V, t >> 2, t & 3);
        const fe x = pr < npairs ? E[(size_t)q * ld + pr] : 0;
        w[t] = ((u32)x + 0x80808080u) ^ 0x80808080u;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
        u32 o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 16; t++) o[t >> 2] |= ((w[t] >> (8 * u)) & 0xFFu) << (8 * (t & 3));
        *(uint4 *)(EB + (size_t)(4 * q + u) * ldeb + p0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__device__ __forceinline__ fe bbsv_red(i64 v) { return centre((int32_t)(v % (i64)BB_P)); }

// block = table T = (side, k, c), thread = digit monomial pi.  tp[(T * 5 + e) * 9 + q]: coefficient e of the table's polynomial (times mu_T), Montgomery words
template <bool NU2>
__global__ void __launch_bounds__(128) k_bbsv_finish1(fe nu, const int32_t *tot, u32 npr, u32 K, u32 ktiles, const fe *coef, const E9C *mu_c, fe *tp, E9C w0c, E9C w1c) {
    const u32 T = blockIdx.x, c = T % RE, k = (T / RE) % K, side = T / (RE * K);
    const u32 grp = (side * RE + c) * ktiles + k / 16, krow = k & 15;
    __shared__ int32_t sm[5 * TAU][128];
    const u32 pi = threadIdx.x;
    E9 P[5];
#pragma unroll
    for (int e = 0; e < 5; e++) P[e] = e9_zero();
    if (pi < npr) {
        const int32_t *base = tot + ((size_t)grp * npr + pi) * 768;
        auto cell = [&](u32 bp) { return (i64)base[(bp >> 4) * 256 + ((bp & 15) + 16 * (krow >> 2)) * 4 + (krow & 3)]; };
        E9 Mp, w0, w1;
#pragma unroll
        for (int q = 0; q < TAU; q++) {
            i64 v = 0;
#pragma unroll
            for (u32 u = 0; u < 4; u++) v += cell(4 * q + u) << (8 * u);
            Mp.c[q] = bbsv_red(v);
            w0.c[q] = w0c.c[q]; w1.c[q] = w1c.c[q];
        }
        const E9 M0 = e9_mul_t<NU2>(w0, Mp, nu), M1 = e9_mul_t<NU2>(w1, Mp, nu), dM = e9_sub(M1, M0);
        const E9 M0n = e9_times_nu_t<NU2>(M0, nu), dMn = e9_times_nu_t<NU2>(dM, nu);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            E9 C;
#pragma unroll
            for (int q = 0; q < TAU; q++) C.c[q] = coef[((size_t)pi * 4 + e) * TAU + q];
            P[e] = e9_add(P[e], e9_mul_pre(C, M0, M0n));
            P[e + 1] = e9_add(P[e + 1], e9_mul_pre(C, dM, dMn));
        }
    }
#pragma unroll
    for (int e = 0; e < 5; e++)
#pragma unroll
        for (int q = 0; q < TAU; q++) sm[TAU * e + q][threadIdx.x] = P[e].c[q];
    __syncthreads();
    if (threadIdx.x < 5 * TAU) {
        i64 s = 0;
        for (u32 i = 0; i < 128; i++) s += sm[threadIdx.x][i];
        sm[threadIdx.x][0] = bbsv_red(s);
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const u32 e = threadIdx.x;
        E9 mu, v;
#pragma unroll
        for (int q = 0; q < TAU; q++) { mu.c[q] = mu_c[(size_t)(side * K + k) * TAU + c / 8].c[q]; v.c[q] = sm[TAU * e + q][0]; }
        const E9 r = e9_mul_t<NU2>(mu, v, nu);
#pragma unroll
        for (int q = 0; q < TAU; q++) tp[((size_t)T * 5 + e) * TAU + q] = r.c[q];
    }
}
// block = (slot, q), wave e = coefficient e of the polynomial: its 64 lanes add the 2K * 9 tables (side, k, 8 d + slot), then five threads evaluate at X = 0..4:
// out[X * 72 + 9 * slot + q] = gpart[..] + sum_tables TP(X)   (canonical words).  (One block of 360 threads walking the 288 tables each took 65 us per round.)
__global__ void __launch_bounds__(320) k_bbsv_finish2(const fe *tp, u32 K, const u64 *gpart, u64 *out) {
    const u32 slot = blockIdx.x / TAU, q = blockIdx.x % TAU, e = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ i64 co[5];
    i64 acc = 0;
    for (u32 tb = lane; tb < 2 * K * (u32)TAU; tb += 64) {
        const u32 sk = tb / TAU, d = tb % TAU;
        acc += tp[((size_t)(sk * RE + 8 * d + slot) * 5 + e) * TAU + q];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down((long long)acc, off, 64);
    if (lane == 0) co[e] = acc;
    __syncthreads();
    if (threadIdx.x < 5) {
        const u32 X = threadIdx.x, i = X * RE + TAU * slot + q;
        const fe xm = from_small((int32_t)X);
        fe v = bbsv_red(co[4]);
        for (int k = 3; k >= 0; k--) v = fadd(fmul(v, xm), bbsv_red(co[k]));
        const u64 sres = gpart[i] % BB_P + to_canon(v);
        out[i] = sres >= BB_P ? sres - BB_P : sres;
    }
}
__global__ void __launch_bounds__(256) k_bb_eq_pairsum(const fe *in, size_t ldi, size_t nout, fe *out, size_t ldo) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nout * TAU) return;
    const size_t q = i / nout, p = i % nout;
    const int2 v = *(const int2 *)(in + q * ldi + 2 * p);
    out[q * ldo + p] = fadd(v.x, v.y);
}
void launch_bb_eq_pairsum(const fe *in, size_t ldi, size_t nout, fe *out, size_t ldo, hipStream_t s) {
    hipLaunchKernelGGL(k_bb_eq_pairsum, dim3((unsigned)cdiv(nout * TAU, 256)), dim3(256), 0, s, in, ldi, nout, out, ldo);
}

bool bbsv_shape_ok(int V, size_t npairs, u32 K) { return lf::sv_shape_ok(V, npairs, K); }
size_t bbsv_eb_bytes(size_t npairs) { return 48 * lf::sv_ldeb_pub(npairs) + 64; }
size_t bbsv_part_words(int V, u32 K) { return lf::sv_part_words_rd(RE, V, K); }
size_t bbsv_tot_words(int V, u32 K) { return lf::sv_tot_words_rd(RE, V, K); }
size_t bbsv_tp_words(u32 K) { return (size_t)2 * K * RE * 5 * TAU; }
size_t bbsv_bits_words(size_t n, u32 K) { return lf::sv_bits_words(n, K, RE); }
void launch_bbsv_bits(const int32_t *planes, size_t ldp, size_t n, u32 K, u32 *bits, hipStream_t s) { lf::launch_sv_bits(planes, ldp, n, K, bits, s, RE); }

// norm part of round log2(V) + 1 in the split eq form, added to the G part `gpart` (5 x 72 canonical words, device) -> out (5 x 72 canonical words).
// E: E_i[9][ldE] (one value per pair), npairs pairs; coef: [sv_num_pairs(V)][4][9] Montgomery words (device); mu_c: [2K][9] constants mu_k^(d+1);
// w0 / w1 = c_i eq(beta_i, 0 / 1).  Returns 0, or -1 when the shape is not handled (the caller keeps its VALU kernels).
int launch_bbsv_round(const DevBb &t, int V, const u32 *bitsL, const u32 *bitsR, size_t nplanes, const fe *E, size_t ldE, size_t npairs, u32 K, const E9C *mu_c,
                      const fe *coef, const E9C &w0, const E9C &w1, unsigned char *EB, int32_t *part, int32_t *tot, fe *tp, const u64 *gpart, u64 *out, hipStream_t s, hipEvent_t gpart_ready) {
    if (!bbsv_shape_ok(V, npairs, K)) return -1;
    const size_t ldeb = lf::sv_ldeb_pub(npairs);
    (void)hipMemsetAsync(EB + 36 * ldeb, 0, 12 * ldeb, s);     // digit columns 36..47: none
    hipLaunchKernelGGL(k_bbsv_pack, dim3((unsigned)cdiv(ldeb / 16 * TAU, 256)), dim3(256), 0, s, E, ldE, npairs, ldeb, V, EB);
    if (lf::launch_sv_gemm_tiles(RE, V, bitsL, bitsR, nplanes, EB, ldeb, npairs, K, part, tot, s) != 0) return -1;
    const u32 npr = (u32)lf::sv_num_pairs(V), ktiles = (K + 15) / 16;
    if (t.nu == BB_TWO) hipLaunchKernelGGL((k_bbsv_finish1<true>), dim3(2 * K * RE), dim3(128), 0, s, t.nu, tot, npr, K, ktiles, coef, mu_c, tp, w0, w1);
    else hipLaunchKernelGGL((k_bbsv_finish1<false>), dim3(2 * K * RE), dim3(128), 0, s, t.nu, tot, npr, K, ktiles, coef, mu_c, tp, w0, w1);
    if (gpart_ready) (void)hipStreamWaitEvent(s, gpart_ready, 0);   // the G part was computed on another stream
    hipLaunchKernelGGL(k_bbsv_finish2, dim3(8 * TAU), dim3(320), 0, s, tp, K, gpart, out);
    return 0;
}
}  // namespace lfbb

// bb_rounds.hip -- BabyBear backend: the sumcheck round kernels of the fold step (gfx950, wave64) -- linearization rounds (generic comb function and the
// R1CS shape, small rounds fused with fix_variables), folding rounds in all their forms (rounds 1-2 from the planes, look-up-table rounds, table rounds with
// fused fix_variables, the two-lane round 5).  Split off bb_kernels.hip in round 6 (no source file above 2 000 lines).
#include "bb_kernels.h"

#include <stdlib.h>

#include "bb_kernels_dev.cuh"

namespace lfbb {

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
    if (mode <= 1 && pairs <= 128)
        while (a2.qsplit < 16 && pairs * a2.qsplit * 2 <= 256) a2.qsplit *= 2;
    // enough threads to fill the chip (~128k): split the 2K*9 tables when there are few pairs
    u32 tch = 1;
    constexpr size_t chunk_threads = (size_t)1 << 17;
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
    if (!Esp && Fout) {     // (unsplit callers: two lanes per pair, k_fold_round5_2l)
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


}  // namespace lfbb

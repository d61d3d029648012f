// lf_rounds.hip -- the sumcheck round kernels of the fold step (gfx950, wave64): linearization rounds (sumcheck/prover.rs:56-162 with the comb function of
// nifs/linearization/utils.rs:90-107), folding rounds in all their forms (nifs/folding/utils.rs:273-325: rounds 1-2 from the planes, look-up-table rounds 3-4,
// product-free table rounds 4-6, materialised table rounds with fused fix_variables) and the persistent tails (k_lin_tail, k_fold_tail: every round below 2048
// entries in ONE launch, messages / challenges through a host-mapped mailbox).  Split off lf_kernels.hip in round 6 (no source file above 2 000 lines).
#include "lf_kernels.h"

#include <stdlib.h>

#include "lf_kernels_dev.cuh"

namespace lf {

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
    constexpr size_t v = 131072;   // measured at C4: 65536 / 131072 / 262144 / 524288 -> 19.58 / 19.28 / 19.35 / 19.74 ms per step
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

}  // namespace lf

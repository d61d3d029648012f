// lfp_kernels.h -- LatticeFold+ slice on the Frog ring Z_p[X]/(X^16 + 1), p = 15912092521325583641, COEFFICIENT form
// (the ring the reference runs latticefold-plus on: cyclotomic-rings/src/rings/frog.rs, crates/latticefold-plus/src/rgchk.rs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace lfp {
typedef uint64_t u64;
typedef uint32_t u32;
constexpr u64 P = 15912092521325583641ull;
constexpr int D = 16;
constexpr int TJ = 32;        // witness rows per LDS tile
constexpr int IG = 4;         // commitment-matrix rows per launch group
constexpr int KG = 4;         // digit planes per launch group
constexpr int RED_WAVES = 16;  // waves per block of the partial-sum reduction

struct Phase1Args {
    const u64 *f;       // n x 16 canonical words
    const u64 *A;       // kappa x n x 16
    u64 n;
    u32 kappa, i0, icnt;
    u32 k, k0, kcnt;
    u64 b;              // digit base
    int sh;             // log2(b) when b is a power of two, else -1
    u32 J;              // rows per block (multiple of TJ)
    int8_t *Df;         // [k][n][16], written when write_df
    u64 *part;          // [nblk][k*kappa*256 + kappa*16] partial sums mod p: comM_f coefficients, then A f
    u32 *err;
    int write_df, do_f;
};
struct Phase2Args {
    const u64 *A;
    const u64 *tau;     // n canonical words
    u64 n;
    u32 kappa, i0, icnt;
    u32 J;
    int8_t *mtau;       // n exponents (centred tau), written when i0 == 0
    u64 *part;          // [nblk][2*kappa*16] partial sums mod p: A tau, then A exp(tau)
    u32 *err;
};
void launch_phase1(const Phase1Args &a, u32 nblk, hipStream_t s);
void launch_phase2(const Phase2Args &a, u32 nblk, hipStream_t s);
void launch_mtau_all(const u64 *tau, u64 n, int8_t *mtau, u32 *err, hipStream_t s);
// out[o] = sum over blocks of part[blk][o] mod p; with l != 0 the first nsplit outputs (comM_f, [k][kappa][16][16]) are also cut into
// their l gadget digits: tau = split(hconcat(comM_f), n, base, l) (utils.rs:12-43), positions [0, kappa*k*16*l*16)
void launch_reduce(const u64 *part, u32 nblk, u32 nout, u64 *out, u32 nsplit, u32 kappa, u32 k, u64 base, u32 l, u64 *tau, hipStream_t s);
// largest launch group (1, 2 or 4) not above `left`
inline u32 group_size(u32 left) { return left >= 4 ? 4 : left >= 2 ? 2 : 1; }
// Decomp::decompose (decomp.rs:32-99): base-B split, fix_variables over ring tables, sparse mat-vec with ring coefficients
void launch_decompose2(const u64 *f, size_t words, u64 B, u64 *F0, u64 *F1, hipStream_t s);
void launch_ring_fix(const u64 *in, u64 *out, u32 ntab, size_t len, const u64 *rM /* [2][16] Montgomery */, hipStream_t s, int const_r = 0);   // const_r: both coordinates are constant polynomials
void launch_spmv_ring(const u32 *rowptr, const u32 *col, const u64 *valM, const u64 *x, size_t nrows, u64 *y, hipStream_t s, int const_coef = 0);   // const_coef: every coefficient is a constant polynomial AND valM holds the constant terms alone, one word per non-zero (LfpMatrix::spmv_vals)
void launch_replicate(const u64 *src, size_t words, u32 copies, u64 *dst, hipStream_t s);
// ---- set check / range check (lfp_rgchk.hip; setchk.rs:65-262, rgchk.rs:81-186) ----------------------------------------------------------
constexpr int8_t LFP_ABSENT = -128;   // a zero entry of a monomial set (an absent sparse-matrix coefficient)
struct PwTab { u64 p[16], q[16]; };   // beta^t and beta^(2t), Montgomery
struct EqPt { u64 c[32], nc[32], one; };   // c_j, 1 - c_j and 1, Montgomery
struct ScDesc { u32 nmat, ncols, nvec, nsets_eff; };   // nsets_eff: sets entering the sumcheck polynomial (setchk.rs:160-197: all, or the first matrix set alone without a batching challenge)
void launch_sc_tables(const int8_t *dig, size_t n, u32 ncols, const PwTab &pw, u64 *tab, size_t ld, hipStream_t s);
void launch_eq_build(const EqPt &pt, u32 nv, u64 *eq, hipStream_t s);
u32 sc_round_max_blocks();
u32 launch_sc_round(const u64 *tab, size_t ld, size_t half, const ScDesc &d, const u64 *coef, u64 *part /* sc_round_max_blocks * 4; returns the blocks used */, hipStream_t s);
void launch_sc_fix(const u64 *in, u64 *out, size_t ld, u32 ntab, size_t half, u64 rM, hipStream_t s);
// the same round / fix + round from the exponent digits of ONE set (ncols 16 or 1; returns the blocks used, 0 = shape not handled): no beta^e tables exist
u32 launch_sc_round0_dig(const int8_t *dig, size_t n, u32 ncols, const PwTab &pw, const u64 *eq, const u64 *coef, u64 *part, hipStream_t s);
u32 launch_sc_fix_round_dig(const int8_t *dig, size_t n, u32 ncols, const PwTab &pw, const u64 *eq, u64 rM, u64 *tab /* the set's 2 ncols + 1 tables */, size_t ld,
                            const u64 *coef, u64 *part, hipStream_t s);
u32 eval_chunks(size_t n);
void launch_sum_parts(const u64 *part, u32 chunks, size_t stride, u32 nout, u64 *out, hipStream_t s);      // out[o] = sum_chunk part[chunk * stride + o]
// out[col][16] = sum_row w[row] X^e(dig[row][col]); wstride 1: scalar Montgomery weights (out canonical), 16: canonical ring weights.  part: eval_chunks(n) * ncols * 16
void launch_wmono(const int8_t *dig, size_t n, u32 ncols, const u64 *w, u32 wstride, u64 *part, u64 *out, hipStream_t s);
// 16 columns against nw <= 4 SCALAR weight tables (Montgomery) in one pass (exponent histogram): out[q * ostride + col * 16 + t]; part: nw * eval_chunks(n) * 256 words
void launch_whist16(const int8_t *dig, size_t n, const u64 *const *w, u32 nw, u64 *part, u64 *out, size_t ostride, hipStream_t s);
// out[16] = sum_row w[row] f[row]; part: eval_chunks(n) * 16
void launch_wring(const u64 *f, size_t n, const u64 *w, u32 wstride, u64 *part, u64 *out, hipStream_t s);
// out[0] = sum_i x[i xstride] y[i]; part: eval_chunks(n) * 4
void launch_wdot(const u64 *x, u32 xstride, int x_mont, const u64 *y, size_t n, u64 *part, u64 *out, hipStream_t s);
void launch_spmvT_eq(const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t n, u64 *w, hipStream_t s);
void launch_spmvT_eq_const(const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t n, u64 *w, hipStream_t s);   // scalar w (Montgomery): constant-coefficient M
// ---- Cm::prove (cm.rs:56-347)
struct CmShort { int32_t v[3][16]; };     // the three folding challenges s (centred coefficients)
struct CmDesc { u32 L, nM; };
void launch_cm_materialize(const int8_t *dig, const u64 *tau, size_t n, u64 *out, hipStream_t s);
// compact instance tables of Cm::prove: m_tau as exponent bytes, M_q tau as scalars (lfp_rgchk.hip)
struct CmCompact { const int8_t *mtau[8]; const u64 *mts; size_t ldm; };      // mts[(l nM + q) ldm + row], canonical
struct CmTabList { uint16_t idx[64]; };
void launch_spmv_scalar_const(const u32 *rowptr, const u32 *col, const u64 *valM, const u64 *x, size_t nrows, u64 *y, hipStream_t s);
void launch_spmv_mono_const(const u32 *rowptr, const u32 *col, const u64 *valM, const int8_t *dig, size_t nrows, u64 *y, hipStream_t s);
void launch_to_mont(const u64 *in, size_t n, u64 *out, hipStream_t s);
void launch_cm_h(const int8_t *Df, size_t n, u32 k, const int32_t *sp, u64 *h, hipStream_t s);
void launch_cm_g(const u64 *tau, const int8_t *mtau, const u64 *f, const u64 *h, size_t n, const CmShort &s, u64 *g, hipStream_t st);
// the round with fix_variables of the previous round fused in (S / R: the previous tables, 4 entries per new pair; the fixed tables go to So / Ro, ld_o entries per table)
void launch_cm_round_fused(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, const CmDesc &d, const u64 *rcp, u64 rM, u64 *So, u64 *Ro, size_t ld_o, u64 *part,
                           hipStream_t s);
u32 cm_round_blocks(size_t half);
void launch_cm_round(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, const CmDesc &d, const u64 *rcp, u64 *part /* blocks * 48 */, hipStream_t s);
void launch_cm_fix(const u64 *in, size_t ld_in, u64 *out, size_t ld_out, u32 w, u32 ntab, size_t half, u64 rM, hipStream_t s);
// the sumcheckers through batched tables (S2 = eq | V, R2 = U | Z: lfp_rgchk.hip); evaluations of ring tables at the point of `eq` (part: cm_eval_chunks(n) * ntab * 16)
void launch_cm_combine(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t n, const CmDesc &d, const u64 *rcp, u64 *S2, u64 *R2, size_t ld2, hipStream_t s);
void launch_cm2_round(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, u64 *part /* blocks * 48 */, hipStream_t s);
void launch_cm2_round_fused(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, u64 rM, u64 *So, u64 *Ro, size_t ld_o, u64 *part, hipStream_t s);
u32 cm_eval_chunks(size_t n);
void launch_cm_evals(const u64 *R, size_t ldr, size_t n, const u64 *eq, u32 ntab, u64 *part, u64 *out, hipStream_t s);
void launch_cm_combine_c(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t n, const CmDesc &d, const u64 *rcp, const CmCompact &cc, u64 *S2, u64 *R2, size_t ld2, hipStream_t s);
void launch_cm_evals_c(const u64 *R, size_t ldr, size_t n, const u64 *eq, u32 ntab, const CmTabList &dense, u32 ndense, const CmCompact &cc, u32 L, u32 nM, u32 per, u64 *part,
                       u64 *out, hipStream_t s);
// ---- ComR1CS::linearize (r1cs.rs:76-139): degree-3 round of eq (ga gb - gc); part: cm_round_blocks(half) * 64
void launch_r1cs_round_fused(const u64 *E, const u64 *G, size_t ld, size_t half, u64 rM, u64 *Eo, u64 *Go, size_t ld_o, u64 *part, hipStream_t s);   // + fix_variables of the previous round (E / G: previous tables)
void launch_r1cs_round(const u64 *E, const u64 *G, size_t ld, size_t half, u64 *part, hipStream_t s);
void launch_vec_add(u64 *acc, const u64 *x, size_t n, hipStream_t s);
void launch_check_canonical(const u64 *x /* 16-byte aligned */, size_t n, u32 *flag, u32 bit, hipStream_t s);
void launch_tensor_level(const u64 *cur, u64 len, u64 r, u64 *nxt, hipStream_t s);
void launch_tensor_product(const u64 *a, u64 m, const u64 *b, u64 n, u64 *out, hipStream_t s);
}  // namespace lfp

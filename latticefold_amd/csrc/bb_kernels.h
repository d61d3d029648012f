// bb_kernels.h -- launchers of the gfx950 kernels of the BabyBearRingNTT backend (bb_kernels.hip).  All pointers are
// DEVICE pointers.  Words are centred Montgomery int32 (bb_field.cuh).
//
// Device layouts (DESIGN.md "BabyBear backend"):
//   ring table   : fe  [72][n]   plane w = 9*slot + coord (NTT form) or coefficient index (coefficient form)
//   fq9 table    : fe  [9][n]    slot-constant values (eq tables)
//   coef planes  : i32 [72][n]   centred integer coefficients of a B-short witness (NOT Montgomery)
//   Ajtai matrix : fe  [kappa][72][n]
#pragma once
#include <hip/hip_runtime.h>

#include "bb_host.h"

namespace lfbb {

struct DevBb {   // compact device copy of BbTables, passed by value
    fe nu;
    fe w4, w2, w10, w1, w7, w5, w11;
    int slot_of_pos[8];
    int pos[TAU][8];
    fe tw[TAU][8];
};
DevBb make_dev_bb(const BbTables &T);
struct E9C { fe c[TAU]; };                 // kernel-argument constant
struct E9PreC { fe v[TAU], vn[TAU]; };     // constant with its nu-multiple
E9C e9c_from_h9(const H9 &h);
E9PreC e9pre_from_h9(const H9 &h, u64 nu);

// ---- layout / utility ------------------------------------------------------------------------------------------
void launch_aos_to_soa(const u64 *aos_canon, fe *soa, size_t n, hipStream_t s);     // [n][72] canonical u64 -> [72][n] fe
void launch_soa_to_aos(const fe *soa, u64 *aos_canon, size_t n, hipStream_t s);
void launch_fill_ajtai(fe *A, u32 kappa, size_t n, size_t n_total, size_t col0, u64 seed, hipStream_t s);
void launch_selftest(const u64 *in_canon /*[n][18]*/, u64 *out_canon /*[n][12]*/, u32 n, fe nu, hipStream_t s);
// i64 partial sums [nblocks][nv] -> canonical u64 [nv]
void launch_reduce_rows(const i64 *partial, u32 nblocks, u32 nv, u64 *out_canon, hipStream_t s);

// ---- CRT / ICRT ------------------------------------------------------------------------------------------------
void launch_crt_fwd(const DevBb &t, const fe *coef, fe *ntt, size_t n, hipStream_t s);
void launch_icrt_dense(const fe *icrt_mat /*72*72*/, const fe *ntt, fe *coef, size_t n, hipStream_t s);
// digit pass of a general commitment (lf_ajtai_i8g.hip) from the NTT form f [72][ld]: inverse CRT map (dense, or its compressed rows sp_val / sp_col [72][8] when
// no row has more than 8 entries) -> centred residues -> NP balanced base-128 digit words pre [NP][72][ldw]
void launch_i8g_cut_ntt(const fe *icrt_mat, const fe *sp_val, const u32 *sp_col, const fe *ntt, size_t ld, size_t n, u32 NP, unsigned long long *pre, size_t ldw,
                        hipStream_t s);

// ---- decomposition ---------------------------------------------------------------------------------------------
void launch_decompose(const fe *coef, size_t n, u64 base, u32 digits, int layout, fe *out, hipStream_t s, int mode = 0);
void launch_recompose(const fe *in, size_t n_out, u64 base, u32 digits, fe *out, hipStream_t s);
void launch_coef_to_i32(const fe *coef, int32_t *planes, size_t n, u32 bound, int *viol, hipStream_t s);
void launch_i32_to_coef(const int32_t *planes, fe *coef, size_t n, hipStream_t s);
void launch_recompose_crt(const DevBb &t, const int32_t *planes, size_t n_planes, u32 wit_len, u32 L, u64 B, u32 K, int mode_bits,
                          fe *out, size_t ldz, size_t off, hipStream_t s);
void launch_linf(const fe *coef, size_t n, u64 *out_max, hipStream_t s);

// ---- Ajtai -----------------------------------------------------------------------------------------------------
// out: canonical u64 AoS [batch][kappa][72]

// ---- MLE / eq ----------------------------------------------------------------------------------------------------
void launch_build_eq(const DevBb &t, const E9PreC *r_dev /*nv*/, const E9PreC *omr_dev /*nv: 1-r*/, u32 nv, fe *eq, hipStream_t s);
void launch_spmv(const DevBb &t, const u32 *rowptr, const u32 *col, const fe *val /*[nnz][72]*/, const fe *z, size_t ldz, fe *out,
                 size_t m, int accumulate, hipStream_t s);
// out = sum_{j<nm} M_j z_j (nm <= 4), z_j = z + j*z_stride: one launch, one write of out
void launch_spmv_sum(const DevBb &t, u32 nm, const u32 *const *rowptr, const u32 *const *col, const fe *const *val, const fe *z, size_t z_stride,
                     size_t ldz, fe *out, size_t m, hipStream_t s);
void launch_spmv_t_eq(const DevBb &t, const u32 *colptr, const u32 *rowidx, const fe *val, const fe *eq, size_t m, fe *q, size_t n,
                      hipStream_t s);
size_t red_partial_words(u32 nv);
// the same on the int8 matrix cores (bb_dot_i8.hip), na <= 16, nb <= 3; scratch: YB bbdot_i8_yb_bytes(n), part bbdot_i8_part_words(n) int32,
// tot bbdot_i8_tot_words() int64.  Returns 0, or -1 if the shape is not handled.
// T[k][c] = sum_i eq[i] digit_k(planes[c][i]) of the K <= 16 binary digit planes on the matrix cores (bb_dot_i8.hip); scratch: EB coef_eval_i8_eb_bytes(n) (16-byte
// aligned), part coef_eval_i8_part_words(nwg) int32, tot coef_eval_i8_tot_words() int64.  out as launch_coef_eval (mode_bits).  0, or -1 if the shape is not handled.
size_t coef_eval_i8_eb_bytes(size_t n);
size_t coef_eval_i8_part_words(u32 nwg);
size_t coef_eval_i8_tot_words();
int launch_coef_eval_i8(const int32_t *planes, size_t ldp, size_t n, const fe *eq, size_t ldeq, u32 K, unsigned char *EB, u32 nwg, int32_t *part, long long *tot,
                        u64 *out, hipStream_t s);
size_t bbdot_i8_yb_bytes(size_t n);
size_t bbdot_i8_part_words(size_t n);
size_t bbdot_i8_tot_words();
int launch_dot_batch_i8(const DevBb &t, const fe *X, size_t ldx, u32 na, const fe *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, int32_t *part,
                        long long *tot, u64 *out, hipStream_t s, bool y_packed = false);
int launch_dot_pack_y(const fe *X, const fe *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, hipStream_t s);
void launch_dot_batch(const DevBb &t, const fe *X, size_t ldx, u32 na, const fe *Y, size_t ldy, u32 nb, size_t n, i64 *partial,
                      u64 *out /*[na][nb][72] canonical*/, hipStream_t s);
void launch_dot_eq(const DevBb &t, const fe *X, size_t ldx, u32 na, const fe *eq, size_t ldeq, size_t n, i64 *partial,
                   u64 *out /*[na][72]*/, hipStream_t s);
void launch_vs_combine(const u64 *vs /* [K][nv] canonical */, u32 K, u32 nv, u64 *v, hipStream_t s);   // v = sum_k 2^k v_s[k]
// T[k][c] = sum_i eq[i] * digit_k(planes[c][i]) (mode_bits) or the full value (K = 1): out canonical [K][72][9]
void launch_coef_eval(const DevBb &t, const int32_t *planes, size_t n, const fe *eq, size_t ldeq, u32 K, int mode_bits, i64 *partial,
                      u64 *out, hipStream_t s);
// per_slot != 0: coef_dev holds K*tt*8 constants, one per slot (ring-element coefficients)
void launch_lincomb_z(const DevBb &t, const fe *z, size_t ldz, u32 K, const E9PreC *coef_dev /*K*tt*/, u32 tt, size_t n, fe *out,
                      hipStream_t s, u32 per_slot = 0);
void launch_add_fhat_comb(const DevBb &t, const int32_t *planes, size_t n_planes, u32 K, const E9C *apow_dev /*K*9*/, fe *G, size_t m,
                          hipStream_t s);

// ---- sumcheck ------------------------------------------------------------------------------------------------------
// rows9 groups of 9 planes: out[g][c][j] = in[g][c][2j] + r * (in[g][c][2j+1] - in[g][c][2j])
void launch_fix(const DevBb &t, const fe *in, size_t ld_in, fe *out, size_t ld_out, size_t n_in, u32 rows9, const E9PreC &r, hipStream_t s);
void launch_fix_final(const DevBb &t, const fe *in, size_t ld_in, u32 rows9, const E9PreC &r, u64 *out /* [rows9][9] canonical */, hipStream_t s);

struct LinDesc {   // CCS multiset structure (nifs/linearization/utils.rs:90-107)
    u32 t, q;
    u32 S_off[9];
    u32 S_idx[16];
    fe c[8][RE];     // coefficients c_i, NTT form, Montgomery
    int c_unit[8];   // +1 / -1 when c_i = +-1
};
void launch_lin_round(const DevBb &t, const LinDesc &desc, const fe *mz, size_t ld, const fe *eq, size_t ldeq, size_t n, u32 deg,
                      i64 *partial, u64 *out, hipStream_t s, u32 max_blocks = 0);
// the R1CS shape (+ Mz_0 Mz_1 - Mz_2) in its own kernel; r != nullptr: fix_variables of the previous round fused in (mz / eq are then the previous tables, 4 * pairs
// entries, and the fixed ones go to mz_out / eq_out); at most 256 pairs: the message goes straight to `out`, no reduction launch
bool lin_desc_is_r1cs(const LinDesc &d);
void launch_lin_r1cs(const DevBb &t, const fe *mz, size_t ld, const fe *eq, size_t ldeq, size_t pairs, const E9PreC *r, fe *mz_out, size_t ld_out, fe *eq_out, size_t ldeq_out,
                     i64 *partial, u64 *out, hipStream_t s, u32 max_blocks = 0);
// a small round (n_prev / 4 <= 256 pairs) in one launch: fix of the previous tables with r (-> mz_out / eq_out, n_prev / 2 entries), evaluation, reduction; message to `out`
void launch_lin_small(const DevBb &t, const LinDesc &desc, const fe *mz_prev, size_t ld_prev, const fe *eq_prev, size_t ldeq_prev, size_t n_prev, const E9PreC &r, fe *mz_out, size_t ld_out,
                      fe *eq_out, size_t ldeq_out, u32 deg, u64 *out, hipStream_t s);

struct FoldArgs {
    const fe *eqL, *eqR, *eqB;   // fq9 tables [9][ld]
    const fe *G1, *G2;           // ring tables [72][ld]
    size_t ld, n;                // leading dimension / current length
    size_t p0, pcnt;             // pair range handled by this launch (all pairs: 0, n/2; a rank's slice when sharded)
    size_t pF0;                  // first pair held by the materialised f-hat buffer (general rounds)
    u32 qsplit = 1;              // k_fold_round modes 0 / 1, small rounds: threads per pair (a power of two <= 16): thread q of a pair takes the tables tb0 + q, + qsplit, ..
};
// the G part of a round message alone (eqL G1 + eqR G2 at X = 0..4); partial: red_partial_words(5 * RE)
void launch_fold_round_g(const DevBb &t, const FoldArgs &a, i64 *partial, u64 *out, hipStream_t s);
// round 1 straight from the coefficient planes (f-hat virtual, b = 2); Mc = mu_k^(d+1), [2K][9] constants
void launch_fold_round1(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                        const E9C *Mc_dev, i64 *partial, u64 *out, hipStream_t s);
// round 2, still from the planes (entries a + b r1 with small integers a, b); a.* are the once-fixed tables (a.n = m/2)
void launch_fold_round2(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                        const E9C *Mc_dev, const H9 &r1, const BbHostRing &ring, i64 *partial, u64 *out, hipStream_t s);
// after r_2: F[(k*9+d)][72][m/4] = sum_b eq((r1,r2), b) * digit(f[4j+b])
// entries j0 <= j < j0 + q of the m/4-entry tables, stored with leading dimension q
void launch_fold_materialize2(const DevBb &t, const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t j0, size_t q, u32 K,
                              const H9 &r1, const H9 &r2, const BbHostRing &ring, fe *F, hipStream_t s);
size_t fold_partial_words(size_t m);   // i64 words of `partial` the general rounds need
void launch_fold_round(const DevBb &t, const FoldArgs &a, const fe *F, size_t ldF, u32 K, const E9PreC *Mpre_dev, i64 *partial, u64 *out,
                       hipStream_t s);
// the same round with fix_variables of the previous round's tables fused in (unsharded large rounds): reads entries 4j..4j+3
// of Fprev, stores the fixed pair to Fout for the next round
void launch_fold_round_fix(const DevBb &t, const FoldArgs &a, const fe *Fprev, size_t ldprev, const H9 &r, const BbHostRing &ring, fe *Fout,
                           size_t ldout, u32 K, const E9PreC *Mpre, i64 *partial, u64 *out, hipStream_t s);
// rounds 3 / 4 straight from the coefficient planes through the 81-entry digit look-up table (build_fold_lut, device copy lut_dev):
// round 3 touches no table, round 4 fixes with r3 and writes the m/8-entry tables
void build_fold_lut(const H9 &r1, const H9 &r2, const BbHostRing &ring, fe *lut_host /* 81 * 9 */);
// round 3 with per-table products M_tb * {value, value^2, value^3} of the look-up values (mutab_dev: 3 * 2K*9 * 81 * 12 words, filled here)
void launch_fold_round_lut_mu(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                              fe *mutab_dev, u32 K, const E9PreC *Mpre, i64 *partial, u64 *out, hipStream_t s);
void launch_fold_round_lut_fix(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                               const H9 &r, const BbHostRing &ring, fe *Fout, size_t ldout, u32 K, const E9PreC *Mpre, i64 *partial, u64 *out,
                               hipStream_t s);
void launch_fold_round_lut_fix_tab(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                                   const H9 &r, const BbHostRing &ring, fe *sq_dev, fe *mt_dev, fe *Fout, size_t ldout, u32 K, const E9PreC *Mpre, i64 *partial,
                                   u64 *out, hipStream_t s, const fe *Esp = nullptr, size_t ldEsp = 0);   // Esp: split form (k_fold_round SPLIT), E_i as [9][ldEsp]
void launch_fold_round_lut_fix5(const DevBb &t, const FoldArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, const fe *lut_dev,
                                const H9 &r3, const H9 &r4, const BbHostRing &ring, fe *xx_dev, fe *yy_dev, fe *mt_dev, fe *Fout, size_t ldout, u32 K,
                                const E9PreC *Mpre, i64 *partial, u64 *out, hipStream_t s, const fe *Esp = nullptr, size_t ldEsp = 0);
// folded witness in the coefficient domain: out[c][j] = sum_{i<2K} (rho_i * bitplane_i)(c) mod X^72 - X^36 + 1
void launch_fold_witness(const int32_t *planesL, const int32_t *planesR, size_t n, u32 K, const int8_t *rho_dev /*[2K][24]*/, int32_t *out,
                         hipStream_t s);

// ---- rounds 1..3 of the folding sumcheck as int8 GEMMs (bb_sv_rounds.hip) ------------------------------------------------------------------------
bool bbsv_shape_ok(int V, size_t npairs, u32 K);
size_t bbsv_eb_bytes(size_t npairs);
size_t bbsv_part_words(int V, u32 K);
size_t bbsv_tot_words(int V, u32 K);
size_t bbsv_tp_words(u32 K);
size_t bbsv_bits_words(size_t n, u32 K);
void launch_bbsv_bits(const int32_t *planes, size_t ldp, size_t n, u32 K, u32 *bits, hipStream_t s);
void launch_bb_eq_pairsum(const fe *in, size_t ldi, size_t nout, fe *out, size_t ldo, hipStream_t s);
int launch_bbsv_round(const DevBb &t, int V, const u32 *bitsL, const u32 *bitsR, size_t nplanes, const fe *E, size_t ldE, size_t npairs, u32 K, const E9C *mu_c,
                      const fe *coef, const E9C &w0, const E9C &w1, unsigned char *EB, int32_t *part, int32_t *tot, fe *tp, const u64 *gpart, u64 *out, hipStream_t s, hipEvent_t gpart_ready = nullptr);
}  // namespace lfbb

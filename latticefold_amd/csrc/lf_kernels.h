// lf_kernels.h -- launchers of the gfx950 kernels (lf_kernels.hip).  All pointers are DEVICE pointers.
//
// Device layouts (DESIGN.md "HBM layout"):
//   ring table   : u64 [24][n]   plane w = 3*slot + coord (NTT form) or coefficient index (coefficient form)
//   fq3 table    : u64 [3][n]    slot-constant values (eq tables)
//   coef planes  : i32 [24][n]   centred coefficients of a B-short witness (|v| <= B/2 <= 2^15)
//   Ajtai matrix : u64 [kappa][24][n]
#pragma once
#include "lf_ajtai_i8.h"
#include "lf_sv_rounds.h"
#include <hip/hip_runtime.h>

#include "lf_host.h"

namespace lf {

struct DevCrt {  // compact copy of CrtTables passed by value as a kernel argument
    u64 nu;
    int nu2p40;
    u64 w4, w2, w10, w1, w7, w5, w11;
    int slot_of_pos[8], pos1[8], pos2[8];
    u64 tw1[8], tw2[8];
};
DevCrt make_dev_crt(const CrtTables &T);

struct Fq3Const { u64 c[3]; };

// ---- layout / utility --------------------------------------------------------------------------------------
void launch_aos_to_soa(const u64 *aos, u64 *soa, size_t n, hipStream_t s);   // [n][24] -> [24][n]
void launch_soa_to_aos(const u64 *soa, u64 *aos, size_t n, hipStream_t s);
void launch_fill_uniform(u64 *dst, size_t words, u64 seed, size_t start, hipStream_t s);  // SplitMix64 stream (workload.py)
// Ajtai matrix generated in place in plane layout, equal to AoS stream splitmix(seed)[((i*n+j)*24+w)]
// columns [col0, col0+n) of the n_total-column matrix
void launch_fill_ajtai(u64 *A, u32 kappa, size_t n, size_t n_total, size_t col0, u64 seed, hipStream_t s, u32 row0 = 0);

// sharded exchanges: modular sum of all-gathered partial vectors; re-layout of all-gathered table slices
void launch_modsum(const u64 *parts, u32 nparts, size_t words, u64 *out, hipStream_t s);
void launch_gather_relayout(const u64 *all, u32 nranks, size_t planes, size_t lcl, u64 *full, hipStream_t s);
void launch_gather_relayout_part(const u64 *all, u32 nranks, size_t planes_tot, size_t p0, size_t planes, size_t lcl, u64 *full, hipStream_t s);

void launch_selftest_field(u64 seed, u32 n, u64 *mism_dev, hipStream_t s);  // arithmetic self-test, see lf_kernels.hip

// ---- CRT / ICRT (a1, a2) ---------------------------------------------------------------------------------------
void launch_crt_fwd(const DevCrt &t, const u64 *coef, u64 *ntt, size_t n, hipStream_t s);
void launch_icrt_dense(const u64 *icrt_mat /*24*24 dev*/, const u64 *ntt, u64 *coef, size_t n, hipStream_t s);

// ---- digit-plane commitments on the int8 matrix cores: lf_ajtai_i8.h

// ---- decomposition (a3) ----------------------------------------------------------------------------------------
// generic balanced digits on canonical coefficient tables: out has `digits` tables; layout 0: element i ->
// out index i*digits+k (n_out = n*digits), layout 1: table k at out + k*24*n
// mode: balanced-digit rule (0 = sign-magnitude truncation, ties kept; 1 = floor rule, digits in [-base/2, base/2); base 2 always 0)
void launch_decompose(const u64 *coef, size_t n, u64 base, u32 digits, int layout, u64 *out, hipStream_t s, int mode = 0);
void launch_recompose(const u64 *in, size_t n_out, u64 base, u32 digits, u64 *out, hipStream_t s);
// canonical coefficients -> centred i32 planes; *viol is set to 1 if some |v| > bound
void launch_coef_to_i32(const u64 *coef, int32_t *planes, size_t n, u32 bound, int *viol, hipStream_t s);
void launch_i32_to_coef(const int32_t *planes, u64 *coef, size_t n, hipStream_t s);
// NTT of bit-plane k (k0 <= k < k1) of every element: out[(k-k0)][24][n]
// z-vector tails: out_k[off + i] = CRT( sum_l B^l * digit_k(planes[i*L + l]) ) for k < K (mode_bits = 1), or the
// full value (mode_bits = 0, K = 1).  out_k = out + k*24*ldz.
void launch_recompose_crt(const DevCrt &t, const int32_t *planes, size_t n_planes, u32 wit_len, u32 L, u64 B, u32 K,
                          int mode_bits, u64 *out, size_t ldz, size_t off, hipStream_t s);
// l-infinity norm (a15): max |centred coefficient| of canonical coefficient table -> *out_max (u64)
void launch_linf(const u64 *coef, size_t n, u64 *out_max, hipStream_t s);

// ---- Ajtai commit (a5) -----------------------------------------------------------------------------------------
// partial[split][slot][i][k][3]; then reduce -> out AoS-ish [k][i][24] (device), canonical
// row-chunked commits (kappa > 48): tmp [batch][kc][24] -> out [batch][kappa][24] at row i0

// ---- MLE / eq (a8, a9, a11) --------------------------------------------------------------------------------------
void launch_build_eq(const DevCrt &t, const Fq3Const *r_dev /*nv*/, u32 nv, u64 *eq /*[3][1<<nv]*/, hipStream_t s);
// the same table as an outer product of two half-size tables (scratch: build_eq_scratch_words(nv) words)
size_t build_eq_scratch_words(u32 nv);
void launch_build_eq2(const DevCrt &t, const Fq3Const *r_dev, u32 nv, u64 *scratch, u64 *eq, hipStream_t s);
// sparse mat-vec (a7): CSR rows m; z ring table [24][n]; out ring table [24][m]; accumulate != 0 adds into out
void launch_spmv(const DevCrt &t, const u32 *rowptr, const u32 *col, const u64 *val /*[nnz][24] AoS*/, const u64 *z,
                 size_t ldz, u64 *out, size_t m, int accumulate, hipStream_t s, size_t r0 = 0, size_t rcnt = (size_t)-1 /* all rows */);
// out = sum_{j<nm} M_j z_j (nm <= 4), z_j = z + j*z_stride: one launch, one write of out
void launch_spmv_sum(const DevCrt &t, u32 nm, const u32 *const *rowptr, const u32 *const *col, const u64 *const *val, const u64 *z,
                     size_t z_stride, size_t ldz, u64 *out, size_t m, hipStream_t s, size_t r0 = 0, size_t rcnt = (size_t)-1 /* all rows */);
// q[col] = sum_{rows} eq[row] * val  (CSC: colptr over n columns, rowidx, val AoS)
// general matrices (more than ~1.5 entries per row): block = 32 rows x 8 slots, z gathered as whole elements from an element-major copy (zaos: scratch of
// nm * n * 24 words; z = nm plane-major vectors [24][n], z_stride words apart, or null when zaos holds the copies already).  out = (accumulate ? out : 0) + sum_j M_j z_j
void launch_spmv_rows(const DevCrt &t, u32 nm, const u32 *const *rowptr, const u32 *const *col, const u64 *const *val, const u64 *z, size_t z_stride, size_t n,
                      u64 *zaos, u64 *out, size_t m, int accumulate, hipStream_t s, size_t r0 = 0, size_t rcnt = (size_t)-1);
void launch_spmv_t_eq(const DevCrt &t, const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t m,
                      u64 *q, size_t n, hipStream_t s, size_t c0 = 0, size_t ccnt = (size_t)-1 /* all columns */);
// dots: out[a][b] = sum_i X_a[i] (.) Y_b[i] (ring tables, slot-wise), a < na, b < nb -> out AoS [na][nb][24]
void launch_dot_batch(const DevCrt &t, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n,
                      u64 *partial, u64 *out, hipStream_t s);
size_t dot_partial_words(u32 na, u32 nb);
// ring table (.) fq3 table: out[a] = sum_i eq[i] * X_a[i]  -> AoS [na][24]
void launch_dot_eq(const DevCrt &t, const u64 *X, size_t ldx, u32 na, const u64 *eq, size_t ldeq, size_t n, u64 *partial,
                   u64 *out, hipStream_t s);
// T[k][c] = sum_i eq[i] * digit_k(planes[c][i]) (K bit-planes) or full value (mode_bits = 0): out AoS fq3 [K][24][3]
// ldp = leading dimension of the planes (0: n); a rank of a sharded step passes its index slice (planes + i0, eq + i0, n = count)
void launch_coef_eval(const DevCrt &t, const int32_t *planes, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits,
                      u64 *partial, u64 *out, hipStream_t s, size_t ldp = 0);
size_t coef_eval_partial_words(u32 K);
// zz_j = sum_k coef[k][j] * z_k  (fq3 scalars; z tables [K][24][ldz]) -> out [t][24][ldz]
// per_slot != 0: coef_dev holds K*tt*8 constants, one per slot (ring-element coefficients, folding.rs:258-268)
void launch_lincomb_z(const DevCrt &t, const u64 *z, size_t ldz, u32 K, const Fq3Const *coef_dev /*K*tt*/, u32 tt, size_t n,
                      u64 *out, hipStream_t s, u32 per_slot = 0);
// G[row][slot] += sum_{k<K} sum_{d<3} apow[k][d] * digit_k(planes[8d+slot][row])   (rows < n_planes)
void launch_add_fhat_comb(const DevCrt &t, const int32_t *planes, size_t n_planes, u32 K, const Fq3Const *apow_dev /*K*3*/,
                          u64 *G, size_t m, hipStream_t s, size_t r0 = 0, size_t rcnt = (size_t)-1 /* all positions */);

// ---- sumcheck (a10) ---------------------------------------------------------------------------------------------
// generic in-place-free fix: ring tables and fq3 tables, new[j] = old[2j] + r*(old[2j+1]-old[2j])
void launch_fix_ring(const DevCrt &t, const u64 *in, u64 *out, size_t n_in, Fq3Const r, hipStream_t s);
void launch_fix_fq3(const DevCrt &t, const u64 *in, u64 *out, size_t n_in, Fq3Const r, hipStream_t s);
// batched inner products on the int8 matrix cores (lf_dot_i8.hip): out[(a*nb + b)*24 + 3*slot + q] = sum_i X_a[slot][i] * Y_b[slot][i], na <= 16, nb <= 3.
// YB / part / tot: scratch (dot_i8_yb_bytes(n), dot_i8_part_words(n) int32, dot_i8_tot_words() int64).  Returns 0, or -1 if the shape is not handled.
size_t dot_i8_yb_bytes(size_t n);
size_t dot_i8_part_words(size_t n);
size_t dot_i8_tot_words();
int launch_dot_batch_i8(const DevCrt &t, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, int32_t *part,
                        long long *tot, u64 *out, hipStream_t s, bool y_packed = false, u32 nb_out = 0, u32 b0 = 0);
// (nb_out, b0: the nb vectors Y are vectors b0 .. b0 + nb - 1 of a set of nb_out -- outputs land at (a * nb_out + b0 + b) -- so that a set of more than three is
// run in groups; 0 = nb)
// the Y digits alone, for launch_dot_batch_i8(.., y_packed = true) calls on X vectors of the same alignment
int launch_dot_pack_y(const u64 *X, const u64 *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, hipStream_t s);
void launch_vs_combine(const u64 *vs /* [K][72] */, u32 K, u64 *v /* [72] */, hipStream_t s);   // v = sum_k 2^k v_s[k]
void launch_fix_final(const DevCrt &t, const u64 *in /* [rows3][3][2] */, u32 rows3, Fq3Const r, u64 *out /* [rows3][3] canonical */, hipStream_t s);

struct LinCombDesc {  // CCS multiset structure for the linearization comb (nifs/linearization/utils.rs:90-107)
    u32 t, q;
    u32 S_off[9];
    u32 S_idx[16];
    u64 c[8][24];  // coefficients c_i (ring elements, AoS)
    int c_unit[8]; // +1 / -1 when c_i is the ring element +-1 (multiplication skipped), else 0
    u32 first[4];  // table j starts a new multiset
    u32 ms[4];     // multiset of table j
};
// round message of the linearization sumcheck: tables Mz [t][24][ld], eq [3][ld]; n = current length
// out: (deg+1) ring elements AoS, deg = d+1 <= 4
// max_blocks (0 = default 256 per slot) bounds the grid: inside a fold step the linearization shares the GPU with the commit chain
// of the other lane, which is the critical path, and yields to it by running on fewer workgroups
void launch_lin_round(const DevCrt &t, const LinCombDesc &desc, const u64 *mz, size_t ld, const u64 *eq, size_t ldeq, size_t n,
                      u32 deg, u64 *partial, u64 *out, hipStream_t s, u32 max_blocks = 0, u32 split_xmask = 0 /* see k_lin_round: eq = the per-pair table E_i */);
void launch_lin_round_fused(const DevCrt &t, const LinCombDesc &desc, const u64 *mz_prev, size_t ld_prev, const u64 *eq_prev, size_t ldeq_prev, Fq3Const r, u64 *mz_out,
                            size_t ld_out, u64 *eq_out, size_t ldeq_out, size_t n, u32 deg, u64 *partial, u64 *out, hipStream_t s, u32 max_blocks = 0,
                            u32 split_xmask = 0 /* eq_prev = E_{i-1}, eq_out = E_i (per pair) */);
void launch_eq_pairsum(const u64 *in, size_t ld_in, size_t n_out, u64 *out, size_t ld_out, hipStream_t s);
void launch_eq_expand(const DevCrt &t, const u64 *E, size_t lde, size_t pairs, Fq3Const w0, Fq3Const w1, u64 *out, size_t ldo, hipStream_t s);

struct FoldRoundArgs {
    const u64 *eqL, *eqR, *eqB;  // fq3 tables [3][ld]
    const u64 *G1, *G2;          // ring tables [24][ld]
    size_t ld;                   // leading dimension of the tables above
    size_t n;                    // current table length
    size_t p0, pcnt;             // pair range handled by this launch (all pairs: 0, n/2; a rank's slice when sharded)
    size_t pF0;                  // first pair held by the materialised f-hat buffer
};
// round 1 of the folding sumcheck straight from the coefficient planes (f-hat virtual, b = 2)
void launch_fold_round1(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                        u32 K, const Fq3Const *mu_pow_dev /*2K*3*/, u64 *partial, u64 *out, hipStream_t s);
// G part (eqL*G1 + eqR*G2) of a round message: out[X*24 + 3*slot + q], X = 0..4 (the norm part: lf_sv_rounds.h)
void launch_fold_round_g(const DevCrt &t, const FoldRoundArgs &a, u64 *partial, u64 *out, hipStream_t s);
// after r_1: materialise fixed f-hat tables F[2K*3][24][n/2] = f0 + r1*(f1-f0)
void launch_fold_materialize(const DevCrt &t, const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t m, u32 K,
                             Fq3Const r1, u64 *F, hipStream_t s);
// round 2, still from the planes: entries are d_a + (d_b - d_a) r1 (a.* are the once-fixed tables, a.n = m/2)
void launch_fold_round2(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                        const Fq3Const *mu_pow_dev, Fq3Const r1, u64 *partial, u64 *out, hipStream_t s);
// rounds 1 / 2 as table look-ups (large rounds): poly_dev [ncode][4][3] = coefficients of h^3 - h for the 9 / 81 digit codes of a pair,
// tp_dev (2K*3 * ncode * 12 words) receives mu_kd * poly; the round kernel gathers and adds, no multiplication per table
void launch_fold_round_tab(const DevCrt &t, int round, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes, u32 K,
                           const Fq3Const *mu_pow_dev, const u64 *poly_dev, u64 *tp_dev, u64 *partial, u64 *out, hipStream_t s);
// after r_2: F[2K*3][24][m/4] = sum_b W_b * digit(f[4j+b]), W = eq((r1,r2), b)
void launch_fold_materialize2(const DevCrt &t, const int32_t *planesL, const int32_t *planesR, size_t n_planes, size_t j0, size_t q, u32 K,
                              const Fq3Const W[4], u64 *F, hipStream_t s);
// general round on materialised tables F [2K*3][24][ldF] (b = 2)
void launch_fold_round(const DevCrt &t, const FoldRoundArgs &a, const u64 *F, size_t ldF, u32 K, const Fq3Const *mu_pow_dev,
                       u64 *partial, u64 *out, hipStream_t s);
// the same round with fix_variables of the previous round's tables fused in (unsharded driver, large rounds): reads entries
// 4p..4p+3 of Fprev, stores the fixed pair to Fout for the next round
void launch_fold_round_lut_fix_tab(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                                   const u64 *lut_dev, Fq3Const r, u64 *sq_dev, u64 *mt_dev, u64 *Fout, size_t ldout, u32 K, const Fq3Const *mu_pow_dev,
                                   u64 *partial, u64 *out, hipStream_t s, const u64 *E = nullptr, size_t ldE = 0 /* split form: the per-pair eq table, see k_fold_round */);
void launch_fold_round_lut_fix5(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                                const u64 *lut_dev, Fq3Const r3, Fq3Const r4, u64 *xx_dev, u64 *yy_dev, u64 *mt_dev, u64 *Fout, size_t ldout, u32 K,
                                const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s, const u64 *E = nullptr, size_t ldE = 0 /* split form: the per-pair eq table, see k_fold_round */);
void launch_fold_round_fix(const DevCrt &t, const FoldRoundArgs &a, const u64 *Fprev, size_t ldprev, Fq3Const r, u64 *Fout, size_t ldout, u32 K,
                           const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s, const u64 *E = nullptr, size_t ldE = 0 /* split form: the per-pair eq table, see k_fold_round */);
// rounds 3 / 4 straight from the coefficient planes through the 81-entry digit look-up table lut_dev [81][3] (entry of code
// sum_b t_b 3^b = sum_b (t_b - 1) W_b, W = eq((r1,r2), .)); round 4 also fixes with r3 and writes the m/8-entry tables
void launch_fold_round_lut(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                           const u64 *lut_dev, u32 K, const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s);
// round 3 with per-table products mu_kd * {value, value^2, value^3} of the look-up values (mutab_dev: 3 * 2K*3 * 81 * 4 words,
// filled by this call): two lazy products per table.  Only for the default non-residue (nu = 2^40).
void launch_fold_round_lut_mu(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                              const u64 *lut_dev, u64 *mutab_dev, u32 K, const Fq3Const *mu_pow_dev, u64 *partial, u64 *out, hipStream_t s);
void launch_fold_round_lut_fix(const DevCrt &t, const FoldRoundArgs &a, const int32_t *planesL, const int32_t *planesR, size_t n_planes,
                               const u64 *lut_dev, Fq3Const r, u64 *Fout, size_t ldout, u32 K, const Fq3Const *mu_pow_dev, u64 *partial,
                               u64 *out, hipStream_t s);
size_t round_partial_words();

// ---- Poseidon sponge on the device (one wave per sponge; ark [30][24], mds [24][24] canonical words in device memory) ----------
void launch_sponge_script(const u64 *ark, const u64 *mds, const u32 *ops, u32 nops, const u64 *words, u64 *out, u64 *state_out, hipStream_t s);

// ---- persistent sumcheck tail (no host hop per round; see k_fold_tail) ------------------------------------------------
constexpr u32 TAIL_MAX_ROUNDS = 32;
struct TailMail {   // host-mapped mailbox (hipHostMallocMapped): GPU -> host round messages, host -> GPU challenges
    u64 msg[TAIL_MAX_ROUNDS][128];     // message of tail round i (5 x 24 words used), valid once msg_seq[i] == epoch
    u32 msg_seq[TAIL_MAX_ROUNDS];
    u64 chal[TAIL_MAX_ROUNDS][4];      // challenge answering tail round i (3 words used), valid once chal_seq[i] == epoch
    u32 chal_seq[TAIL_MAX_ROUNDS];
    u32 abort_seq;                     // host: = epoch to make the kernel give up
    u32 err;                           // device: = epoch when a wait timed out
    u64 chal_out[TAIL_MAX_ROUNDS][4];  // device-transcript mode: the challenge the device sponge drew after tail round i (for the host's record)
    u64 sponge[32];                    // device-transcript mode: the sponge after the last round (24 state words, rate index, mode)
    u64 dbg[TAIL_MAX_ROUNDS][8];       // LF_TAIL_DEBUG builds: wall-clock stamps (100 MHz) of the round's stages
};
struct FoldTailArgs {
    u64 *T[2];            // T[0]: the 57-plane special-table buffer (eqL eqR eqB G1 G2) of the round BEFORE the tail (n0 entries, ld n0); T[1] unused
    u64 *F[2];            // F[0]: f-hat tables [2K*3][24][n0] of that round; F[1]: receives the fully fixed tables [2K*3][24][2]
    size_t n0;
    u32 rounds, K;
    const Fq3Const *mu_pow;
    u64 *partial;         // round_partial_words()
    u64 *eqpriv;          // fold_tail_eqpriv_words(n0, K): private working sets of the workgroups (128-byte aligned)
    u32 *counters;        // [TAIL_MAX_ROUNDS], zero before the first launch (self-resetting)
    u64 *dev_chal;        // [TAIL_MAX_ROUNDS][4] device scratch
    TailMail *mail;       // device address of the mapped mailbox
    u32 epoch;            // > 0, different for every launch
    Fq3Const r_first;     // challenge answering the round before the tail
    // LF_DEVICE_TRANSCRIPT=1: the Fiat-Shamir transcript of the tail rounds runs on the device (sponge_*_wave): no host round trip at all
    u32 dev_transcript;
    const u64 *pos_ark, *pos_mds;   // Poseidon constants in device memory
    u64 *sponge_state;    // device [26]: the sponge when the tail starts (written by the host), updated every round
};
struct LinTailArgs {     // persistent tail of the linearization sumcheck (k_lin_tail): same mailbox protocol
    const u64 *T;         // Mz tables [t][24][n0] of the round BEFORE the tail
    const u64 *E;         // eq table [3][n0] of that round
    u64 *Tout;            // receives the fully fixed Mz tables [t][24][2]
    size_t n0;
    u32 rounds, deg;      // deg = degree of the round polynomial (deg + 1 evaluations per message)
    u64 *partial;         // 128 words
    u64 *priv;            // lin_tail_priv_words(n0, t)
    u32 *counters;
    u64 *dev_chal;
    TailMail *mail;
    u32 epoch;
    Fq3Const r_first;
    u32 dev_transcript;
    const u64 *pos_ark, *pos_mds;
    u64 *sponge_state;
};
size_t lin_tail_priv_words(size_t n0, u32 t);
u32 launch_lin_tail(const DevCrt &t, const LinCombDesc &desc, const LinTailArgs &A, hipStream_t s);
constexpr u32 FOLD_TAIL_MAX_BLOCKS = 256;
size_t fold_tail_eqpriv_words(size_t n0, u32 K);
u32 launch_fold_tail(const DevCrt &t, const FoldTailArgs &A, int num_cus, hipStream_t s);

// ---- folded witness (a13 in coefficient domain) ---------------------------------------------------------------
// out[c][j] = sum_{i<2K} (rho_i * bitplane_i)(c)  with rho_i small-coefficient polynomials (i8 [2K][24])
void launch_fold_witness(const int32_t *planesL, const int32_t *planesR, size_t n, u32 K, const int8_t *rho_dev, int32_t *out,
                         hipStream_t s);

}  // namespace lf

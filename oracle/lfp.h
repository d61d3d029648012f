/*
 * lfp.h -- CPU ORACLE of the LatticeFold+ double-commitment construction (TEST INFRASTRUCTURE ONLY, same rule as lfo.h: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product never links it).
 *
 * Restates, for the ring the reference runs LatticeFold+ on -- FrogRing RqPoly, coefficient form, Z_p[X]/(X^16 + 1),
 * p = 15912092521325583641 (cyclotomic-rings/src/rings/frog.rs:1-25):
 *   tensor / tensor_product           crates/latticefold-plus/src/utils.rs:45-83   (KATs utils.rs:118-131 -> tests/golden/kats.json "lfp_tensor")
 *   split                             utils.rs:12-43
 *   RgInstance::from_f                rgchk.rs:260-331   (the "double commitment": benches/double_commitment.rs:53-82)
 *   Decomp::decompose                 decomp.rs:32-99    (base-B split into two parts, their commitments and MLE evaluations; no transcript)
 *   PoseidonTranscript, set check, range check (prover and verifier): lfp_protocol.c (transcript.rs, setchk.rs, rgchk.rs:81-258)
 *
 * PARITY STATUS: **unpinned beyond the two tensor KATs**.  The arithmetic of this path lives in stark-rings @ 886a89f (absent):
 * `exp`, `decompose_to_vec`, `gadget_decompose`, `Matrix::{try_mul_mat, try_mul_vec, hconcat}`.  Restated here from their call sites
 * and from the LatticeFold+ paper (section 4): exp(a) = sgn(a) X^a in Z_p[X]/(X^d + 1), i.e. X^a for a >= 0 and X^(d+a) for
 * -d/2 < a < 0 (always a positive unit monomial); balanced digits as in lfo_ring.c mode 0; gadget_decompose maps element e of a row to
 * positions [e*l, (e+1)*l), least significant digit first; hconcat keeps the k blocks of d columns in order.  Coefficient-form ring
 * products (negacyclic convolutions) are convention-free.
 */
#ifndef LFP_H
#define LFP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define LFP_P 15912092521325583641ULL
#define LFP_D 16

void lfp_ring_mul(const uint64_t *a, const uint64_t *b, uint64_t *out);           /* mod X^16 + 1, mod p */
void lfp_tensor_product(const uint64_t *a, size_t m, const uint64_t *b, size_t n, uint64_t *out);   /* utils.rs:45-66 over F_p */
void lfp_tensor(const uint64_t *r, size_t n, uint64_t *out /* 2^n */);             /* utils.rs:68-83 */
void lfp_balanced_digits(uint64_t v, uint64_t base, unsigned digits, int64_t *out);
int lfp_exp(int64_t c, uint64_t *out /* 16 */);                                    /* unit monomial of a digit, -d/2 < c < d/2 (else -1) */
/* A: kappa x n ring elements row-major, f: n ring elements (16 canonical words each, coefficient form) */
void lfp_commit(const uint64_t *A, uint32_t kappa, size_t n, const uint64_t *f, uint64_t *out /* kappa*16 */);   /* Matrix::try_mul_vec */
/* RgInstance::from_f (rgchk.rs:260-331).  Outputs:
 *   Df      k*n*16 digits (D_f[k_i][n_i][d_i], int8)
 *   comMf   k * kappa * 16 ring elements: comM_f[k_i][row i][column c] = sum_j A[i][j] * exp(D_f[k_i][j][c])
 *   tau     n base-ring values (canonical), = split(hconcat(comM_f), n, d/2, l)
 *   cm_f, C_Mf, cm_mtau   kappa ring elements each
 * returns 0, -1 if a digit is outside the exp domain, -2 if tau does not fit n (the reference panics) */
int lfp_rg_from_f(const uint64_t *f, size_t n, const uint64_t *A, uint32_t kappa, uint64_t b, uint32_t k, uint32_t l, int8_t *Df, uint64_t *comMf,
                  uint64_t *tau, uint64_t *cm_f, uint64_t *C_Mf, uint64_t *cm_mtau);
/* Decomp::decompose (decomp.rs:32-99).  r_a / r_b: log2(n) ring elements each (the two components of the evaluation point pairs);
 * nm matrices with n rows in CSR form, coefficients = ring elements (16 words per non-zero).  Outputs: F0, F1 n ring elements;
 * C0, C1 kappa; v0, v1: (1 + nm) pairs (value at r_a, value at r_b) of ring elements.  returns -1 unless n is a power of two */
int lfp_decompose(const uint64_t *f, size_t n, const uint64_t *A, uint32_t kappa, uint64_t B, const uint64_t *r_a, const uint64_t *r_b, uint32_t nm,
                  const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val, uint64_t *F0, uint64_t *F1, uint64_t *C0, uint64_t *C1,
                  uint64_t *v0, uint64_t *v1);
void lfp_splitmix_fill(uint64_t seed, uint64_t start, size_t count, uint64_t *out);   /* uniform words < p (workload generator) */

/* ---- lfp_protocol.c: the transcript-driven part (PoseidonTranscript<RqPoly>, set check, range check), prover AND verifier ------------ */
typedef struct lfp_tr lfp_tr;
lfp_tr *lfp_tr_new(void);                                          /* PoseidonTranscript::empty::<FrogPoseidonConfig>() (transcript.rs:20-32) */
void lfp_tr_free(lfp_tr *t);
lfp_tr *lfp_tr_clone(const lfp_tr *t);
void lfp_tr_absorb(lfp_tr *t, const uint64_t *ring, size_t count); /* Transcript::absorb / absorb_slice: 16 words per element */
uint64_t lfp_tr_challenge(lfp_tr *t);                              /* get_challenge: one F_p word (squeezed, absorbed back) */
void lfp_tr_squeeze_bytes(lfp_tr *t, size_t n, uint8_t *out);
void lfp_short_challenge(lfp_tr *t, uint64_t *out16);              /* utils::short_challenge(128, ..) (utils.rs:87-101) */
void lfp_short_challenge_from_bytes(const uint8_t *bs16, uint64_t *out16);   /* = FrogChallengeSet decoding (rings/frog.rs:35-55, KAT :66-96) */
void lfp_poseidon_params(uint64_t *ark720, uint64_t *mds576);      /* regenerated table (checksums: kats.json "poseidon_frog_params") */
void lfp_poseidon_permute(uint64_t *state24);
void lfp_psi(uint64_t *out16);                                     /* psi with ct(psi exp(a)) = a, -d/2 < a < d/2 */
uint64_t lfp_ct_psi_mul(const uint64_t *b16);
/* In::set_check (setchk.rs:65-262) / Out::verify (:266-340); shapes in lfp_protocol.c */
int lfp_set_check(lfp_tr *tr, unsigned nvars, const uint64_t *msets, unsigned nmat, unsigned ncols, const uint64_t *vsets, unsigned nvec, unsigned nM,
                  const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out,
                  uint64_t *b_out);
int lfp_set_check_verify(lfp_tr *tr, unsigned nvars, unsigned nmat, unsigned ncols, unsigned nvec, unsigned nM, const uint64_t *msgs, const uint64_t *e,
                         const uint64_t *b, uint64_t *r_out);
/* Rg::range_check (rgchk.rs:81-186) / Dcom::verify (:193-258) */
int lfp_range_check(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, const uint64_t *const *Mf, const uint64_t *const *tau, const uint64_t *const *mtau,
                    const uint64_t *const *f, unsigned nM, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val,
                    uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out, uint64_t *v_out, uint64_t *a_out, uint64_t *bb_out, uint64_t *c_out);
int lfp_range_check_verify(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, unsigned nM, const uint64_t *msgs, const uint64_t *e, const uint64_t *b,
                           const uint64_t *v, const uint64_t *a, const uint64_t *bb, const uint64_t *c, uint64_t *r_out);
/* Cm::prove (cm.rs:56-347) / CmProof::verify (:349-580); shapes in lfp_protocol.c */
int lfp_cm_prove(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, unsigned ell, unsigned kappa, const uint64_t *const *Mf, const uint64_t *const *tau,
                 const uint64_t *const *mtau, const uint64_t *const *f, const uint64_t *const *comMf, const uint64_t *const *fcoms, unsigned nM,
                 const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out,
                 uint64_t *b_out, uint64_t *v_out, uint64_t *a_out, uint64_t *bb_out, uint64_t *c_out, uint64_t *comh, uint64_t *pa, uint64_t *pb,
                 uint64_t *ea, uint64_t *eb, uint64_t *g, uint64_t *cm_g, uint64_t *ro, uint64_t *vo);
int lfp_cm_verify(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, unsigned ell, unsigned kappa, unsigned nM, const uint64_t *const *fcoms,
                  const uint64_t *msgs, const uint64_t *e, const uint64_t *b, const uint64_t *v, const uint64_t *a, const uint64_t *bb, const uint64_t *c,
                  const uint64_t *comh, const uint64_t *pa, const uint64_t *pb, const uint64_t *ea, const uint64_t *eb, uint64_t *cm_g, uint64_t *ro,
                  uint64_t *vo);
/* ComR1CS::linearize / ComR1CSProof::verify (r1cs.rs:76-162), DecompProof::verify (decomp.rs:101-123); shapes in lfp_protocol.c */
int lfp_r1cs_linearize(lfp_tr *tr, unsigned nvars, const uint64_t *f, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val,
                       uint64_t *msgs, uint64_t *ro, uint64_t *evals);
int lfp_r1cs_verify(lfp_tr *tr, unsigned nvars, const uint64_t *msgs, const uint64_t *evals, uint64_t *ro);
int lfp_decomp_verify(const uint64_t *C0, const uint64_t *C1, unsigned kappa, const uint64_t *v0, const uint64_t *v1, unsigned count, const uint64_t *cm_f,
                      const uint64_t *v, uint64_t B);
#ifdef __cplusplus
}
#endif
#endif

/*
 * lfo.h -- CPU ORACLE for the LatticeFold prover hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is a plain-C restatement of the reference algorithm
 * (NethermindEth/latticefold @ 2025-12-26, `NIFSProver::prove` and everything it calls).
 * It exists to CHECK the HIP product path; it is never linked, imported or executed by
 * the product (latticefold_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * PARITY STATUS.  Pinned against every known-answer test the reference holds for this
 * path (tests/golden/kats.json, extracted by tests/tools/extract_kats.py): Poseidon sponge
 * + challenge derivation, short-challenge decoding, RotSum, f-hat layout, Ajtai closed form.
 * The arithmetic itself lives in the un-vendored dependency stark-rings @ 886a89f1 (absent
 * from /root/reference, no network).  Three conventions of that crate are NOT pinned by any
 * in-tree reference test and are therefore DATA here (lfo_set_ring / digit_mode), with
 * mathematically-derived defaults:  (1) the CRT slot map (slot order + per-slot cube root),
 * (2) the F_{p^3} non-residue (default 2^40), (3) the balanced-digit tie/sign rule.
 * => "parity unpinned" for exactly those three items; everything else is KAT-pinned.
 *
 * Data format everywhere: flat little-endian uint64_t canonical residues in [0,p).
 * Ring element = d words (24 Goldilocks / 72 BabyBear).  Coefficient form: X^0..X^{d-1}.  NTT form:
 * slot-major, slot k = words [tau*k, tau*k+tau) = coordinates of an F_{p^tau} element (the order
 * `coeffs().flat_map(to_base_prime_field_elements)` yields, transcript/poseidon.rs:40-47).
 */
#ifndef LFO_H
#define LFO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t u64;
typedef uint32_t u32;

/* Ring selection at compile time: default GoldilocksRingNTT; -DLFO_RING_BABYBEAR builds
 * liblfo_bb.so for BabyBearRingNTT (cyclotomic-rings/src/rings/babybear.rs:1-25): p = 15*2^27+1,
 * Phi_216 = X^72 - X^36 + 1 = prod over the 8 primitive 24th roots zeta of (X^9 - zeta). */
#ifdef LFO_RING_BABYBEAR
#define LFO_P 2013265921ULL
#define LFO_D 72
#define LFO_TAU 9
#define LFO_MOD_BITS 31
#else
#define LFO_P 0xFFFFFFFF00000001ULL /* Goldilocks 2^64 - 2^32 + 1 */
#define LFO_D 24                    /* ring degree (Phi_72 = X^24 - X^12 + 1) */
#define LFO_TAU 3
#define LFO_MOD_BITS 64
#endif
#define LFO_SLOTS 8
int lfo_ring_degree(void);  /* LFO_D of this build */
int lfo_ring_tau(void);
u64 lfo_modulus(void);

/* ---- ring tables (data, see header comment) ------------------------------------------ */
/* nonres: F_{p^3} = F_p[Y]/(Y^3 - nonres).  y[8][3]: image of X in slot k (y_k^3 must be a
 * primitive 24th root of unity in F_p, all distinct).  Returns 0 or <0 if inconsistent. */
int lfo_set_ring(u64 nonres, const u64 *y /* 8*3 */);
void lfo_get_ring(u64 *nonres, u64 *y /* 8*3 */);
/* the fully general data form of SURVEY 8(c): CRT as a dense d x d matrix (row r = slot-major output coordinate, column c = coefficient
 * index) and the structure constants of F_{p^tau} in an arbitrary F_p-basis with e_0 = 1 (tau^3 words, e_i e_j = sum_k t[i][j][k] e_k).
 * Returns 0, or <0 if e_0 is not the unit / the matrix is singular / the map is not multiplicative.  lfo_set_ring returns to the
 * binomial form. */
int lfo_set_ring_general(const u64 *crt_matrix, const u64 *tensor);
void lfo_set_digit_mode(int mode); /* 0 = sign-magnitude truncation (default), 1 = floor/Euclid */

/* ---- element-wise ring ops ------------------------------------------------------------ */
void lfo_crt(const u64 *in, u64 *out, size_t count);  /* stark-rings CRT::elementwise_crt  */
void lfo_icrt(const u64 *in, u64 *out, size_t count); /* stark-rings ICRT::elementwise_icrt */
void lfo_ring_mul_ntt(const u64 *a, const u64 *b, u64 *out, size_t count);
void lfo_ring_mul_coeff(const u64 *a, const u64 *b, u64 *out); /* schoolbook mod Phi_72 */
void lfo_fq3_mul(const u64 *a, const u64 *b, u64 *out);

/* balanced decomposition (stark_rings::balanced_decomposition; call sites arith.rs:235,
 * nifs/decomposition/utils.rs:23-31,48).  layout 0 = "chunked": element i -> out[i*digits+j]
 * (gadget_decompose); layout 1 = "transposed": out[j*count + i] (decompose_to_vec.transpose). */
void lfo_decompose(const u64 *coeff_in, size_t count, u64 base, u32 digits, int layout, u64 *out);
/* out[i] = sum_j base^j * in[i*digits + j]  (either form; arith.rs:305,330) */
void lfo_recompose(const u64 *in, size_t count_out, u64 base, u32 digits, u64 *out);

/* cyclotomic-rings/src/rotation.rs:45-104 */
void lfo_rot_lin_combination(const u64 *rho_coeff, const u64 *theta, u32 n, u32 tau_elems, u64 *out);
/* cyclotomic-rings/src/rings/goldilocks.rs:36-68 */
int lfo_short_challenge_from_bytes(const uint8_t *bs, size_t n, u64 *coeff_out);

/* ---- Poseidon transcript (transcript/poseidon.rs, arkworks-0.4 duplex sponge) ---------- */
typedef struct lfo_transcript lfo_transcript;
lfo_transcript *lfo_transcript_new(void);
void lfo_transcript_free(lfo_transcript *);
void lfo_transcript_absorb_fq(lfo_transcript *, const u64 *x, size_t n); /* raw sponge.absorb */
void lfo_transcript_absorb_ring(lfo_transcript *, const u64 *elems, size_t count);
void lfo_transcript_get_challenge(lfo_transcript *, u64 *fq3_out);
void lfo_transcript_get_short_challenge(lfo_transcript *, u64 *coeff_out);
void lfo_poseidon_params(u64 *ark /*720*/, u64 *mds /*576*/);
void lfo_poseidon_permute(u64 *state /*24*/);

/* ---- Ajtai (commitment/commitment_scheme.rs:37-54) --------------------------------------- */
/* out[i] = sum_j A[i*n+j] (.) f[j], all NTT form.  Returns 0. */
int lfo_ajtai_commit(const u64 *A, u32 kappa, size_t n, const u64 *f, u64 *out);

/* ---- MLE helpers (utils/sumcheck/utils.rs:100-170, utils/mle_helpers.rs:65-88) ---------- */
/* points are ring elements (NTT form), exactly like the reference's `&[R]` */
void lfo_build_eq(const u64 *r /* nv ring elems */, u32 nv, u64 *out /* (1<<nv) ring elems */);
/* evaluate a table of ring elements (NTT form, len entries, zero-padded to 1<<nv) */
void lfo_mle_eval(const u64 *table, size_t len, const u64 *r, u32 nv, u64 *out);

/* ---- protocol ---------------------------------------------------------------------------- */
typedef struct {
    u32 s;       /* log2 m (CCS rows, = sumcheck variables) */
    u32 wit_len; /* ring elements in w_ccs */
    u32 l;       /* public inputs x_len */
    u32 L, K, b; /* DecompositionParams */
    u64 B;
    u32 kappa;
    u32 t, q, d; /* CCS: #matrices, #multisets, degree */
} lfo_params;

/* CCS in CSR form, matrices m x n with n = l + 1 + wit_len; values are NTT-form ring elements */
typedef struct {
    const u32 *const *rowptr; /* [t][m+1] */
    const u32 *const *col;    /* [t][nnz] */
    const u64 *const *val;    /* [t][nnz*24] */
    const u32 *S_off;         /* [q+1] */
    const u32 *S_idx;
    const u64 *c;             /* [q*24] */
} lfo_ccs;

/* flat sizes (in ring elements) */
size_t lfo_lcccs_len(const lfo_params *);  /* r[s] v[tau] cm[kappa] u[t] x_w[l] h */
size_t lfo_cccs_len(const lfo_params *);   /* cm[kappa] x_ccs[l] */
size_t lfo_proof_len(const lfo_params *);  /* see lfo_protocol.c: proof layout */

/* Witness::from_w_ccs (arith.rs:230-248): w_ccs NTT[wit_len] -> f_coeff[N] (N = wit_len*L) */
void lfo_witness_from_w_ccs(const lfo_params *, const u64 *w_ccs, u64 *f_coeff);

/* LFLinearizationProver::prove (nifs/linearization.rs:145-189) on a fresh transcript state
 * supplied by the caller.  lcccs_out flat; proof part written to lin_proof_out
 * (s*(d+2) + tau + t ring elements). */
int lfo_linearize(const lfo_params *, const lfo_ccs *, lfo_transcript *, const u64 *cccs,
                  const u64 *f_coeff, u64 *lcccs_out, u64 *lin_proof_out);

/* NIFSProver::prove (nifs.rs:48-103).  acc = LCCCS flat, w_acc_f_coeff[N*24], cm_i = CCCS flat,
 * w_i_f_coeff[N*24].  Outputs: folded LCCCS flat, folded witness f_0 (NTT, N*24), proof flat. */
int lfo_fold_step(const lfo_params *, const lfo_ccs *, const u64 *A, lfo_transcript *,
                  const u64 *acc, const u64 *w_acc_f_coeff, const u64 *cm_i,
                  const u64 *w_i_f_coeff, u64 *lcccs_out, u64 *f0_out, u64 *proof_out);

/* NIFSVerifier::verify (nifs.rs:117-163): returns 0 on accept and writes the folded LCCCS,
 * <0 on reject (code tells which check failed). */
int lfo_verify(const lfo_params *, const lfo_ccs *, lfo_transcript *, const u64 *acc,
               const u64 *cm_i, const u64 *proof, u64 *lcccs_out);

/* LFDecompositionProver::prove (nifs/decomposition.rs:33-88) alone: dec_proof_out = u_s[K][t] | v_s[K][tau] | x_s[K][l+1] | y_s[K][kappa],
 * lcccs_s_out (optional) = the K decomposed LCCCS flat */
int lfo_decomposition_prove(const lfo_params *, const lfo_ccs *, const u64 *A, lfo_transcript *, const u64 *lcccs,
                            const u64 *f_coeff, u64 *dec_proof_out, u64 *lcccs_s_out);
/* building blocks of LFFoldingProver::prove on their own (ABI parity tests): the folding sumcheck (utils/sumcheck.rs:53-80 with the
 * comb of folding/utils.rs:273-325) on a caller-supplied mle list, the Horner combine (folding.rs:208-226) and compute_f_0 (258-268) */
int lfo_sumcheck_fold(const lfo_params *, lfo_transcript *, const u64 *tables, const u64 *mu, u64 *msgs_out, u64 *point_out);
void lfo_horner_combine(const u64 *tables, u32 groups, u32 per_group, size_t len, const u64 *ch, u64 *out);
void lfo_lincomb(const u64 *coef, const u64 *tables, u32 n_terms, size_t len, u64 *out);
/* workload.py::splitmix_fq in C (synthetic benchmark inputs) */
void lfo_splitmix_fill(u64 seed, u64 start, size_t count, u64 *out);

int lfo_num_threads(void);
void lfo_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif

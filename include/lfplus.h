/*
 * lfplus.h -- C ABI of the LatticeFold+ slice of liblfhip.so (MI355X / gfx950): the DOUBLE COMMITMENT of the range-check instance,
 * on the ring the reference runs latticefold-plus on (FrogRing RqPoly: Z_p[X]/(X^16 + 1), p = 15912092521325583641, COEFFICIENT form,
 * 16 canonical little-endian u64 words per ring element).  SURVEY.md section 8(f) row 4.
 *
 * Reference interfaces replaced (crates/latticefold-plus):
 *   RgInstance::from_f(f, &A, &DecompParameters{b,k,l})      src/rgchk.rs:260-331   (bench: benches/double_commitment.rs:53-82)
 *     cfs -> decompose_to_vec(b, k) -> D_f           rgchk.rs:263-284
 *     M_f = exp(D_f), comM_f = A * M_f               rgchk.rs:286-303
 *     tau = split(hconcat(comM_f), n, d/2, l)        rgchk.rs:304-306, utils.rs:12-43
 *     m_tau = exp(tau); cm_f, C_Mf, cm_mtau          rgchk.rs:308-320
 *   Matrix::try_mul_vec (Ajtai commitment, coefficient form)   stark-rings-linalg, call sites rgchk.rs:313-319
 *   Decomp::decompose(&A, B)                         src/decomp.rs:32-99
 *   utils::tensor / tensor_product                   src/utils.rs:45-83 (KATs utils.rs:118-131)
 *   PoseidonTranscript<RqPoly>, utils::short_challenge       src/transcript.rs:20-78, src/utils.rs:87-101
 *   In::set_check / Out::verify                      src/setchk.rs:65-262 / 266-340
 *   Rg::range_check / Dcom::verify                   src/rgchk.rs:81-186 / 193-258
 *
 * M_f and m_tau are matrices / vectors of unit monomials; they cross this boundary as their EXPONENT digits (int8 in (-d/2, d/2)):
 * exp(a) = X^a for a >= 0 and X^(d + a) for a < 0.  The Rust binding rebuilds Matrix<R> from them if a caller needs the dense form
 * (INTEGRATION.md); nothing on this path does.
 *
 * Everything runs on the GPU; there is no CPU fallback (LFPLUS_E_NO_DEVICE).  Parity status: bit-exact against oracle/lfp.c, which is
 * pinned to the reference only through the tensor KATs -- the digit / gadget conventions of the un-vendored stark-rings crate are
 * restated from call sites (see oracle/lfp.h).
 */
#ifndef LFPLUS_H
#define LFPLUS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LFPLUS_D 16
#define LFPLUS_P 15912092521325583641ULL

enum {
    LFPLUS_OK = 0,
    LFPLUS_E_ARG = -1,        /* null pointer, non-canonical word, shape mismatch, parameter outside the envelope */
    LFPLUS_E_NO_DEVICE = -2,  /* no HIP device: the library has no CPU path */
    LFPLUS_E_HIP = -3,
    LFPLUS_E_EXP_DOMAIN = -4, /* a digit of f or of tau is outside (-d/2, d/2): the reference's exp() returns None and from_f panics */
    LFPLUS_E_SMALL_N = -5,    /* kappa*k*d*l*d >= n: the reference's split() panics ("small n unsupported") */
    LFPLUS_E_REJECT = -6      /* a verifier rejected the proof (*stage says where) */
};
#define LFPLUS_ABSENT (-128)  /* exponent digit of a zero entry of a monomial set (an absent sparse-matrix coefficient) */

typedef struct lfplus_ctx lfplus_ctx;
int lfplus_ctx_create(int device, lfplus_ctx **out);
void lfplus_ctx_destroy(lfplus_ctx *ctx);
/* Scratch memory (tables of the protocol stages) is pooled per context and, when a context is destroyed, kept in a process-wide per-device cache for the next
 * contexts (a prover built per proof would otherwise pay hipMalloc for every table inside every prove).  Bounded per device: LFPLUS_CACHE_GB when set (0 = off),
 * otherwise min(32 GB, a quarter of the device's memory).  Every allocator of the library releases the cache and retries before it reports out-of-memory.
 * lfplus_scratch_trim frees what the cache holds on `device` (< 0: on every device); lfplus_scratch_bytes returns the bytes it holds there now.
 * No counterpart in the reference (the Rust allocator owns its Vecs). */
void lfplus_scratch_trim(int device);
size_t lfplus_scratch_bytes(int device);
const char *lfplus_last_error(const lfplus_ctx *ctx);

/* ---- multi-GPU: one prover column-sharded over `world` ranks, one GPU each (BASELINE configs[4]) ------------------------------------------
 * Rank g owns rows [g n / world, (g + 1) n / world) of every n-indexed object: its COLUMNS of the commitment matrix (lfplus_set_matrix then takes the
 * kappa x n / world slice), its rows of D_f, of the set-check / Cm / linearization tables and of the folded witness g.  Witnesses (lfplus_set_witness, the
 * F0 / F1 lfplus_decompose returns) are whole on every rank, as the host hands them over; tau and m_tau (a short non-zero prefix) are rebuilt whole.
 * Exchanged, each as "all-gather + sum mod p": the partial commitments (2 per double commitment, 1 per lfplus_commit / lfplus_decompose), the partial
 * sumcheck round messages of the first log2(n / world) rounds (then the tables are gathered, one entry per rank, and the last log2(world) rounds run
 * replicated), the partial evaluations; all-gathered whole: h (per instance, only with constraint matrices) and the folded witness.  Every rank runs the
 * identical transcript and returns the identical proof.  Call before lfplus_set_matrix; contexts that lfplus_share_matrix the matrix share the transport.
 * n / world must be a power of two >= 4 world.  lfplus_set_sharding: host transport, `cb` must all-gather `words` u64 from every rank into
 * recv_all[world * words] in rank order and return 0 (gloo in the tests).  lfplus_dist_init: an RCCL communicator owned by the library (one rank per GPU,
 * xGMI) from a 128-byte ncclUniqueId that rank 0 obtains with lfplus_dist_unique_id and the launcher distributes; RCCL is loaded with dlopen. */
typedef int (*lfplus_exchange_fn)(void *user, const uint64_t *send, uint64_t *recv_all, size_t words);
int lfplus_set_sharding(lfplus_ctx *ctx, int rank, int world, lfplus_exchange_fn cb, void *user);
int lfplus_dist_unique_id(uint8_t *id128);
int lfplus_dist_init(lfplus_ctx *ctx, int rank, int world, const uint8_t *id128);
/* exchange log of the context's transport: number of exchanges, summed and maximal host-side latency in microseconds (reset != 0 clears it) */
int lfplus_dist_stats(lfplus_ctx *ctx, uint64_t *n_exchanges, double *total_us, double *max_us, int reset);
/* u64 words this rank contributed to its exchanges (reset != 0 clears the counter) */
int lfplus_dist_stats_words(lfplus_ctx *ctx, uint64_t *words_sent, int reset);
/* TIMING MODEL, not a transport (as lf_set_sharding_model in lfhip.h): rank `rank` of `world` with no peers; zeros stand in for their words.  The "proofs" of such a
 * prover are meaningless; tools/shard_model.py --lfplus measures a rank's share of a sharded PlusProver::prove with it on a one-GPU box. */
int lfplus_set_sharding_model(lfplus_ctx *ctx, int rank, int world);

/* Ajtai matrix A (kappa x n ring elements, row-major, coefficient form); stays resident in HBM.  kappa <= 64.  In a sharded context: the rank's columns,
 * kappa x (n / world), and `n` is that local width. */
int lfplus_set_matrix(lfplus_ctx *ctx, const uint64_t *A, uint32_t kappa, uint64_t n);
/* use the commitment matrix resident in `from` (same device) without copying it; the allocation is reference-counted and lives until its last holder
 * re-uploads or is destroyed */
int lfplus_share_matrix(lfplus_ctx *ctx, lfplus_ctx *from);
/* witness vector f (n ring elements); stays resident */
int lfplus_set_witness(lfplus_ctx *ctx, const uint64_t *f, uint64_t n);

/* RgInstance::from_f on the resident (A, f).  b >= 2 (digits must land in (-8, 8): b <= 14), 1 <= k <= 16, 1 <= l <= 64.
 * Results stay on the device until lfplus_rg_read. */
int lfplus_rg_from_f(lfplus_ctx *ctx, uint64_t b, uint32_t k, uint32_t l);
/* The same pass enqueued on the context's second stream; returns at once.  from_f needs the witness, the matrix and the parameters only -- no challenge --
 * so a prover issues it as soon as the witness is resident and lets it run next to the linearization's latency-bound sumcheck rounds (Mlin::mlin calls
 * from_f after every linearization, mlin.rs:52-60; the order is not observable).  lfplus_rg_from_f / lfplus_mlin with the same parameters collect the result
 * and report the pass's errors; every other call that touches the witness or the from_f buffers waits for the pass first (lfplus_join_async does only that).
 * A sharded context ignores the hint.  LFPLUS_NO_ASYNC_FROM_F=1 in the environment turns the hint off. */
int lfplus_rg_from_f_async(lfplus_ctx *ctx, uint64_t b, uint32_t k, uint32_t l);
int lfplus_join_async(lfplus_ctx *ctx);
/* Any pointer may be NULL.  Df: k*n*16 int8 (D_f[k_i][n_i][d_i]); comMf: k*kappa*16*16 words (comM_f[k_i][row][column] ring elements);
 * tau: n words; mtau: n int8 (exponent digits of m_tau); cm_f / C_Mf / cm_mtau: kappa*16 words each. */
int lfplus_rg_read(lfplus_ctx *ctx, int8_t *Df, uint64_t *comMf, uint64_t *tau, int8_t *mtau, uint64_t *cm_f, uint64_t *C_Mf, uint64_t *cm_mtau);
/* the same computation `iters` times back to back, timed with HIP events on the library's stream (inputs resident): average ms */
int lfplus_rg_from_f_timed(lfplus_ctx *ctx, uint64_t b, uint32_t k, uint32_t l, uint32_t iters, double *ms_avg);

/* Decomp::decompose(&A, B) (src/decomp.rs:32-99; no transcript) on the resident (A, f), n a power of two:
 *   F = f.decompose_to_vec(B, 2).transpose() -> (F0, F1);  C_i = A F_i;
 *   v_i = [(mle(F_i)(r_a), mle(F_i)(r_b)), then per matrix M_j (mle(M_j F_i)(r_a), mle(M_j F_i)(r_b))]
 * r_a / r_b: log2(n) ring elements each (the components of Decomp::r); the nm matrices have n rows, CSR with ring-element coefficients
 * (val[j]: 16 words per non-zero).  Outputs (any may be NULL): F0, F1 n*16 words; C0, C1 kappa*16; v0, v1 (1+nm)*2*16. */
int lfplus_decompose(lfplus_ctx *ctx, uint64_t B, const uint64_t *r_a, const uint64_t *r_b, uint32_t nm, const uint32_t *const *rowptr,
                     const uint32_t *const *col, const uint64_t *const *val, uint64_t *F0, uint64_t *F1, uint64_t *C0, uint64_t *C1, uint64_t *v0,
                     uint64_t *v1);
/* The same, with the parts left on the device: F0 / F1 become the resident witnesses of dst0 / dst1 (contexts of the same device and width; `ctx` itself may be
 * one of them -- its witness, the folded g, is consumed first).  The accumulator of a folding prover (plus.rs:96-103: the two LinB of one prove are inputs of the
 * next) then never crosses PCIe: 2 x 134 MB down and up again per prove at 2^20 rows.  lfplus_get_witness reads a context's resident witness back. */
int lfplus_decompose_resident(lfplus_ctx *ctx, uint64_t B, const uint64_t *r_a, const uint64_t *r_b, uint32_t nm, const uint32_t *const *rowptr,
                              const uint32_t *const *col, const uint64_t *const *val, lfplus_ctx *dst0, lfplus_ctx *dst1, uint64_t *C0, uint64_t *C1,
                              uint64_t *v0, uint64_t *v1);
int lfplus_get_witness(lfplus_ctx *ctx, uint64_t *f_out /* n * 16 words */, uint64_t n);

/* Matrix::try_mul_vec: out (kappa*16 words) = A * v for a general vector of n ring elements (host pointer) */
int lfplus_commit(lfplus_ctx *ctx, const uint64_t *v, uint64_t n, uint64_t *out);

/* utils::tensor(r) over the base field: out has 2^n words (n <= 28); utils::tensor_product: out has m*n words (or the non-empty side) */
int lfplus_tensor(lfplus_ctx *ctx, const uint64_t *r, uint32_t n, uint64_t *out);
int lfplus_tensor_product(lfplus_ctx *ctx, const uint64_t *a, uint64_t m, const uint64_t *b, uint64_t n, uint64_t *out);

/* ---- the transcript-driven part: PoseidonTranscript<RqPoly>, monomial set check, range check (prover on the GPU, verifier on the host) ----
 * PoseidonTranscript::empty::<FrogPoseidonConfig>() (src/transcript.rs:20-78; table cyclotomic-rings/src/rings/poseidon/frog.rs): a host
 * object.  absorb = Transcript::absorb / absorb_slice (16 words per ring element), challenge = get_challenge (ONE F_p word: RqPoly's base ring
 * has extension degree 1; squeezed and absorbed back), squeeze_bytes as arkworks (7 bytes per element), short_challenge =
 * utils::short_challenge(128, ..) (utils.rs:87-101). */
typedef struct lfplus_transcript lfplus_transcript;
lfplus_transcript *lfplus_transcript_new(void);
lfplus_transcript *lfplus_transcript_clone(const lfplus_transcript *t);
void lfplus_transcript_free(lfplus_transcript *t);
int lfplus_transcript_absorb(lfplus_transcript *t, const uint64_t *ring, size_t count);
int lfplus_transcript_challenge(lfplus_transcript *t, uint64_t *out);
int lfplus_transcript_squeeze_bytes(lfplus_transcript *t, size_t n, uint8_t *out);
int lfplus_short_challenge(lfplus_transcript *t, uint64_t *out16);
int lfplus_poseidon_params(uint64_t *ark720, uint64_t *mds576);   /* the regenerated Frog table (checksummed against the reference's) */
int lfplus_poseidon_permute(uint64_t *state24, int plain);        /* one permutation; plain = 1: the textbook form, 2: the scalar sparse form, 0: what the transcript runs (self-tests) */
int lfplus_poseidon_simd(void);                                   /* 1 when the transcript's permutation runs on the host's AVX-512 IFMA lanes (lfp_poseidon_simd.cc) */

/* In::set_check (src/setchk.rs:65-262): nmat matrix sets of n x ncols unit monomials and nvec vector sets of n, n = 2^nvars, as exponent
 * digits (int8 in (-8, 8); LFPLUS_ABSENT = zero entry); nM matrices (n x n, CSR, ring coefficients) for the M_i f rows of Step 3.
 * Outputs: r (nvars words), msgs (nvars * 4 ring elements: the sumcheck messages, constants), e ((1 + nM) * nmat * ncols ring elements,
 * e[q][set][column]), b (nvec ring elements).  The transcript advances exactly as the reference's. */
int lfplus_set_check(lfplus_ctx *ctx, lfplus_transcript *t, uint32_t nvars, const int8_t *mat_digits, uint32_t nmat, uint32_t ncols, const int8_t *vec_digits,
                     uint32_t nvec, uint32_t nM, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val, uint64_t *r_out,
                     uint64_t *msgs, uint64_t *e_out, uint64_t *b_out);
/* Out::verify (setchk.rs:266-340), host only.  LFPLUS_OK, or LFPLUS_E_REJECT with *stage = 1 / 2 (sumcheck) or 3 (final evaluation). */
int lfplus_set_check_verify(lfplus_transcript *t, uint32_t nvars, uint32_t nmat, uint32_t ncols, uint32_t nvec, uint32_t nM, const uint64_t *msgs, const uint64_t *e,
                            const uint64_t *b, uint64_t *r_out, int *stage);
/* Rg::range_check (src/rgchk.rs:81-186) over L instances: ctxs[l] holds witness f_l and its lfplus_rg_from_f results (same device, n = 2^nvars,
 * k).  Outputs: the set check's r, msgs, e ((1 + nM) * (L k) * 16 ring elements), b (L), and per instance v (16 words), a (1 + nM words),
 * bb (1 + nM ring elements: DcomEvals::b), c (1 + nM ring elements). */
int lfplus_range_check(lfplus_ctx *const *ctxs, uint32_t L, lfplus_transcript *t, uint32_t nM, const uint32_t *const *rowptr, const uint32_t *const *col,
                       const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out, uint64_t *v_out, uint64_t *a_out,
                       uint64_t *bb_out, uint64_t *c_out);
/* Dcom::verify (rgchk.rs:193-258), host only: stages 1-3 set check, 4 ct(psi b) != a, 5 ct(psi sum_i (d/2)^i u_i) != v / c */
int lfplus_range_check_verify(lfplus_transcript *t, uint32_t nvars, uint32_t L, uint32_t k, uint32_t nM, const uint64_t *msgs, const uint64_t *e, const uint64_t *b,
                              const uint64_t *v, const uint64_t *a, const uint64_t *bb, const uint64_t *c, uint64_t *r_out, int *stage);
/* Cm::prove (src/cm.rs:56-347) over L resident instances (as lfplus_range_check; ell = DecompParameters::l): the range check, the folding
 * challenges s (3) and s' (k d), h = M_f s' and comh, the two degree-2 ring-valued sumcheckers, the folded witness g = s0 tau + s1 m_tau + s2 f + h
 * and the folded instance ComX (cm.rs:545-580).  Outputs: r .. c_out as lfplus_range_check; comh (L x kappa ring elements); pa / pb (nvars x 3 ring
 * elements: the sumcheck messages); ea / eb (L x (4 + 4 nM) ring elements in the reference's table order: tau, m_tau, f, h, then per matrix M tau,
 * M m_tau, M f, M h); cm_g (L x kappa); ro (ro_a | ro_b, 2 x nvars words); vo (L x (1 + nM) x 2).  g stays on the device in ctxs[l]
 * (lfplus_cm_read_g: n ring elements); g_out may be null or receive L x n ring elements.  LFPLUS_E_ARG when t0 does not fit n (the reference panics). */
int lfplus_cm_prove(lfplus_ctx *const *ctxs, uint32_t L, lfplus_transcript *t, uint32_t ell, uint32_t nM, const uint32_t *const *rowptr, const uint32_t *const *col,
                    const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out, uint64_t *v_out, uint64_t *a_out, uint64_t *bb_out,
                    uint64_t *c_out, uint64_t *comh, uint64_t *pa, uint64_t *pb, uint64_t *ea, uint64_t *eb, uint64_t *cm_g, uint64_t *ro, uint64_t *vo,
                    uint64_t *g_out);
int lfplus_cm_read_g(lfplus_ctx *ctx, uint64_t *g_out);
/* CmProof::verify (cm.rs:349-543), host only.  fcoms[l] = cm_f | C_Mf | cm_mtau of instance l (kappa ring elements each).  LFPLUS_OK: accepted, and
 * cm_g / ro / vo are the folded instance recomputed from the proof.  LFPLUS_E_REJECT with *stage = 1-5 (range check), 6 (a sumcheck round),
 * 7 (t0 does not fit n), 8 (final evaluation of a sumchecker) */
int lfplus_cm_verify(lfplus_transcript *t, uint32_t nvars, uint32_t L, uint32_t k, uint32_t ell, uint32_t kappa, uint32_t nM, const uint64_t *const *fcoms,
                     const uint64_t *msgs, const uint64_t *e, const uint64_t *b, const uint64_t *v, const uint64_t *a, const uint64_t *bb, const uint64_t *c,
                     const uint64_t *comh, const uint64_t *pa, const uint64_t *pb, const uint64_t *ea, const uint64_t *eb, uint64_t *cm_g, uint64_t *ro,
                     uint64_t *vo, int *stage);

/* The constraint-system matrices (n x n, CSR, ring coefficients) made resident once: every entry point that takes (nM, rowptr, col, val) uses them when
 * rowptr is NULL (nM must equal their number; for the multi-instance calls they are taken from ctxs[0]).  lfplus_set_matrices(ctx, n, 0, ..) drops them. */
int lfplus_set_matrices(lfplus_ctx *ctx, uint64_t n, uint32_t nM, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val);
int lfplus_share_matrices(lfplus_ctx *ctx, lfplus_ctx *from);   /* use `from`'s resident matrices (same device, no copy; reference-counted) */
/* ComR1CS::linearize (src/r1cs.rs:76-139) on the resident witness f (lfplus_set_witness; n = 2^nvars ring elements) and the R1CS matrices A, B, C
 * (n x n, CSR, ring coefficients): g_q = M_q f, the degree-3 ring-valued sumcheck of eq(r, x) (g_A g_B - g_C)(x), evaluations at ro.
 * Outputs: msgs (nvars x 4 ring elements), ro (nvars words), evals = v | va | vb | vc (4 ring elements). */
int lfplus_r1cs_linearize(lfplus_ctx *ctx, lfplus_transcript *t, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val, uint64_t *msgs,
                          uint64_t *ro, uint64_t *evals);
/* ComR1CSProof::verify (r1cs.rs:141-162), host only: LFPLUS_E_REJECT with *stage = 1 (a sumcheck round) or 2 (e (va vb - vc) != s; the reference asserts) */
int lfplus_r1cs_verify(lfplus_transcript *t, uint32_t nvars, const uint64_t *msgs, const uint64_t *evals, uint64_t *ro, int *stage);
/* DecompProof::verify (src/decomp.rs:101-123), host only: *stage = 1 (C0 + B C1 != cm_f) or 2 (v0 + B v1 != v over `count` pairs of ring elements) */
int lfplus_decomp_verify(const uint64_t *C0, const uint64_t *C1, uint32_t kappa, const uint64_t *v0, const uint64_t *v1, uint32_t count, const uint64_t *cm_f,
                         const uint64_t *v, uint64_t B, int *stage);
/* Mlin::mlin (src/mlin.rs:42-107) over the L resident witnesses of ctxs (same commitment matrix, n, device): RgInstance::from_f(b, k, l) on each, Cm::prove
 * (outputs as lfplus_cm_prove), then the folded LinB2: cm_g_sum (kappa ring elements) and vo_sum ((1 + nM) x 2) on the host, g = sum_l g_l on the device where
 * it REPLACES ctxs[0]'s resident witness (PlusProver::prove hands it to Decomp::decompose: plus.rs:90-95).  fcoms_out (may be null): L x 3 x kappa ring
 * elements, cm_f | C_Mf | cm_mtau per instance (what CmProof::verify needs). */
int lfplus_mlin(lfplus_ctx *const *ctxs, uint32_t L, lfplus_transcript *t, uint64_t b, uint32_t k, uint32_t l, uint32_t nM, const uint32_t *const *rowptr,
                const uint32_t *const *col, const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out, uint64_t *v_out,
                uint64_t *a_out, uint64_t *bb_out, uint64_t *c_out, uint64_t *comh, uint64_t *pa, uint64_t *pb, uint64_t *ea, uint64_t *eb, uint64_t *cm_g,
                uint64_t *ro, uint64_t *vo, uint64_t *fcoms_out, uint64_t *cm_g_sum, uint64_t *vo_sum);

#ifdef __cplusplus
}
#endif
#endif

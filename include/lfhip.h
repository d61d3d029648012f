/*
 * lfhip.h -- C ABI of liblfhip.so: the MI355X (gfx950) LatticeFold prover hot path.
 *
 * This is the drop-in boundary for NethermindEth/latticefold's `NIFSProver::prove`
 * (crates/latticefold/src/nifs.rs:48-103).  The reference is 100 % safe Rust with
 * `#![forbid(unsafe_code)]` (crates/latticefold/src/lib.rs:4), so these entry points are what
 * a new `latticefold-hip-sys` crate would bind (INTEGRATION.md shows the stub); each one cites
 * the reference interface it replaces.
 *
 * Data format: flat little-endian uint64_t CANONICAL residues in [0,p), p = 2^64 - 2^32 + 1.
 * Ring element = 24 words.  Coefficient form: X^0..X^23 of Z_p[X]/(X^24 - X^12 + 1).  NTT form:
 * slot-major, slot k = words [3k,3k+3) = (c0,c1,c2) of an F_{p^3} element -- the order
 * `coeffs().flat_map(to_base_prime_field_elements)` yields (transcript/poseidon.rs:40-47).
 * (arkworks stores Montgomery form internally; the Rust shim converts with into_bigint/from.)
 *
 * Ownership: the caller owns every host buffer; the context owns all device memory.  All
 * pointers below are HOST pointers.  Every function returns 0 on success or a negative
 * LF_ERR_* code (lf_strerror); nothing aborts.  A context serialises its own calls
 * (internal mutex), so it may be shared between threads (the reference calls `commit` from
 * Rayon workers, nifs/decomposition.rs:185-187); use the batched entry points to collapse
 * such loops into one call.
 */
#ifndef LFHIP_H
#define LFHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_RING_WORDS 24

/* ABI version: bumped whenever an existing entry point changes meaning (additions alone do not bump it).  lf_abi_version() returns the constant the
 * library was built with; a binding compares it with the LFHIP_ABI_VERSION it was generated from.
 *   5  lf_last_fold_paths: *sv_round_mask is the GEMM-round mask alone on both rings (bits 8..15 used to carry the split-round count of the
 *      linearization on Goldilocks, and BabyBear contexts returned 0): the count moved to lf_last_lin_split_rounds, the table rounds' mask to
 *      lf_last_fold_split_rounds.  lf_debug_i8_prof is declared (it was exported without a prototype).
 *   4  (round 3) first numbered state of this header. */
#define LFHIP_ABI_VERSION 5
int lf_abi_version(void);

enum {
    LF_OK = 0,
    LF_ERR_INVALID = -1,        /* bad argument / CommitmentError::WrongWitnessLength (commitment.rs:14-27),
                                   MleEvaluationError::IncorrectLength (utils/mle_helpers.rs:13-17) */
    LF_ERR_HIP = -2,            /* HIP runtime failure (no GPU, out of memory, ...) */
    LF_ERR_UNSUPPORTED = -3,    /* parameter outside what the kernels implement */
    LF_ERR_BAD_TABLES = -4,     /* ring tables are not a ring isomorphism */
    LF_ERR_NORM = -5,           /* witness coefficient exceeds the decomposition bound */
    LF_ERR_SIZE_BOUNDS = -6,    /* CSError::InvalidSizeBounds (nifs.rs:165-173) */
    LF_ERR_REJECT = -8,         /* lf_verify_host: the proof does not verify */
    LF_ERR_STATE = -7,          /* call sequence misuse (the reference panics: sumcheck/prover.rs:41,63,75,81) */
};
const char *lf_strerror(int code);

typedef struct lf_ctx lf_ctx;
typedef struct lf_witness lf_witness;       /* device-resident Witness (arith.rs:213-223): centred f_coeff */
typedef struct lf_transcript lf_transcript; /* PoseidonTranscript (transcript/poseidon.rs:17-75), host */

/* ---- context ------------------------------------------------------------------------------- */
int lf_ctx_create(lf_ctx **out, int device);   /* GoldilocksRingNTT */
/* Ring selection (cyclotomic-rings/src/rings/{goldilocks,babybear}.rs:1-25).  A context, its witnesses and the transcripts
 * used with it belong to ONE ring; every entry point below then works on ring elements of lf_ring_words(ring) u64 words
 * (24 / 72) and F_{p^tau} challenges of lf_ring_tau(ring) words (3 / 9), canonical residues of lf_ring_modulus(ring).
 * BabyBearRingNTT (BASELINE configs[2]): p = 15*2^27+1, Phi_216 = X^72 - X^36 + 1, 8 slots of F_{p^9}; on the device the
 * words are 31-bit centred Montgomery residues.  lf_set_ring_tables takes y[8][tau] (tau = 9 words per slot for BabyBear). */
#define LF_RING_GOLDILOCKS 0
#define LF_RING_BABYBEAR 1
int lf_ctx_create_ring(lf_ctx **out, int device, int ring);
int lf_ctx_ring(const lf_ctx *);
int lf_ring_words(int ring);
int lf_ring_tau(int ring);
uint64_t lf_ring_modulus(int ring);
void lf_ctx_destroy(lf_ctx *);
/* The CRT slot map, the F_{p^3} non-residue and the digit rule live in the un-vendored crate
 * stark-rings @ 886a89f and are DATA here: y[8*3] = image of X in slot k.  Defaults are installed
 * by lf_ctx_create.  tools/probe_stark_rings.rs prints the true tables. */
int lf_set_ring_tables(lf_ctx *, uint64_t nonres, const uint64_t *y);
int lf_get_ring_tables(lf_ctx *, uint64_t *nonres, uint64_t *y);
/* External coordinate basis of F_{p^tau} (the other half of "conventions are data"): the kernels compute in the binomial basis
 * 1, Y, .., Y^(tau-1) of F_p[Y]/(Y^tau - nonres) -- `nonres` and `y` above are expressed in it -- while the caller's library may keep
 * field elements in another F_p-basis: a tower basis F_{p^3}[Z]/(Z^3 - Y) with coordinates ordered 3j+i (a permutation), a rescaled
 * or entirely different basis.  T (tau x tau, row-major, canonical words) maps internal to external coordinates, ext = T * int, and
 * must fix 1 (first column = e_0; the base field is coordinate 0 in both bases).  From then on every NTT-form ring element and every
 * F_{p^tau} challenge crossing this ABI is in EXTERNAL coordinates -- inputs are converted on entry, outputs on exit -- and the
 * Fiat-Shamir transcript absorbs / squeezes external coordinates, so proofs are those of a prover computing natively in that basis.
 * The multiplication (structure) tensor the caller's field has is then T applied to the binomial one; tools/probe_stark_rings.rs prints
 * it.  Identity T switches the feature off (default; no cost).  LF_ERR_BAD_TABLES if T is singular or does not fix 1.  lf_verify_host
 * keeps the default basis. */
int lf_set_ext_basis(lf_ctx *, const uint64_t *T);
/* The balanced-digit rule of stark_rings::balanced_decomposition, the third unpinned convention, as data: mode 0 (default) = truncate
 * toward zero and move |rem| > base/2 to the other side (ties +-base/2 keep the sign of the value); mode 1 = floor rule, digits in
 * [-base/2, base/2) (a tie becomes -base/2 with a carry).  Applies to every base-B decomposition on the path (lf_decompose, the gadget
 * decomposition of Witness::from_w_ccs, x_s); for base 2 the digits are {-1, 0, 1} = sign and bits of the magnitude under either mode
 * (the floor rule does not terminate on positive values in base 2).  tools/probe_stark_rings.rs prints the edge cases that tell the
 * modes apart. */
int lf_set_digit_mode(lf_ctx *, int mode);
int lf_device_synchronize(lf_ctx *);
int lf_mem_info(lf_ctx *, size_t *free_bytes, size_t *total_bytes); /* hipMemGetInfo of the context's device */
/* device arithmetic self-test: fast F_{p^3} product / lazy accumulators vs the generic schoolbook path on n pseudo-random
 * and edge operand sets; *mismatches must come back 0 */
int lf_selftest_field(lf_ctx *, uint64_t seed, uint32_t n, uint64_t *mismatches);

/* ---- a1/a2: CRT::elementwise_crt / ICRT::elementwise_icrt (arith.rs:232,238,300,327) -------- */
int lf_ntt_fwd(lf_ctx *, const uint64_t *in, uint64_t *out, size_t count); /* in/out may alias */
int lf_ntt_inv(lf_ctx *, const uint64_t *in, uint64_t *out, size_t count);

/* ---- a3: balanced decomposition (stark_rings::balanced_decomposition) ------------------------
 * layout 0 = gadget_decompose (element i -> out[i*digits + j], arith.rs:235);
 * layout 1 = decompose_to_vec(..).transpose() (digit table j at out + j*count*24,
 *            nifs/decomposition/utils.rs:45-49).  Power-of-two bases only. */
int lf_decompose(lf_ctx *, const uint64_t *coeff_in, size_t count, uint64_t base, unsigned digits, int layout,
                 uint64_t *out);
/* gadget_recompose / recompose (arith.rs:305,330): out[i] = sum_j base^j in[i*digits + j] */
int lf_recompose(lf_ctx *, const uint64_t *in, size_t count_out, uint64_t base, unsigned digits, uint64_t *out);
/* a15: l-infinity norm of ICRT(f) with centred representatives.  *ok = (max < bound);
 * *max_out (optional) receives the maximum.  `unsigned_variant` != 0 instead reproduces the literal
 * `Witness::within_bound` (arith.rs:372-386): every canonical coefficient < bound. */
int lf_linf_check(lf_ctx *, const uint64_t *f_ntt, size_t count, uint64_t bound, int unsigned_variant, int *ok,
                  uint64_t *max_out);

/* ---- a5: AjtaiCommitmentScheme::{new, commit, commit_ntt} (commitment_scheme.rs:23-77) -------- */
/* kappa <= 128 (Goldilocks: more than 48 rows of A are committed in equal row chunks, one LDS tile each) / 32 (BabyBear);
 * larger -> LF_ERR_INVALID */
/* The context keeps A in ONE resident form: coefficient form, cut into bytes, in int8-MFMA operand order (lf_ajtai_i8.hip) -- 8*24*kappa*n bytes for
 * Goldilocks, 4*72*kappa*n for BabyBear, built once here (rows pass through one row buffer on their way in), never per step.  The digit-plane commitments
 * of the decomposition step (decomposition.rs:178-201) and the general commitments below (commit_ntt, Witness::commit: lf_ajtai_i8g.hip) both stream it;
 * there is no NTT-form copy and no integer-multiplier commit kernel any more. */
/* free / total bytes of the context's device as the HIP runtime reports them (hipMemGetInfo): what a caller sizes batches against */
int lf_device_memory(lf_ctx *, size_t *free_bytes, size_t *total_bytes);
int lf_ajtai_load(lf_ctx *, const uint64_t *A /* kappa*n ring elements, row-major, NTT form */, size_t kappa, size_t n);
/* synthetic i.i.d. matrix generated on the device (bench; same stream as workload.Workload.ajtai_matrix) */
int lf_ajtai_generate(lf_ctx *, uint64_t seed, size_t kappa, size_t n);
/* out[b*kappa + i] = sum_j A[i][j] (.) f[b*n + j];  n != width -> LF_ERR_INVALID (WrongWitnessLength) */
int lf_ajtai_commit(lf_ctx *, const uint64_t *f, size_t n, size_t batch, uint64_t *out);

/* Multi-GPU: a rank that holds only a COLUMN SLICE of A (lf_ajtai_load / lf_ajtai_generate on that slice)
 * gets the partial commitment of its slice from lf_ajtai_commit; partial commitments are exchanged with one all-gather
 * and added with lf_modsum (canonical residues; plain ncclSum would wrap mod 2^64, not mod p).
 * parts = nparts x words canonical words, out = words. */
int lf_modsum(const uint64_t *parts, size_t nparts, size_t words, uint64_t *out);
int lf_modsum_ring(const uint64_t *parts, size_t nparts, size_t words, uint64_t *out, int ring);   /* the same for either modulus */
/* Intra-step sharding of one fold step over `world` (power of two) ranks, one GPU each.  Must be set before the Ajtai matrix is
 * loaded/generated: each rank then keeps columns [rank*n/world, (rank+1)*n/world) of A.  Witnesses, CCS and all O(n) vectors are
 * replicated; sharded are the Ajtai commitments (partial commitments all-gathered + added mod p) and the folding-sumcheck rounds
 * (index slice by the high bits; partial round messages all-gathered + added; the table slices are gathered -- one all-gather per sumcheck -- once the
 * tables are down to LF_SHARD_LIN_MIN = 16384 / LF_SHARD_FOLD_MIN = 2048 entries, at most m / 16, or fewer than 64 pairs per rank remain).
 * Every rank runs the identical transcript and returns the identical proof.  `cb` must all-gather `words` u64 from every rank
 * into recv_all[world*words] in rank order and return 0 (RCCL/xGMI via torch.distributed in latticefold_amd/dist.py). */
typedef int (*lf_exchange_fn)(void *user, const uint64_t *send, uint64_t *recv_all, size_t words);
int lf_set_sharding(lf_ctx *, int rank, int world, lf_exchange_fn cb, void *user);
/* The production transport: one RCCL communicator per context (one rank per GPU, xGMI), created from a 128-byte ncclUniqueId that rank 0
 * obtains with lf_dist_unique_id (called twice, see lf_dist_init) and the launcher distributes (MPI, torch.distributed store, a file ...).  Exchanges then run on DEVICE
 * buffers in the context's streams: ncclAllGather + a modular-sum kernel, no host hop and no callback.  RCCL is loaded with dlopen (the
 * library has no link-time dependency on it); LF_ERR_UNSUPPORTED if it is not installed.  Same ordering rule as lf_set_sharding: before the
 * Ajtai matrix is loaded.  A rank whose step fails aborts the communicator so that its peers error out instead of waiting forever. */
int lf_dist_unique_id(uint8_t *id128);
/* ids = TWO unique ids (2 x 128 bytes): the two lanes of a fold step exchange concurrently and each gets its own communicator.
 * Start-up self-check: lf_dist_init runs the first collectives of both communicators concurrently from the two threads of a fold step, each on its lane's
 * stream, and checks every word received.  Passed: sharded steps run the threaded two-lane schedule.  Wrong words: the communicators stay usable, one
 * host thread issues every exchange.  No completion within LF_DIST_HANDSHAKE_MS (default 20000): both communicators are aborted and LF_ERR_STATE is
 * returned -- make fresh ids and call again with LF_DIST_NO_HANDSHAKE=1 in the environment (the conservative schedule).
 * lf_dist_two_lanes(ctx, -1) reports the outcome (1 = threaded schedule), (ctx, 0 / 1) overrides it: EVERY rank must run the same schedule, so a launcher
 * combines the ranks' outcomes (minimum) and sets the result on all of them (latticefold_amd/dist.py does). */
int lf_dist_init(lf_ctx *, int rank, int world, const uint8_t *ids);
int lf_dist_two_lanes(lf_ctx *, int set);
/* host transport with one callback per lane (the callbacks run on different threads, possibly at the same time: give each its own
 * ordered channel, e.g. two process groups); lf_set_sharding installs the same callback for both, which is only safe if it is
 * order-independent */
int lf_set_sharding_lanes(lf_ctx *, int rank, int world, lf_exchange_fn cb0, void *user0, lf_exchange_fn cb1, void *user1);
/* exchange log of the context: number of exchanges, summed and maximal host-side latency in microseconds (reset != 0 clears it) */
int lf_dist_stats(lf_ctx *, uint64_t *n_exchanges, double *total_us, double *max_us, int reset);
/* u64 words this rank contributed to its exchanges so far (an all-gather delivers (world - 1) times as many to it); reset != 0 clears the counter */
int lf_dist_stats_words(lf_ctx *, uint64_t *words_sent, int reset);
/* TIMING MODEL, not a transport: the context behaves as rank `rank` of `world` with no peers -- every kernel and host stage does that rank's share, every
 * exchange is enqueued in the lane's stream with zeros standing in for the peers' words, the schedule is the threaded two-lane one.  The "proofs" such a
 * context returns are meaningless; it exists so that the per-rank compute time of a G-way sharded step can be measured on a box with one GPU
 * (tools/shard_model.py).  Goldilocks contexts only; same ordering rule as lf_set_sharding. */
int lf_set_sharding_model(lf_ctx *, int rank, int world);

/* ---- a8/a9/a11: eq table and batched MLE evaluation (sumcheck/utils.rs:100-170, mle_helpers.rs:65-88) */
/* point = nv challenges in F_{p^3} (3 words each): the reference's points are always diagonal embeddings
 * of transcript challenges (linearization/utils.rs:119-122, folding/utils.rs:59-92). */
int lf_build_eq(lf_ctx *, const uint64_t *point, unsigned nv, uint64_t *out /* (1<<nv)*3 */);
int lf_mle_eval_batch(lf_ctx *, const uint64_t *tables /* ntables x len ring elems (NTT) */, size_t ntables, size_t len,
                      const uint64_t *point, unsigned nv, uint64_t *out /* ntables ring elems */);

/* ---- a7, a16: CCS statement (arith.rs:50-74) ---------------------------------------------------- */
typedef struct {
    uint32_t s;       /* log2 m */
    uint32_t wit_len; /* ring elements in w_ccs */
    uint32_t l;       /* x_len */
    uint32_t L, K, b; /* DecompositionParams (decomposition_parameters.rs:11-20) */
    uint64_t B;
    uint32_t kappa;
    uint32_t t, q, d; /* #matrices, #multisets, degree */
} lf_params;
/* matrices in CSR (rows m = 1<<s, cols n = l + 1 + wit_len), values = NTT-form ring elements */
int lf_ccs_load(lf_ctx *, const lf_params *, const uint32_t *const *rowptr, const uint32_t *const *col,
                const uint64_t *const *val, const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *c);
/* mat_vec_mul (arith/utils.rs:52-65): out (m ring elements) = M_j * z (n ring elements) */
int lf_spmv(lf_ctx *, unsigned j, const uint64_t *z, uint64_t *out);

size_t lf_lcccs_len(const lf_params *); /* ring elements: r[s] v[3] cm[kappa] u[t] x_w[l] h */
size_t lf_cccs_len(const lf_params *);  /* cm[kappa] x_ccs[l] */
size_t lf_proof_len(const lf_params *); /* LFProof flat, see DESIGN.md */
size_t lf_lcccs_len_ring(const lf_params *, int ring); /* same layouts with v[tau]: tau = 9 for BabyBear */
size_t lf_cccs_len_ring(const lf_params *, int ring);
size_t lf_proof_len_ring(const lf_params *, int ring);

/* ---- Witness (arith.rs:230-338) ------------------------------------------------------------------ */
int lf_witness_from_w_ccs(lf_ctx *, const uint64_t *w_ccs /* wit_len NTT */, lf_witness **out); /* from_w_ccs */
/* The same witness, built NEXT TO a running fold step (a chain's next instance, examples/e2e.rs made a loop): _begin starts the PCIe upload, the ICRT and
 * the gadget decomposition on a worker thread and the context's lowest-priority stream (buffers of its own) and returns at once; _finish waits, hands the
 * witness over and frees the job (out = NULL: abandon it).  w_ccs must stay valid until _finish returns; do not reload the constraint system meanwhile.
 * Goldilocks contexts in the default basis overlap with lf_fold_step; other configurations run the blocking call on the worker (same result). */
typedef struct lf_witness_job lf_witness_job;
int lf_witness_from_w_ccs_begin(lf_ctx *, const uint64_t *w_ccs /* wit_len NTT */, lf_witness_job **job);
int lf_witness_job_finish(lf_witness_job *job, lf_witness **out);
int lf_witness_from_f_coeff(lf_ctx *, const uint64_t *f_coeff /* N coeff-form */, lf_witness **out);
int lf_witness_from_f(lf_ctx *, const uint64_t *f_ntt /* N NTT */, lf_witness **out);           /* from_f */
int lf_witness_get_f_coeff(lf_ctx *, const lf_witness *, uint64_t *out /* N */);
int lf_witness_get_f(lf_ctx *, const lf_witness *, uint64_t *out /* N, NTT */);
int lf_witness_get_w_ccs(lf_ctx *, const lf_witness *, uint64_t *out /* wit_len, NTT */);
int lf_witness_commit(lf_ctx *, const lf_witness *, uint64_t *cm_out /* kappa */);              /* Witness::commit */
void lf_witness_free(lf_witness *);

/* ---- transcript (transcript.rs:13-51) --------------------------------------------------------------- */
lf_transcript *lf_transcript_new(void);
lf_transcript *lf_transcript_new_ring(int ring);
lf_transcript *lf_transcript_clone(const lf_transcript *);
void lf_transcript_free(lf_transcript *);
void lf_transcript_absorb_fq(lf_transcript *, const uint64_t *x, size_t n);       /* sponge.absorb */
void lf_transcript_absorb_ring(lf_transcript *, const uint64_t *elems, size_t n); /* absorb_slice */
void lf_transcript_get_challenge(lf_transcript *, uint64_t *fq3_out);
void lf_transcript_get_short_challenge(lf_transcript *, uint64_t *coeff_out);
void lf_transcript_squeeze_bytes(lf_transcript *, uint8_t *out, size_t n);         /* Transcript::squeeze_bytes (poseidon.rs:62-64): what a Rust impl of the trait over this handle needs */
void lf_poseidon_params(uint64_t *ark /* 720 */, uint64_t *mds /* 576 */);
/* one Poseidon permutation on 24 words; plain = 0: the transcript's path (sparse-matrix partial rounds; AVX-512 IFMA / AVX2
 * lanes chosen at run time, LF_POSEIDON_SCALAR=1 disables them), 1: the textbook round loop, 2: the scalar sparse-matrix code.
 * All three give the same output. */
void lf_poseidon_permute(uint64_t *state, int plain);
void lf_poseidon_params_ring(uint64_t *ark, uint64_t *mds, int ring);
void lf_poseidon_permute_ring(uint64_t *state, int plain, int ring);

/* PoseidonSponge on the DEVICE (Goldilocks): runs a script of absorb / squeeze operations on a fresh sponge in one
 * wave -- ops[i] = (kind << 24) | count, kind 0 = absorb the next `count` words of absorb_words, 1 = squeeze `count` words.  Same
 * duplex semantics as lf_transcript_* (transcript/poseidon.rs:29-75); pinned by the reference's challenge KATs.  state_out (optional):
 * 24 state words, rate index, mode.  The fold step's default transcript is the host one (a permutation is a serial chain: ~20 us on
 * a GPU wave, 1.5 us on a host core); LF_DEVICE_TRANSCRIPT=1 moves the tail rounds' transcript into the persistent kernel. */
int lf_device_sponge(lf_ctx *, const uint32_t *ops, size_t nops, const uint64_t *absorb_words, size_t n_words, uint64_t *squeezed_out,
                     size_t n_out, uint64_t *state_out);

/* ---- sumcheck split at the transcript (utils/sumcheck.rs:53-80, sumcheck/prover.rs:56-162) ---------
 * Generic entry for the linearization-shaped polynomial  comb = (sum_i c_i prod_{j in S_i} T_j) * T_last
 * over t ring tables + one slot-constant table: begin / round / end.  Rounds must be called in order;
 * r_prev = NULL in round 1.  (The folding sumcheck runs inside lf_fold_step on virtual f-hat tables.) */
int lf_sumcheck_lin_begin(lf_ctx *, const uint64_t *tables /* t x m ring elems */, const uint64_t *eq_point /* s*3 */);
int lf_sumcheck_lin_round(lf_ctx *, const uint64_t *r_prev /* 3 words or NULL */, uint64_t *evals_out /* (d+2) ring elems */);
int lf_sumcheck_lin_end(lf_ctx *);

/* The folding sumcheck, same split: MLSumcheck::prove_as_subprotocol (utils/sumcheck.rs:53-80) with the comb function of
 * nifs/folding/utils.rs:273-325 (b = 2).  `tables` = the mle list create_sumcheck_polynomial builds (folding/utils.rs:200-259):
 * [eq(r_L), G_L, eq(r_R), G_R, eq(beta), f-hat_{0,0} .. f-hat_{2K-1,tau-1}], (5 + 2K*tau) x m ring elements (NTT); the three eq
 * tables must be slot-constant (else LF_ERR_UNSUPPORTED); mu = 2K challenges of tau words (the last one is ONE in the protocol).
 * Each round returns 2b+1 = 5 ring elements.  (lf_fold_step itself runs this sumcheck on virtual f-hat tables.) */
int lf_sumcheck_fold_begin(lf_ctx *, const uint64_t *tables, const uint64_t *mu);
int lf_sumcheck_fold_round(lf_ctx *, const uint64_t *r_prev /* tau words or NULL */, uint64_t *evals_out /* (2b+1) ring elems */);
int lf_sumcheck_fold_end(lf_ctx *);
/* compute_f_0 (nifs/folding.rs:258-268): out[j] = sum_{i<n_terms} coef_i (.) tables_i[j]; coef = n_terms ring elements (NTT, e.g. the
 * CRT of the short challenges rho_i), tables = n_terms x len ring elements (NTT) */
int lf_lincomb(lf_ctx *, const uint64_t *coef, const uint64_t *tables, size_t n_terms, size_t len, uint64_t *out);
/* calculate_challenged_mz_mle (nifs/folding.rs:208-226) / prepare_g1_and_3_k_mles_list (nifs/folding/utils.rs:524-546):
 * out[x] = sum_{i<groups} sum_{j<per_group} c_i^(j+1) T_{i,j}[x]; tables = groups x per_group x len ring elements (NTT),
 * challenges = groups x tau words */
int lf_horner_combine(lf_ctx *, const uint64_t *tables, size_t groups, size_t per_group, size_t len, const uint64_t *challenges,
                      uint64_t *out);

/* ---- the path itself ----------------------------------------------------------------------------------- */
/* LFLinearizationProver::prove (nifs/linearization.rs:145-189) */
int lf_linearize(lf_ctx *, lf_transcript *, const uint64_t *cccs, const lf_witness *wit, uint64_t *lcccs_out,
                 uint64_t *lin_proof_out /* s*(d+2) + 3 + t ring elems */);
/* NIFSProver::prove (nifs.rs:48-103): one fold step.  w_out receives the folded witness handle. */
int lf_fold_step(lf_ctx *, lf_transcript *, const uint64_t *acc_lcccs, const lf_witness *w_acc, const uint64_t *cm_i_cccs,
                 const lf_witness *w_i, uint64_t *lcccs_out, lf_witness **w_out, uint64_t *proof_out);

/* The two other sub-provers of the reference as entry points of their own (the reference exposes all three as public traits).
 * LFDecompositionProver::prove (nifs/decomposition.rs:33-88): dec_proof_out = u_s[K][t] | v_s[K][tau] | x_s[K][l+1] | y_s[K][kappa]
 * ring elements; lcccs_s_out (optional) = the K decomposed LCCCS, flat; the K decomposed witnesses stay virtual (base-b parts of
 * `wit`).  Needs the CCS and the Ajtai matrix. */
int lf_decomposition_prove(lf_ctx *, lf_transcript *, const uint64_t *lcccs, const lf_witness *wit, uint64_t *lcccs_s_out,
                           uint64_t *dec_proof_out);
/* LFFoldingProver::prove (nifs/folding.rs:42-130): lcccs_s = 2K LCCCS (K parts of the accumulator's decomposition, then K of the
 * linearized instance's), w_left / w_right = the witnesses those are the base-b parts of.  fold_proof_out =
 * msgs[s][2b+1] | theta[2K][tau] | eta[2K][t] ring elements. */
int lf_folding_prove(lf_ctx *, lf_transcript *, const uint64_t *lcccs_s, const lf_witness *w_left, const lf_witness *w_right,
                     uint64_t *lcccs_out, lf_witness **w_out, uint64_t *fold_proof_out);

/* ---- measurement hooks (bench.py): HIP-event time of the last fold step, per phase, in ms ------------ */
#define LF_N_PHASES 8
int lf_last_phase_ms(lf_ctx *, float *out /* LF_N_PHASES */);
const char *lf_phase_name(int i);
/* dominant-kernel timing: total HIP-event time and launch count of the folding-sumcheck round kernel and of
 * the Ajtai kernel in the last fold step */
int lf_last_kernel_stats(lf_ctx *, float *fold_round_ms, int *fold_round_launches, float *ajtai_ms, int *ajtai_launches);

/* wall-clock marks of the calling thread during the last lf_fold_step (measurement only; bench.py's `roofline.phases`): mark i has its
 * name (NUL-terminated, <= 31 characters) at names + 32 i and ms[i] = milliseconds since the start of the step.  Returns the number of marks
 * written (<= max_marks), or an error code < 0 */
int lf_last_timeline(lf_ctx *, char *names /* 32 * max_marks */, double *ms, int max_marks);

/* which rounds of the last folding sumcheck ran as int8 matrix-core GEMMs (bit i-1 = round i; lf_sv_rounds.h) -- test hook.
 * ABI 5: the mask only, on both rings (see LFHIP_ABI_VERSION above for the old packing) */
int lf_last_fold_paths(lf_ctx *, unsigned *sv_round_mask);
/* diagnostic: cycle counters of the int8 commit kernel's producer / multiplier waves when the library was built with -DLF_I8_PROF (tools/i8_prof.py);
 * 64 words, all zero in a normal build */
int lf_debug_i8_prof(uint64_t *out64);
/* how many rounds of the last linearization sumcheck ran in the split eq form -- test hook */
int lf_last_lin_split_rounds(lf_ctx *, unsigned *rounds);
/* which table rounds of the last folding sumcheck ran in the split eq form (three lazy products per table, message completed on the host;
 * bit i-1 = round i) -- test hook */
int lf_last_fold_split_rounds(lf_ctx *, unsigned *round_mask);

/* NIFSVerifier::verify (nifs.rs:117-163) on the host: O(proof size), NO GPU and no lf_ctx needed.  The CCS enters only
 * through its shape (lf_params), the multisets S (S_off[q+1], S_idx) and the coefficients c (q ring elements) -- the
 * verifier never touches the matrices (linearization.rs:220-243).  `t` is a transcript of the same ring in the state the
 * prover's transcript had before lf_fold_step.  Returns LF_OK and the folded LCCCS, or LF_ERR_REJECT with *failed_stage =
 * 1 linearization sumcheck, 2 linearization claim, 3 / 4 left / right decomposition recomposition, 5 folding sumcheck,
 * 6 folding claim.  Uses the default ring tables of the ring. */
int lf_verify_host(int ring, const lf_params *, const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *c, lf_transcript *t,
                   const uint64_t *acc_lcccs, const uint64_t *cm_i_cccs, const uint64_t *proof, uint64_t *lcccs_out, int *failed_stage);

/* ---- wire format -------------------------------------------------------------------------------------
 * Bytes of an LFProof as the reference's derived CanonicalSerialize writes them with Compress::Yes (nifs.rs:28-34 and the
 * structs it contains; round trip in nifs/folding/tests/mod.rs:656-680): fields in declaration order, every Vec prefixed by
 * its u64 little-endian length, ring elements as their d base-field words (8 bytes LE each, canonical) in the flat order of
 * this ABI.  The ring-element serializer itself is in the un-vendored stark-rings crate: that part of the layout is an
 * assumption (parity unpinned), see lf_wire.cpp.  `proof` is the flat proof of lf_fold_step.
 * serialize: non-canonical word or cap < lf_proof_wire_size -> LF_ERR_INVALID.  deserialize: any length prefix that differs
 * from the parameters, a non-canonical word, truncated or trailing bytes -> LF_ERR_INVALID (Validate::Yes). */
size_t lf_proof_wire_size(const lf_params *, int ring);
int lf_proof_serialize(const lf_params *, int ring, const uint64_t *proof, uint8_t *out, size_t cap);
int lf_proof_deserialize(const lf_params *, int ring, const uint8_t *in, size_t len, uint64_t *proof);

#ifdef __cplusplus
}
#endif
#endif

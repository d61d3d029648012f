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
    LFPLUS_E_SMALL_N = -5     /* kappa*k*d*l*d >= n: the reference's split() panics ("small n unsupported") */
};

typedef struct lfplus_ctx lfplus_ctx;
int lfplus_ctx_create(int device, lfplus_ctx **out);
void lfplus_ctx_destroy(lfplus_ctx *ctx);
const char *lfplus_last_error(const lfplus_ctx *ctx);

/* Ajtai matrix A (kappa x n ring elements, row-major, coefficient form); stays resident in HBM.  kappa <= 64. */
int lfplus_set_matrix(lfplus_ctx *ctx, const uint64_t *A, uint32_t kappa, uint64_t n);
/* witness vector f (n ring elements); stays resident */
int lfplus_set_witness(lfplus_ctx *ctx, const uint64_t *f, uint64_t n);

/* RgInstance::from_f on the resident (A, f).  b >= 2 (digits must land in (-8, 8): b <= 14), 1 <= k <= 16, 1 <= l <= 64.
 * Results stay on the device until lfplus_rg_read. */
int lfplus_rg_from_f(lfplus_ctx *ctx, uint64_t b, uint32_t k, uint32_t l);
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

/* Matrix::try_mul_vec: out (kappa*16 words) = A * v for a general vector of n ring elements (host pointer) */
int lfplus_commit(lfplus_ctx *ctx, const uint64_t *v, uint64_t n, uint64_t *out);

/* utils::tensor(r) over the base field: out has 2^n words (n <= 28); utils::tensor_product: out has m*n words (or the non-empty side) */
int lfplus_tensor(lfplus_ctx *ctx, const uint64_t *r, uint32_t n, uint64_t *out);
int lfplus_tensor_product(lfplus_ctx *ctx, const uint64_t *a, uint64_t m, const uint64_t *b, uint64_t n, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif

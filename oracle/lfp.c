/* lfp.c -- see lfp.h (ORACLE, test infrastructure only) */
#include "lfp.h"
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
typedef unsigned __int128 u128;
#define P LFP_P
#define D LFP_D

static inline u64 fadd(u64 a, u64 b) { u128 s = (u128)a + b; return (u64)(s >= P ? s - P : s); }
static inline u64 fsub(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
static inline u64 fmul(u64 a, u64 b) { return (u64)(((u128)a * b) % P); }
static inline u64 from_i64(int64_t v) { return v >= 0 ? (u64)v % P : (P - ((u64)(-v) % P)) % P; }

void lfp_ring_mul(const u64 *a, const u64 *b, u64 *out) {
    u64 r[D] = {0};
    for (int i = 0; i < D; i++) {
        if (!a[i]) continue;
        for (int j = 0; j < D; j++) {
            u64 pr = fmul(a[i], b[j]);
            if (i + j < D) r[i + j] = fadd(r[i + j], pr);
            else r[i + j - D] = fsub(r[i + j - D], pr); /* X^16 = -1 */
        }
    }
    memcpy(out, r, sizeof(r));
}
/* tensor_product (utils.rs:45-66): result[i*n + j] = a[i] * b[j]; an empty side returns the other */
void lfp_tensor_product(const u64 *a, size_t m, const u64 *b, size_t n, u64 *out) {
    if (!m) { memcpy(out, b, n * sizeof(u64)); return; }
    if (!n) { memcpy(out, a, m * sizeof(u64)); return; }
    for (size_t i = 0; i < m; i++)
        for (size_t j = 0; j < n; j++) out[i * n + j] = fmul(a[i], b[j]);
}
/* tensor (utils.rs:68-83): result = [1]; for r_i: result = tensor_product(result, [1 - r_i, r_i]) */
void lfp_tensor(const u64 *r, size_t n, u64 *out) {
    size_t len = 1;
    u64 *cur = (u64 *)malloc(sizeof(u64) << n), *nxt = (u64 *)malloc(sizeof(u64) << n);
    cur[0] = 1;
    for (size_t i = 0; i < n; i++) {
        u64 term[2] = {fsub(1, r[i] % P), r[i] % P};
        lfp_tensor_product(cur, len, term, 2, nxt);
        len *= 2;
        u64 *t = cur; cur = nxt; nxt = t;
    }
    memcpy(out, cur, len * sizeof(u64));
    free(cur); free(nxt);
}
/* stark_rings::balanced_decomposition as recollected (the rule of lfo_ring.c mode 0): centred lift, truncating remainder, |rem| <= b/2
 * kept, otherwise rem -+ b with carry +-1 */
void lfp_balanced_digits(u64 v, u64 base, unsigned digits, int64_t *out) {
    __int128 b = (__int128)base, half = b / 2;
    __int128 cur = v <= (P - 1) / 2 ? (__int128)v : (__int128)v - (__int128)P;
    for (unsigned k = 0; k < digits; k++) {
        __int128 rem = cur % b, q = cur / b;
        __int128 ar = rem < 0 ? -rem : rem;
        if (ar > half) {
            if (rem < 0) { rem += b; q -= 1; }
            else { rem -= b; q += 1; }
        }
        out[k] = (int64_t)rem;
        cur = q;
    }
}
/* exp(a) = sgn(a) X^a in Z_p[X]/(X^d + 1) (LatticeFold+ section 4.1): X^a for a >= 0, X^(d+a) for a < 0 -- X^(-|a|) = -X^(d-|a|) */
int lfp_exp(int64_t c, u64 *out) {
    if (c <= -(D / 2) || c >= D / 2) return -1;
    memset(out, 0, D * sizeof(u64));
    out[c >= 0 ? c : D + c] = 1;
    return 0;
}
void lfp_commit(const u64 *A, uint32_t kappa, size_t n, const u64 *f, u64 *out) {
    for (uint32_t i = 0; i < kappa; i++) {
        u64 acc[D] = {0}, t[D];
        for (size_t j = 0; j < n; j++) {
            lfp_ring_mul(A + ((size_t)i * n + j) * D, f + j * D, t);
            for (int c = 0; c < D; c++) acc[c] = fadd(acc[c], t[c]);
        }
        memcpy(out + (size_t)i * D, acc, sizeof(acc));
    }
}
int lfp_rg_from_f(const u64 *f, size_t n, const u64 *A, uint32_t kappa, u64 b, uint32_t k, uint32_t l, int8_t *Df, u64 *comMf, u64 *tau,
                  u64 *cm_f, u64 *C_Mf, u64 *cm_mtau) {
    /* cfs -> dec -> D_f (rgchk.rs:263-284) */
    int64_t dg[64];
    for (size_t ni = 0; ni < n; ni++)
        for (int di = 0; di < D; di++) {
            lfp_balanced_digits(f[ni * D + di], b, k, dg);
            for (uint32_t ki = 0; ki < k; ki++) {
                if (dg[ki] <= -(D / 2) || dg[ki] >= D / 2) return -1;
                Df[((size_t)ki * n + ni) * D + di] = (int8_t)dg[ki];
            }
        }
    /* M_f = exp(D_f); comM_f[k_i] = A * M_f[k_i]  (rgchk.rs:286-303): multiplying by the monomial X^e is a negacyclic shift */
    memset(comMf, 0, (size_t)k * kappa * D * D * sizeof(u64));
    for (uint32_t ki = 0; ki < k; ki++)
        for (uint32_t i = 0; i < kappa; i++)
            for (size_t j = 0; j < n; j++) {
                const u64 *a = A + ((size_t)i * n + j) * D;
                for (int c = 0; c < D; c++) {
                    int d8 = Df[((size_t)ki * n + j) * D + c], e = d8 >= 0 ? d8 : D + d8;
                    u64 *o = comMf + (((size_t)ki * kappa + i) * D + c) * D;
                    for (int t = 0; t < D; t++) {
                        int s = t - e;
                        o[t] = s >= 0 ? fadd(o[t], a[s]) : fsub(o[t], a[s + D]);
                    }
                }
            }
    /* com = hconcat(comM_f): row i = [comM_f[0][i][0..d), comM_f[1][i][0..d), ..]; tau = split(com, n, d/2, l) (utils.rs:12-43) */
    size_t need = (size_t)kappa * k * D * l * D;
    if (need >= n) return -2; /* the reference panics when tau does not fit below n */
    size_t pos = 0;
    int64_t *digs = (int64_t *)malloc(sizeof(int64_t) * l);
    for (uint32_t i = 0; i < kappa; i++)
        for (uint32_t ki = 0; ki < k; ki++)
            for (int c = 0; c < D; c++) {
                const u64 *e = comMf + (((size_t)ki * kappa + i) * D + c) * D;
                /* gadget_decompose(d/2, l): element -> l elements, digit j of every coefficient; then their coefficients in order */
                for (uint32_t j = 0; j < l; j++)
                    for (int t = 0; t < D; t++) {
                        lfp_balanced_digits(e[t], D / 2, l, digs);
                        tau[pos + (size_t)j * D + t] = from_i64(digs[j]);
                    }
                pos += (size_t)l * D;
            }
    free(digs);
    for (size_t j = pos; j < n; j++) tau[j] = 0;
    /* m_tau = exp(tau); cm_f = A f; C_Mf = A * R::from(tau); cm_mtau = A * m_tau (rgchk.rs:305-320) */
    lfp_commit(A, kappa, n, f, cm_f);
    for (uint32_t i = 0; i < kappa; i++) {
        u64 acc1[D] = {0}, acc2[D] = {0};
        for (size_t j = 0; j < n; j++) {
            const u64 *a = A + ((size_t)i * n + j) * D;
            u64 tj = tau[j];
            int64_t tc = tj <= (P - 1) / 2 ? (int64_t)tj : -(int64_t)(P - tj);
            if (tc <= -(D / 2) || tc >= D / 2) return -1;
            int e = tc >= 0 ? (int)tc : D + (int)tc;
            for (int t = 0; t < D; t++) {
                acc1[t] = fadd(acc1[t], fmul(a[t], tj));
                int s = t - e;
                acc2[t] = s >= 0 ? fadd(acc2[t], a[s]) : fsub(acc2[t], a[s + D]);
            }
        }
        memcpy(C_Mf + (size_t)i * D, acc1, sizeof(acc1));
        memcpy(cm_mtau + (size_t)i * D, acc2, sizeof(acc2));
    }
    return 0;
}
void lfp_splitmix_fill(u64 seed, u64 start, size_t count, u64 *out) {
    for (size_t i = 0; i < count; i++) {
        u64 z = seed + (start + (u64)i + 1) * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        out[i] = z % P;
    }
}

/* ---- Decomp::decompose (crates/latticefold-plus/src/decomp.rs:32-99): transcript-free --------------------------------------------
 * F = f.decompose_to_vec(B, 2).transpose() -> (F0, F1), f = F0 + B F1 coefficient-wise with balanced digits;
 * v_i = [ (mle(F_i)(r_a), mle(F_i)(r_b)), then for every matrix M_j: (mle(M_j F_i)(r_a), mle(M_j F_i)(r_b)) ];  C_i = A F_i.
 * MLE evaluation = fix_variables, variable 0 (index bit 0) first: new[j] = old[2j] + r (old[2j+1] - old[2j]) with RING products (the
 * point's coordinates are ring elements).  Matrices in CSR with ring-element coefficients (stark_rings_linalg::SparseMatrix rows of
 * (value, column)); every matrix has n rows.  n must be a power of two (nvars = log2(A.ncols)). */
static void ring_add(u64 *d, const u64 *a, const u64 *b) { for (int i = 0; i < D; i++) d[i] = fadd(a[i], b[i]); }
static void ring_sub(u64 *d, const u64 *a, const u64 *b) { for (int i = 0; i < D; i++) d[i] = fsub(a[i], b[i]); }
static void mle_eval(const u64 *tab, size_t n, unsigned nvars, const u64 *r /* nvars x 16 */, u64 *out) {
    u64 *cur = (u64 *)malloc(n * D * sizeof(u64));
    memcpy(cur, tab, n * D * sizeof(u64));
    size_t len = n;
    for (unsigned k = 0; k < nvars; k++) {
        for (size_t j = 0; j < len / 2; j++) {
            u64 diff[D], pr[D];
            ring_sub(diff, cur + (2 * j + 1) * D, cur + (2 * j) * D);
            lfp_ring_mul(r + (size_t)k * D, diff, pr);
            ring_add(cur + j * D, cur + (2 * j) * D, pr);
        }
        len /= 2;
    }
    memcpy(out, cur, D * sizeof(u64));
    free(cur);
}
int lfp_decompose(const u64 *f, size_t n, const u64 *A, uint32_t kappa, u64 B, const u64 *r_a, const u64 *r_b, uint32_t nm,
                  const uint32_t *const *rowptr, const uint32_t *const *col, const u64 *const *val, u64 *F0, u64 *F1, u64 *C0, u64 *C1, u64 *v0,
                  u64 *v1) {
    if (!n || (n & (n - 1))) return -1;
    unsigned nvars = 0;
    while (((size_t)1 << nvars) < n) nvars++;
    int64_t dg[2];
    for (size_t i = 0; i < n * D; i++) {
        lfp_balanced_digits(f[i], B, 2, dg);
        F0[i] = from_i64(dg[0]);
        F1[i] = from_i64(dg[1]);
    }
    u64 *mv = (u64 *)malloc(n * D * sizeof(u64));
    for (int s = 0; s < 2; s++) {
        const u64 *Fi = s ? F1 : F0;
        u64 *v = s ? v1 : v0;
        mle_eval(Fi, n, nvars, r_a, v);
        mle_eval(Fi, n, nvars, r_b, v + D);
        for (uint32_t j = 0; j < nm; j++) {
            for (size_t row = 0; row < n; row++) {
                u64 acc[D] = {0}, t[D];
                for (uint32_t k = rowptr[j][row]; k < rowptr[j][row + 1]; k++) {
                    lfp_ring_mul(val[j] + (size_t)k * D, Fi + (size_t)col[j][k] * D, t);
                    ring_add(acc, acc, t);
                }
                memcpy(mv + row * D, acc, sizeof(acc));
            }
            mle_eval(mv, n, nvars, r_a, v + (size_t)(1 + j) * 2 * D);
            mle_eval(mv, n, nvars, r_b, v + (size_t)(1 + j) * 2 * D + D);
        }
        lfp_commit(A, kappa, n, Fi, s ? C1 : C0);
    }
    free(mv);
    return 0;
}

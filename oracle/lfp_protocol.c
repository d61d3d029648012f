/* lfp_protocol.c -- CPU ORACLE (test infrastructure only, see lfp.h) of the transcript-driven part of LatticeFold+ on the Frog ring:
 *
 *   PoseidonTranscript<RqPoly>          crates/latticefold-plus/src/transcript.rs:20-78   (parameters: cyclotomic-rings/src/rings/poseidon/frog.rs)
 *   utils::short_challenge              src/utils.rs:87-101
 *   In::set_check / Out::verify         src/setchk.rs:65-262 / 266-340                     (monomial set check, batched sumcheck)
 *   Rg::range_check / Dcom::verify      src/rgchk.rs:81-186 / 193-258
 *   MLSumcheck::{prove,verify}_as_subprotocol   crates/latticefold/src/utils/sumcheck.rs:53-110, utils/sumcheck/{prover,verifier}.rs
 *
 * The coefficient ring RqPoly has BaseRing = F_p (extension degree 1): a challenge is ONE field element, R::from(challenge) is a constant
 * polynomial, and every table of the set-check sumcheck holds constants (ev(m, beta), its square, eq(c, .)) -- products of constants are
 * products in F_p, so the sumcheck is restated over F_p words and its messages are embedded as constant ring elements where the reference
 * absorbs / returns ring elements.  The evaluations of Step 3 (e, b) and of the range check (v, a, b, c) are genuine ring elements.
 *
 * Poseidon table: the reference's Frog table holds the SAME 64-bit literals as the Goldilocks one, embedded with Fq::from(i128), i.e. the
 * Grain-LFSR constants of a 64-bit prime reduced mod p_frog (tests/golden/kats.json "poseidon_frog_params": identity of the literal lists +
 * checksums).  They are regenerated here, not copied.  Sponge: ark-crypto-primitives 0.4.0 PoseidonSponge as in lfo_poseidon.c (KAT-pinned
 * on Goldilocks by transcript/poseidon.rs:85-142; the Frog instance has no transcript KAT in the reference -- the code path is the same).
 *
 * PARITY STATUS: unpinned like the rest of this slice (lfp.h).  psi is DERIVED from exp by its defining property ct(psi * exp(a)) = a for
 * -d/2 < a < d/2 (LatticeFold+ Lemma 2.2): psi = sum_{0 < i < d/2} i (X^i - X^(d-i)); DenseMultilinearExtension fixes the low variable
 * first (sumcheck/prover.rs:112-123), SparseMatrix rows are (coefficient, column) lists. */
#include "lfp.h"
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
typedef unsigned __int128 u128;
#define P LFP_P
#define D LFP_D
#define W 24
#define RATE 20
#define CAP 4
#define RF 8
#define RP 22

static inline u64 fadd(u64 a, u64 b) { u128 s = (u128)a + b; return (u64)(s >= P ? s - P : s); }
static inline u64 fsub(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
static inline u64 fmul(u64 a, u64 b) { return (u64)(((u128)a * b) % P); }
static u64 fpow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; } return r; }

/* ---- Poseidon parameters: Grain LFSR for a 64-bit prime, width 24, 8 + 22 rounds (as lfo_poseidon.c), reduced mod p_frog -------------- */
static u64 g_ark[(RF + RP) * W], g_mds[W * W];
static int g_init = 0;
typedef struct { unsigned char st[80]; int head; } grain;
static int grain_update(grain *g) {
    int h = g->head;
    unsigned char nb = g->st[(h + 62) % 80] ^ g->st[(h + 51) % 80] ^ g->st[(h + 38) % 80] ^ g->st[(h + 23) % 80] ^ g->st[(h + 13) % 80] ^ g->st[h];
    g->st[h] = nb;
    g->head = (h + 1) % 80;
    return nb;
}
static void grain_put(grain *g, int lo, int hi, u64 v) { for (int i = hi; i >= lo; i--) { g->st[i] = v & 1; v >>= 1; } }
static u64 grain_bits64(grain *g) {
    u64 v = 0;
    for (int i = 0; i < 64; i++) {
        int nb = grain_update(g);
        while (!nb) { grain_update(g); nb = grain_update(g); }
        v = (v << 1) | (u64)grain_update(g);
    }
    return v;
}
static void poseidon_init(void) {
    if (g_init) return;
    grain g;
    memset(&g, 0, sizeof(g));
    g.st[1] = 1;
    grain_put(&g, 6, 17, 64);
    grain_put(&g, 18, 29, W);
    grain_put(&g, 30, 39, RF);
    grain_put(&g, 40, 49, RP);
    for (int i = 50; i < 80; i++) g.st[i] = 1;
    for (int i = 0; i < 160; i++) grain_update(&g);
    const u64 PG = 0xFFFFFFFF00000001ULL;   /* the table was generated for the Goldilocks prime */
    for (int i = 0; i < (RF + RP) * W; i++) {
        u64 v;
        do v = grain_bits64(&g); while (v >= PG);
        g_ark[i] = v % P;
    }
    u64 xs[W], ys[W];
    for (int i = 0; i < W; i++) xs[i] = grain_bits64(&g) % PG;
    for (int i = 0; i < W; i++) ys[i] = grain_bits64(&g) % PG;
    for (int i = 0; i < W; i++)
        for (int j = 0; j < W; j++) {   /* Cauchy matrix over Goldilocks: 1 / (x_i + y_j) */
            u128 sm = ((u128)xs[i] + ys[j]) % PG;
            u64 r = 1, bs = (u64)sm;
            for (u64 e = PG - 2; e; e >>= 1) { if (e & 1) r = (u64)(((u128)r * bs) % PG); bs = (u64)(((u128)bs * bs) % PG); }
            g_mds[i * W + j] = r % P;
        }
    g_init = 1;
}
void lfp_poseidon_params(u64 *ark, u64 *mds) {
    poseidon_init();
    memcpy(ark, g_ark, sizeof(g_ark));
    memcpy(mds, g_mds, sizeof(g_mds));
}
static inline u64 pow7(u64 x) { u64 x2 = fmul(x, x), x4 = fmul(x2, x2); return fmul(fmul(x4, x2), x); }
void lfp_poseidon_permute(u64 *st) {
    poseidon_init();
    u64 nw[W];
    for (int r = 0; r < RF + RP; r++) {
        for (int i = 0; i < W; i++) st[i] = fadd(st[i], g_ark[r * W + i]);
        if (r < RF / 2 || r >= RF / 2 + RP) for (int i = 0; i < W; i++) st[i] = pow7(st[i]);
        else st[0] = pow7(st[0]);
        for (int i = 0; i < W; i++) {
            u64 a = 0;
            for (int j = 0; j < W; j++) a = fadd(a, fmul(st[j], g_mds[i * W + j]));
            nw[i] = a;
        }
        memcpy(st, nw, sizeof(nw));
    }
}

/* ---- duplex sponge + transcript (transcript.rs:20-78) --------------------------------------------------------------------------------- */
struct lfp_tr { u64 st[W]; int squeezing, idx; };
lfp_tr *lfp_tr_new(void) { poseidon_init(); return (lfp_tr *)calloc(1, sizeof(lfp_tr)); }
void lfp_tr_free(lfp_tr *t) { free(t); }
lfp_tr *lfp_tr_clone(const lfp_tr *t) { lfp_tr *c = (lfp_tr *)malloc(sizeof(*c)); memcpy(c, t, sizeof(*c)); return c; }
static void absorb_fq(lfp_tr *t, const u64 *x, size_t n) {
    if (!n) return;
    int idx;
    if (!t->squeezing) { idx = t->idx; if (idx == RATE) { lfp_poseidon_permute(t->st); idx = 0; } }
    else { lfp_poseidon_permute(t->st); idx = 0; }
    for (;;) {
        if ((size_t)idx + n <= RATE) {
            for (size_t i = 0; i < n; i++) t->st[CAP + idx + i] = fadd(t->st[CAP + idx + i], x[i] % P);
            t->squeezing = 0;
            t->idx = idx + (int)n;
            return;
        }
        size_t take = RATE - idx;
        for (size_t i = 0; i < take; i++) t->st[CAP + idx + i] = fadd(t->st[CAP + idx + i], x[i] % P);
        lfp_poseidon_permute(t->st);
        x += take; n -= take; idx = 0;
    }
}
static void squeeze_fq(lfp_tr *t, u64 *out, size_t n) {
    int idx;
    if (!t->squeezing) { lfp_poseidon_permute(t->st); idx = 0; }
    else { idx = t->idx; if (idx == RATE) { lfp_poseidon_permute(t->st); idx = 0; } }
    for (;;) {
        if ((size_t)idx + n <= RATE) {
            memcpy(out, t->st + CAP + idx, n * sizeof(u64));
            t->squeezing = 1;
            t->idx = idx + (int)n;
            return;
        }
        size_t take = RATE - idx;
        memcpy(out, t->st + CAP + idx, take * sizeof(u64));
        if (n != RATE) lfp_poseidon_permute(t->st);
        out += take; n -= take; idx = 0;
    }
}
/* Transcript::absorb(R): the 16 coefficients of each element (extension degree 1: one base-prime-field word per coefficient) */
void lfp_tr_absorb(lfp_tr *t, const u64 *e, size_t count) { for (size_t i = 0; i < count; i++) absorb_fq(t, e + i * D, D); }
static void absorb_const(lfp_tr *t, u64 c) { u64 e[D] = {0}; e[0] = c % P; lfp_tr_absorb(t, e, 1); }   /* absorb(&R::from(c)) */
/* get_challenge (transcript.rs:44-53): squeeze extension_degree = 1 word, absorb it back */
u64 lfp_tr_challenge(lfp_tr *t) { u64 c; squeeze_fq(t, &c, 1); absorb_fq(t, &c, 1); return c; }
/* squeeze_bytes (ark PoseidonSponge): 7 low little-endian bytes per element of a 64-bit prime field, ceil(n / 7) elements */
void lfp_tr_squeeze_bytes(lfp_tr *t, size_t n, uint8_t *out) {
    size_t ne = (n + 6) / 7;
    u64 *e = (u64 *)malloc(ne * sizeof(u64));
    squeeze_fq(t, e, ne);
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(e[i / 7] >> (8 * (i % 7)));
    free(e);
}
/* utils::short_challenge(128, ..) (utils.rs:87-101): u = 2^(128/16) = 256, coefficient = byte % u - u/2.  The same map as
 * FrogChallengeSet::short_challenge_from_random_bytes (cyclotomic-rings/src/rings/frog.rs:35-55; KAT :66-96) */
void lfp_short_challenge_from_bytes(const uint8_t *bs, u64 *out) {
    for (int i = 0; i < D; i++) { int v = (int)bs[i] % 256 - 128; out[i] = v >= 0 ? (u64)v : P - (u64)(-v); }
}
void lfp_short_challenge(lfp_tr *t, u64 *out) { uint8_t bs[D]; lfp_tr_squeeze_bytes(t, D, bs); lfp_short_challenge_from_bytes(bs, out); }

/* ---- ring helpers ------------------------------------------------------------------------------------------------------------------ */
static void radd(u64 *a, const u64 *b) { for (int i = 0; i < D; i++) a[i] = fadd(a[i], b[i]); }
static void rscale_add(u64 *acc, const u64 *e, u64 s) { for (int i = 0; i < D; i++) acc[i] = fadd(acc[i], fmul(e[i], s)); }
static u64 ev(const u64 *r, u64 x) { u64 acc = 0, pw = 1; for (int i = 0; i < D; i++) { acc = fadd(acc, fmul(r[i], pw)); pw = fmul(pw, x); } return acc; }   /* setchk.rs:47-59 */
static u64 ct_psi_mul(const u64 *b) {   /* ct(psi * b), psi = sum_{0<i<d/2} i (X^i - X^(d-i)): ct picks -psi_{d-j} b_j */
    u64 acc = 0;
    for (int i = 1; i < D / 2; i++) {
        /* psi_i = i pairs with b_{d-i} (X^i X^(d-i) = -1): contributes -i b_{d-i};  psi_{d-i} = -i pairs with b_i: contributes +i b_i */
        acc = fadd(acc, fmul((u64)i, b[i]));
        acc = fsub(acc, fmul((u64)i, b[D - i]));
    }
    return acc;
}
void lfp_psi(u64 *out) { memset(out, 0, D * sizeof(u64)); for (int i = 1; i < D / 2; i++) { out[i] = (u64)i; out[D - i] = P - (u64)i; } }
u64 lfp_ct_psi_mul(const u64 *b) { return ct_psi_mul(b); }

/* build_eq_x_r (latticefold utils/sumcheck/utils.rs:100-170) over F_p: eq[i] = prod_j (bit_j(i) ? c_j : 1 - c_j), bit 0 <-> c[0] */
static u64 *build_eq(const u64 *c, unsigned nv) {
    size_t n = (size_t)1 << nv;
    u64 *eq = (u64 *)malloc(n * sizeof(u64));
    eq[0] = 1;
    for (unsigned j = 0; j < nv; j++) {
        size_t half = (size_t)1 << j;
        for (size_t i = 0; i < half; i++) { u64 v = eq[i], hi = fmul(v, c[j]); eq[i + half] = hi; eq[i] = fsub(v, hi); }
    }
    return eq;
}
static u64 eq_eval(const u64 *x, const u64 *y, unsigned nv) {   /* utils.rs:78-92 */
    u64 r = 1;
    for (unsigned i = 0; i < nv; i++) { u64 xy = fmul(x[i], y[i]); r = fmul(r, fadd(fsub(fsub(fadd(xy, xy), x[i]), y[i]), 1)); }
    return r;
}
/* MLE of a vector of n ring elements (zero-padded to 2^nv) at a point of constants: sum_i eq(r, i) v[i] */
static void mle_eval_ring(const u64 *v, size_t n, const u64 *eqr, u64 *out) {
    memset(out, 0, D * sizeof(u64));
    for (size_t i = 0; i < n; i++) rscale_add(out, v + i * D, eqr[i]);
}
/* CSR matrix with ring coefficients times a vector of ring elements (SparseMatrix::try_mul_vec) */
typedef struct { size_t nrows; const uint32_t *rowptr, *col; const u64 *val; } csr;
static void spmv_ring(const csr *m, const u64 *v, u64 *out /* nrows*16 */) {
    for (size_t r = 0; r < m->nrows; r++) {
        u64 acc[D] = {0}, t[D];
        for (uint32_t k = m->rowptr[r]; k < m->rowptr[r + 1]; k++) { lfp_ring_mul(m->val + (size_t)k * D, v + (size_t)m->col[k] * D, t); radd(acc, t); }
        memcpy(out + r * D, acc, sizeof(acc));
    }
}

/* ---- scalar sumcheck (utils/sumcheck.rs:53-80, sumcheck/prover.rs:56-162): tables of F_p words, messages as constant ring elements --- */
typedef u64 (*comb_fn)(const u64 *vals, void *ctx);
static void sumcheck_prove(lfp_tr *tr, u64 **tab, unsigned ntab, unsigned nv, unsigned deg, comb_fn comb, void *ctx, u64 *msgs /* nv*(deg+1)*16 */, u64 *point) {
    absorb_const(tr, nv);
    absorb_const(tr, deg);
    size_t n = (size_t)1 << nv;
    u64 *vals = (u64 *)malloc(ntab * sizeof(u64)), *step = (u64 *)malloc(ntab * sizeof(u64));
    for (unsigned rnd = 0; rnd < nv; rnd++) {
        size_t half = n >> 1;
        u64 sums[8] = {0};
        for (size_t b = 0; b < half; b++) {
            for (unsigned t = 0; t < ntab; t++) { vals[t] = tab[t][2 * b]; step[t] = fsub(tab[t][2 * b + 1], tab[t][2 * b]); }
            sums[0] = fadd(sums[0], comb(vals, ctx));
            for (unsigned x = 1; x <= deg; x++) {
                for (unsigned t = 0; t < ntab; t++) vals[t] = fadd(vals[t], step[t]);
                sums[x] = fadd(sums[x], comb(vals, ctx));
            }
        }
        u64 *m = msgs + (size_t)rnd * (deg + 1) * D;
        memset(m, 0, (size_t)(deg + 1) * D * sizeof(u64));
        for (unsigned x = 0; x <= deg; x++) m[x * D] = sums[x];
        lfp_tr_absorb(tr, m, deg + 1);
        u64 r = lfp_tr_challenge(tr);
        absorb_const(tr, r);
        point[rnd] = r;
        for (unsigned t = 0; t < ntab; t++)    /* fix_variables: new[j] = old[2j] + r (old[2j+1] - old[2j]) */
            for (size_t b = 0; b < half; b++) tab[t][b] = fadd(tab[t][2 * b], fmul(r, fsub(tab[t][2 * b + 1], tab[t][2 * b])));
        n = half;
    }
    free(vals); free(step);
}
/* interpolate_uni_poly (sumcheck/verifier.rs:141-257) at r through the values at x = 0..deg */
static u64 interpolate(const u64 *y, unsigned cnt, u64 r) {
    u64 res = 0;
    for (unsigned i = 0; i < cnt; i++) {
        u64 num = 1, den = 1;
        for (unsigned j = 0; j < cnt; j++) {
            if (j == i) continue;
            num = fmul(num, fsub(r, j));
            den = fmul(den, fsub(i, j));
        }
        res = fadd(res, fmul(y[i], fmul(num, fpow(den, P - 2))));
    }
    return res;
}
/* verify_as_subprotocol (sumcheck.rs:84-104): returns 0 and (point, expected evaluation), or < 0.  Messages must be constant ring elements
 * here (a non-constant message cannot pass: the honest messages of these sumchecks are constants). */
static int sumcheck_verify(lfp_tr *tr, unsigned nv, unsigned deg, u64 claimed, const u64 *msgs, u64 *point, u64 *expected) {
    absorb_const(tr, nv);
    absorb_const(tr, deg);
    u64 cur = claimed;
    for (unsigned rnd = 0; rnd < nv; rnd++) {
        const u64 *m = msgs + (size_t)rnd * (deg + 1) * D;
        lfp_tr_absorb(tr, m, deg + 1);
        u64 r = lfp_tr_challenge(tr);
        absorb_const(tr, r);
        point[rnd] = r;
        u64 y[8];
        for (unsigned x = 0; x <= deg; x++) {
            for (int c = 1; c < D; c++) if (m[x * D + c]) return -2;
            y[x] = m[x * D];
        }
        if (fadd(y[0], y[1]) != cur) return -1;   /* p(0) + p(1) = claim */
        cur = interpolate(y, deg + 1, r);
    }
    *expected = cur;
    return 0;
}

/* ---- set check (setchk.rs) --------------------------------------------------------------------------------------------------------- */
typedef struct { unsigned nmat, ncols, nvec; const u64 *alpha; int have_rc; u64 rc; } sc_ctx;
/* comb_fn of setchk.rs:160-197, literally: without the batching challenge (a single matrix set) the closure RETURNS after the first matrix
 * set -- vector sets then do not enter the sumcheck polynomial at all */
static u64 sc_comb(const u64 *vals, void *vctx) {
    const sc_ctx *c = (const sc_ctx *)vctx;
    u64 lc = 0, rcp = 1;
    for (unsigned i = 0; i < c->nmat; i++) {
        unsigned s = i * (2 * c->ncols + 1);
        u64 res = 0, ap = 1;
        for (unsigned j = 0; j < c->ncols; j++) {
            res = fadd(res, fmul(fsub(fmul(vals[s + 2 * j], vals[s + 2 * j]), vals[s + 2 * j + 1]), ap));
            ap = fmul(ap, c->alpha[i]);
        }
        res = fmul(res, vals[s + 2 * c->ncols]);
        if (!c->have_rc) return res;
        lc = fadd(lc, fmul(res, rcp));
        rcp = fmul(rcp, c->rc);
    }
    for (unsigned i = 0; i < c->nvec; i++) {
        unsigned s = c->nmat * (2 * c->ncols + 1) + 3 * i;
        u64 res = fmul(fsub(fmul(vals[s], vals[s]), vals[s + 1]), c->alpha[c->nmat + i]);
        res = fmul(res, vals[s + 2]);
        if (!c->have_rc) return res;
        lc = fadd(lc, fmul(res, rcp));
        rcp = fmul(rcp, c->rc);
    }
    return lc;
}
/* In::set_check.  msets: nmat matrices of n x ncols ring elements (dense, row-major; a zero element = an absent sparse entry), vsets: nvec
 * vectors of n ring elements; n = 2^nvars rows.  M: nM CSR matrices (n x n, ring coefficients) for the M_i f evaluations of Step 3.
 * Outputs: r (nvars words), msgs (nvars * 4 ring elements), e ((1 + nM) * nmat * ncols ring elements: e[q][set][col]), b (nvec ring elements). */
int lfp_set_check(lfp_tr *tr, unsigned nvars, const u64 *msets, unsigned nmat, unsigned ncols, const u64 *vsets, unsigned nvec, unsigned nM,
                  const uint32_t *const *rowptr, const uint32_t *const *col, const u64 *const *val, u64 *r_out, u64 *msgs, u64 *e_out, u64 *b_out) {
    if (nmat < 1) return -1;   /* "Currently requires k >= 1 monomial matrices sets" */
    size_t n = (size_t)1 << nvars;
    unsigned ntab = nmat * (2 * ncols + 1) + 3 * nvec;
    u64 **tab = (u64 **)malloc(ntab * sizeof(u64 *));
    u64 *alpha = (u64 *)malloc((nmat + nvec) * sizeof(u64)), *cch = (u64 *)malloc(nvars * sizeof(u64));
    unsigned ti = 0;
    for (unsigned i = 0; i < nmat + nvec; i++) {
        for (unsigned j = 0; j < nvars; j++) cch[j] = lfp_tr_challenge(tr);
        u64 beta = lfp_tr_challenge(tr);
        unsigned cols = i < nmat ? ncols : 1;
        for (unsigned j = 0; j < cols; j++) {
            u64 *mj = (u64 *)malloc(n * sizeof(u64)), *mp = (u64 *)malloc(n * sizeof(u64));
            for (size_t row = 0; row < n; row++) {
                const u64 *el = i < nmat ? msets + (((size_t)i * n + row) * ncols + j) * D : vsets + ((size_t)(i - nmat) * n + row) * D;
                mj[row] = ev(el, beta);
                mp[row] = fmul(mj[row], mj[row]);   /* m_prime_j = m_j * m_j (setchk.rs:112, 146) */
            }
            tab[ti++] = mj; tab[ti++] = mp;
        }
        tab[ti++] = build_eq(cch, nvars);
        alpha[i] = lfp_tr_challenge(tr);
    }
    sc_ctx ctx = {nmat, ncols, nvec, alpha, nmat > 1, 0};
    if (ctx.have_rc) ctx.rc = lfp_tr_challenge(tr);
    sumcheck_prove(tr, tab, ntab, nvars, 3, sc_comb, &ctx, msgs, r_out);
    for (unsigned t = 0; t < ntab; t++) free(tab[t]);
    free(tab); free(alpha); free(cch);
    /* Step 3: e[0] = columns of the sets at r; e[1 + q] = columns of M_q * set at r; b = vector sets at r */
    u64 *eqr = build_eq(r_out, nvars), *colv = (u64 *)malloc(n * D * sizeof(u64)), *mv = (u64 *)malloc(n * D * sizeof(u64));
    for (unsigned i = 0; i < nmat; i++)
        for (unsigned j = 0; j < ncols; j++) {
            for (size_t row = 0; row < n; row++) memcpy(colv + row * D, msets + (((size_t)i * n + row) * ncols + j) * D, D * sizeof(u64));
            mle_eval_ring(colv, n, eqr, e_out + ((size_t)i * ncols + j) * D);
            for (unsigned q = 0; q < nM; q++) {
                csr m = {n, rowptr[q], col[q], val[q]};
                spmv_ring(&m, colv, mv);
                mle_eval_ring(mv, n, eqr, e_out + (((size_t)(1 + q) * nmat + i) * ncols + j) * D);
            }
        }
    for (unsigned i = 0; i < nvec; i++) mle_eval_ring(vsets + (size_t)i * n * D, n, eqr, b_out + (size_t)i * D);
    free(eqr); free(colv); free(mv);
    /* absorb_evaluations (setchk.rs:342-353) */
    lfp_tr_absorb(tr, e_out, (size_t)(1 + nM) * nmat * ncols);
    lfp_tr_absorb(tr, b_out, nvec);
    return 0;
}
/* Out::verify (setchk.rs:266-340): 0 = accepted; -1/-2 sumcheck, -3 final evaluation mismatch.  r_out: the sumcheck point */
int lfp_set_check_verify(lfp_tr *tr, unsigned nvars, unsigned nmat, unsigned ncols, unsigned nvec, unsigned nM, const u64 *msgs, const u64 *e, const u64 *b,
                         u64 *r_out) {
    unsigned nclaims = nmat + nvec;
    u64 *cs = (u64 *)malloc((size_t)nclaims * nvars * sizeof(u64)), *beta = (u64 *)malloc(nclaims * sizeof(u64)), *alpha = (u64 *)malloc(nclaims * sizeof(u64));
    for (unsigned i = 0; i < nclaims; i++) {
        for (unsigned j = 0; j < nvars; j++) cs[(size_t)i * nvars + j] = lfp_tr_challenge(tr);
        beta[i] = lfp_tr_challenge(tr);
        alpha[i] = lfp_tr_challenge(tr);
    }
    int have_rc = nmat > 1;
    u64 rc = have_rc ? lfp_tr_challenge(tr) : 1, v;
    int rcode = sumcheck_verify(tr, nvars, 3, 0, msgs, r_out, &v);
    if (!rcode) {
        lfp_tr_absorb(tr, e, (size_t)(1 + nM) * nmat * ncols);
        lfp_tr_absorb(tr, b, nvec);
        u64 ver = 0, rcp = 1;
        for (unsigned i = 0; i < nmat; i++) {
            u64 eq = eq_eval(cs + (size_t)i * nvars, r_out, nvars), sum = 0, ap = 1, b2 = fmul(beta[i], beta[i]);
            for (unsigned j = 0; j < ncols; j++) {
                const u64 *ej = e + ((size_t)i * ncols + j) * D;
                u64 e1 = ev(ej, beta[i]), e2 = ev(ej, b2);
                sum = fadd(sum, fmul(fsub(fmul(e1, e1), e2), ap));
                ap = fmul(ap, alpha[i]);
            }
            ver = fadd(ver, fmul(fmul(eq, sum), rcp));
            rcp = fmul(rcp, rc);
        }
        for (unsigned i = 0; i < nvec; i++) {
            unsigned k = nmat + i;
            u64 eq = eq_eval(cs + (size_t)k * nvars, r_out, nvars), b2 = fmul(beta[k], beta[k]);
            u64 e1 = ev(b + (size_t)i * D, beta[k]), e2 = ev(b + (size_t)i * D, b2);
            ver = fadd(ver, fmul(fmul(fmul(eq, alpha[k]), fsub(fmul(e1, e1), e2)), rcp));
            rcp = fmul(rcp, rc);
        }
        if (ver != v) rcode = -3;
    }
    free(cs); free(beta); free(alpha);
    return rcode;
}

/* ---- range check (rgchk.rs:81-258) --------------------------------------------------------------------------------------------------- */
/* L instances; instance l: Mf (k matrices of n x 16 unit monomials, dense ring elements [k][n][16][16 words]), tau (n words), mtau (n ring
 * elements), f (n ring elements).  Outputs: the set check's (r, msgs, e ((1 + nM) * (L k) * 16 ring elements), b (L)) and per instance
 * v (16 words), a (1 + nM words), bb (1 + nM ring elements), c (1 + nM ring elements). */
int lfp_range_check(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, const u64 *const *Mf, const u64 *const *tau, const u64 *const *mtau, const u64 *const *f,
                    unsigned nM, const uint32_t *const *rowptr, const uint32_t *const *col, const u64 *const *val, u64 *r_out, u64 *msgs, u64 *e_out, u64 *b_out,
                    u64 *v_out, u64 *a_out, u64 *bb_out, u64 *c_out) {
    size_t n = (size_t)1 << nvars;
    /* sets: all instances' M_f matrices, then all instances' m_tau vectors (rgchk.rs:87-96) */
    u64 *msets = (u64 *)malloc((size_t)L * k * n * D * D * sizeof(u64)), *vsets = (u64 *)malloc((size_t)L * n * D * sizeof(u64));
    for (unsigned l = 0; l < L; l++) {
        memcpy(msets + (size_t)l * k * n * D * D, Mf[l], (size_t)k * n * D * D * sizeof(u64));
        memcpy(vsets + (size_t)l * n * D, mtau[l], n * D * sizeof(u64));
    }
    int rc = lfp_set_check(tr, nvars, msets, L * k, D, vsets, L, nM, rowptr, col, val, r_out, msgs, e_out, b_out);
    free(msets); free(vsets);
    if (rc) return rc;
    u64 *eqr = build_eq(r_out, nvars), *tmp = (u64 *)malloc(n * D * sizeof(u64)), *rt = (u64 *)malloc(n * D * sizeof(u64));
    for (unsigned l = 0; l < L; l++) {
        u64 *v = v_out + (size_t)l * D, *a = a_out + (size_t)l * (1 + nM), *bb = bb_out + (size_t)l * (1 + nM) * D, *c = c_out + (size_t)l * (1 + nM) * D;
        /* v: MLE of every coefficient of f (the transposed coefficient matrix) at r -- equal to the coefficients of c[0] */
        mle_eval_ring(f[l], n, eqr, c);
        memcpy(v, c, D * sizeof(u64));
        u64 a0 = 0;
        for (size_t i = 0; i < n; i++) a0 = fadd(a0, fmul(eqr[i], tau[l][i] % P));
        a[0] = a0;
        memcpy(bb, b_out + (size_t)l * D, D * sizeof(u64));
        for (unsigned q = 0; q < nM; q++) {
            csr m = {n, rowptr[q], col[q], val[q]};
            u64 ev16[D];
            memset(rt, 0, n * D * sizeof(u64));
            for (size_t i = 0; i < n; i++) rt[i * D] = tau[l][i] % P;
            spmv_ring(&m, rt, tmp);
            mle_eval_ring(tmp, n, eqr, ev16);
            a[1 + q] = ev16[0];                                   /* .ct() */
            spmv_ring(&m, mtau[l], tmp);
            mle_eval_ring(tmp, n, eqr, bb + (size_t)(1 + q) * D);
            spmv_ring(&m, f[l], tmp);
            mle_eval_ring(tmp, n, eqr, c + (size_t)(1 + q) * D);
        }
    }
    free(eqr); free(tmp); free(rt);
    /* absorb_evaluations (rgchk.rs:333-338): a as constants, then c */
    for (unsigned l = 0; l < L; l++) {
        for (unsigned i = 0; i < 1 + nM; i++) absorb_const(tr, a_out[(size_t)l * (1 + nM) + i]);
        lfp_tr_absorb(tr, c_out + (size_t)l * (1 + nM) * D, 1 + nM);
    }
    return 0;
}
/* Dcom::verify (rgchk.rs:193-258): 0 = accepted; -1..-3 set check; -4 ct(psi b) != a; -5 ct(psi sum d'^i u_i) != v / c */
int lfp_range_check_verify(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, unsigned nM, const u64 *msgs, const u64 *e, const u64 *b, const u64 *v,
                           const u64 *a, const u64 *bb, const u64 *c, u64 *r_out) {
    int rc = lfp_set_check_verify(tr, nvars, L * k, D, L, nM, msgs, e, b, r_out);
    if (rc) return rc;
    for (unsigned l = 0; l < L; l++) {
        for (unsigned i = 0; i < 1 + nM; i++) absorb_const(tr, a[(size_t)l * (1 + nM) + i]);
        lfp_tr_absorb(tr, c + (size_t)l * (1 + nM) * D, 1 + nM);
    }
    const u64 dprime = D / 2;
    for (unsigned l = 0; l < L; l++) {
        for (unsigned i = 0; i < 1 + nM; i++)
            if (ct_psi_mul(bb + ((size_t)l * (1 + nM) + i) * D) != a[(size_t)l * (1 + nM) + i]) return -4;
        for (unsigned ni = 0; ni < 1 + nM; ni++) {
            /* u_comb[t] = sum_{i<k} d'^i e[ni][k l + i][t] (ring elements, one per column t); v_rec[t] = ct(psi u_comb[t]) */
            for (unsigned t = 0; t < D; t++) {
                u64 uc[D] = {0}, pw = 1;
                for (unsigned i = 0; i < k; i++) {
                    rscale_add(uc, e + ((((size_t)ni * L * k) + (size_t)k * l + i) * D + t) * D, pw);
                    pw = fmul(pw, dprime);
                }
                u64 vr = ct_psi_mul(uc);
                u64 want = ni == 0 ? v[(size_t)l * D + t] : c[((size_t)l * (1 + nM) + ni) * D + t];
                if (vr != want) return -5;
            }
        }
    }
    return 0;
}

/* ---- ring-valued sumcheck (MLSumcheck over RqPoly tables): messages are ring elements, challenges constants ------------------------------ */
typedef void (*rcomb_fn)(const u64 *vals /* ntab x 16 */, void *ctx, u64 *out16);
static void ring_sumcheck_prove(lfp_tr *tr, u64 **tab, unsigned ntab, unsigned nv, unsigned deg, rcomb_fn comb, void *ctx, u64 *msgs, u64 *point) {
    absorb_const(tr, nv);
    absorb_const(tr, deg);
    size_t n = (size_t)1 << nv;
    u64 *vals = (u64 *)malloc((size_t)ntab * D * sizeof(u64)), *step = (u64 *)malloc((size_t)ntab * D * sizeof(u64));
    for (unsigned rnd = 0; rnd < nv; rnd++) {
        size_t half = n >> 1;
        u64 *m = msgs + (size_t)rnd * (deg + 1) * D, o[D];
        memset(m, 0, (size_t)(deg + 1) * D * sizeof(u64));
        for (size_t b = 0; b < half; b++) {
            for (unsigned t = 0; t < ntab; t++)
                for (int c = 0; c < D; c++) {
                    vals[t * D + c] = tab[t][(2 * b) * D + c];
                    step[t * D + c] = fsub(tab[t][(2 * b + 1) * D + c], tab[t][(2 * b) * D + c]);
                }
            for (unsigned x = 0; x <= deg; x++) {
                if (x) for (unsigned i = 0; i < ntab * D; i++) vals[i] = fadd(vals[i], step[i]);
                comb(vals, ctx, o);
                radd(m + x * D, o);
            }
        }
        lfp_tr_absorb(tr, m, deg + 1);
        u64 r = lfp_tr_challenge(tr);
        absorb_const(tr, r);
        point[rnd] = r;
        for (unsigned t = 0; t < ntab; t++)
            for (size_t b = 0; b < half; b++)
                for (int c = 0; c < D; c++) {
                    u64 lo = tab[t][(2 * b) * D + c], hi = tab[t][(2 * b + 1) * D + c];
                    tab[t][b * D + c] = fadd(lo, fmul(r, fsub(hi, lo)));
                }
        n = half;
    }
    free(vals); free(step);
}
static int ring_sumcheck_verify(lfp_tr *tr, unsigned nv, unsigned deg, const u64 *claimed16, const u64 *msgs, u64 *point, u64 *expected16) {
    absorb_const(tr, nv);
    absorb_const(tr, deg);
    u64 cur[D];
    memcpy(cur, claimed16, sizeof(cur));
    for (unsigned rnd = 0; rnd < nv; rnd++) {
        const u64 *m = msgs + (size_t)rnd * (deg + 1) * D;
        lfp_tr_absorb(tr, m, deg + 1);
        u64 r = lfp_tr_challenge(tr);
        absorb_const(tr, r);
        point[rnd] = r;
        for (int c = 0; c < D; c++) {
            u64 y[8];
            for (unsigned x = 0; x <= deg; x++) y[x] = m[x * D + c] % P;
            if (fadd(y[0], y[1]) != cur[c]) return -1;
            cur[c] = interpolate(y, deg + 1, r);
        }
    }
    memcpy(expected16, cur, sizeof(cur));
    return 0;
}
static void rmul_acc(u64 *acc, const u64 *a, const u64 *b) { u64 t[D]; lfp_ring_mul(a, b, t); radd(acc, t); }

/* ---- Cm::prove / CmProof::verify (cm.rs:56-347 / 349-580) ---------------------------------------------------------------------------------- */
/* calculate_t_z (cm.rs:593-603): tensor(c) (x) s' (x) (1, d', .., d'^(l-1)) (x) (1, X, .., X^(d-1)), zero-padded to n ring elements */
static int calc_t_z(const u64 *c, unsigned logk, const u64 *sp /* kd ring */, unsigned kd, unsigned ell, size_t n, u64 *out /* n*16, zeroed */) {
    size_t tl = (size_t)1 << logk;
    if (tl * kd * ell * D > n) return -1;     /* "t0 too large!" */
    u64 *tc = (u64 *)malloc(tl * sizeof(u64));
    lfp_tensor(c, logk, tc);
    memset(out, 0, n * D * sizeof(u64));
    for (size_t a = 0; a < tl; a++)
        for (unsigned b = 0; b < kd; b++) {
            u64 pw = 1;
            for (unsigned i = 0; i < ell; i++) {
                for (int m = 0; m < D; m++) {
                    /* tensor_c[a] * s'[b] * d'^i * X^m: the short element scaled, then rotated by m (X^16 = -1) */
                    u64 *o = out + (((a * kd + b) * ell + i) * D + m) * D;
                    u64 sc = fmul(tc[a], pw);
                    for (int t = 0; t < D; t++) {
                        u64 v = fmul(sp[(size_t)b * D + t], sc);
                        if (t + m < D) o[t + m] = v; else o[t + m - D] = fsub(0, v);
                    }
                }
                pw = fmul(pw, D / 2);
            }
        }
    free(tc);
    return 0;
}
typedef struct { unsigned L, nM, ntab; const u64 *rcps; } cm_ctx;
/* comb_fn of cm.rs:287-311: vals = [eq | per instance: tau, m_tau, f, h, then per matrix M tau, M m_tau, M f, M h | t0, t1] */
static void cm_comb(const u64 *vals, void *vctx, u64 *out) {
    const cm_ctx *c = (const cm_ctx *)vctx;
    memset(out, 0, D * sizeof(u64));
    unsigned per = 4 + 4 * c->nM;
    for (unsigned l = 0; l < c->L; l++) {
        unsigned l_idx = 1 + l * per;
        u64 inner[D] = {0}, t[D];
        for (unsigned j = 0; j < per; j++) rscale_add(inner, vals + (size_t)(l_idx + j) * D, c->rcps[l_idx - 1 + j]);
        rmul_acc(out, vals, inner);                                   /* eq * (...) */
        lfp_ring_mul(vals + (size_t)l_idx * D, vals + (size_t)(c->ntab - 2) * D, t);
        rscale_add(out, t, c->rcps[c->ntab - 3]);                     /* (tau * t0) * rc^z_idx */
        lfp_ring_mul(vals + (size_t)l_idx * D, vals + (size_t)(c->ntab - 1) * D, t);
        rscale_add(out, t, c->rcps[c->ntab - 2]);
    }
}
/* Cm::prove.  Inputs per instance l: Mf (k x n x 16 monomials, dense), tau (n), mtau (n ring), f (n ring), comMf (k x kappa x 16 ring), fcoms
 * (3 x kappa ring: cm_f | C_Mf | cm_mtau).  Outputs: the range check's (r .. c as lfp_range_check), comh (L x kappa), sumcheck proofs pa / pb
 * (nvars x 3 ring each), evals ea / eb (L x (1 + nM) x 4 ring), g (L x n ring), cm_g (L x kappa), ro (2 x nvars words: ro_a | ro_b), vo (L x (1 + nM)
 * x 2 ring).  ell = DecompParameters::l. */
int lfp_cm_prove(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, unsigned ell, unsigned kappa, const u64 *const *Mf, const u64 *const *tau,
                 const u64 *const *mtau, const u64 *const *f, const u64 *const *comMf, const u64 *const *fcoms, unsigned nM, const uint32_t *const *rowptr,
                 const uint32_t *const *col, const u64 *const *val, u64 *r_out, u64 *msgs, u64 *e_out, u64 *b_out, u64 *v_out, u64 *a_out, u64 *bb_out,
                 u64 *c_out, u64 *comh, u64 *pa, u64 *pb, u64 *ea, u64 *eb, u64 *g, u64 *cm_g, u64 *ro, u64 *vo) {
    size_t n = (size_t)1 << nvars;
    int rc = lfp_range_check(tr, nvars, L, k, Mf, tau, mtau, f, nM, rowptr, col, val, r_out, msgs, e_out, b_out, v_out, a_out, bb_out, c_out);
    if (rc) return rc;
    u64 s[3][D], *sp = (u64 *)malloc((size_t)k * D * D * sizeof(u64));
    for (int i = 0; i < 3; i++) lfp_short_challenge(tr, s[i]);
    for (unsigned i = 0; i < k * D; i++) lfp_short_challenge(tr, sp + (size_t)i * D);
    /* h = sum_ki M_f[ki] s'_ki (cm.rs:82-103); comh = sum_ki comM_f[ki] s'_ki (:105-126) */
    u64 **h = (u64 **)malloc(L * sizeof(u64 *));
    for (unsigned l = 0; l < L; l++) {
        h[l] = (u64 *)calloc(n * D, sizeof(u64));
        for (unsigned ki = 0; ki < k; ki++)
            for (size_t row = 0; row < n; row++)
                for (int j = 0; j < D; j++) rmul_acc(h[l] + row * D, Mf[l] + (((size_t)ki * n + row) * D + j) * D, sp + ((size_t)ki * D + j) * D);
        for (unsigned i = 0; i < kappa; i++) {
            u64 *o = comh + ((size_t)l * kappa + i) * D;
            memset(o, 0, D * sizeof(u64));
            for (unsigned ki = 0; ki < k; ki++)
                for (int j = 0; j < D; j++) rmul_acc(o, comMf[l] + ((((size_t)ki * kappa + i) * D) + j) * D, sp + ((size_t)ki * D + j) * D);
        }
    }
    lfp_tr_absorb(tr, comh, (size_t)L * kappa);
    unsigned logk = 0;
    while (((unsigned)1 << logk) < kappa) logk++;
    u64 cz[2][32];
    for (int z = 0; z < 2; z++) for (unsigned j = 0; j < logk; j++) cz[z][j] = lfp_tr_challenge(tr);
    u64 *t0 = (u64 *)malloc(n * D * sizeof(u64)), *t1 = (u64 *)malloc(n * D * sizeof(u64));
    if (calc_t_z(cz[0], logk, sp, k * D, ell, n, t0) || calc_t_z(cz[1], logk, sp, k * D, ell, n, t1)) return -7;
    /* the two sumcheckers (cm.rs:201-347) */
    unsigned per = 4 + 4 * nM, ntab = 1 + L * per + 2;
    for (int pass = 0; pass < 2; pass++) {
        u64 rcv = lfp_tr_challenge(tr), *rcps = (u64 *)malloc(ntab * sizeof(u64));
        rcps[0] = 1;
        for (unsigned i = 1; i < ntab; i++) rcps[i] = fmul(rcps[i - 1], rcv);
        u64 **tab = (u64 **)malloc(ntab * sizeof(u64 *));
        tab[0] = (u64 *)calloc(n * D, sizeof(u64));
        { u64 *eq = build_eq(r_out, nvars); for (size_t i = 0; i < n; i++) tab[0][i * D] = eq[i]; free(eq); }
        for (unsigned l = 0; l < L; l++) {
            u64 **tl = tab + 1 + l * per;
            for (unsigned j = 0; j < per; j++) tl[j] = (u64 *)calloc(n * D, sizeof(u64));
            for (size_t i = 0; i < n; i++) tl[0][i * D] = tau[l][i] % P;
            memcpy(tl[1], mtau[l], n * D * sizeof(u64));
            memcpy(tl[2], f[l], n * D * sizeof(u64));
            memcpy(tl[3], h[l], n * D * sizeof(u64));
            for (unsigned q = 0; q < nM; q++) {
                csr m = {n, rowptr[q], col[q], val[q]};
                for (int j = 0; j < 4; j++) spmv_ring(&m, tl[j], tl[4 + 4 * q + j]);
            }
        }
        tab[ntab - 2] = (u64 *)malloc(n * D * sizeof(u64)); memcpy(tab[ntab - 2], t0, n * D * sizeof(u64));
        tab[ntab - 1] = (u64 *)malloc(n * D * sizeof(u64)); memcpy(tab[ntab - 1], t1, n * D * sizeof(u64));
        cm_ctx ctx = {L, nM, ntab, rcps};
        u64 *proof = pass ? pb : pa, *evs = pass ? eb : ea, *rop = ro + (size_t)pass * nvars;
        ring_sumcheck_prove(tr, tab, ntab, nvars, 2, cm_comb, &ctx, proof, rop);
        /* evals: every instance table at ro = the fully fixed tables (one entry left) */
        for (unsigned l = 0; l < L; l++)
            for (unsigned j = 0; j < per; j++) memcpy(evs + ((size_t)l * per + j) * D, tab[1 + l * per + j], D * sizeof(u64));
        lfp_tr_absorb(tr, evs, (size_t)L * per);
        for (unsigned t = 0; t < ntab; t++) free(tab[t]);
        free(tab); free(rcps);
    }
    /* g = s0 tau + s1 m_tau + s2 f + h (cm.rs:164-181); x = CmProof::x (cm.rs:545-580) */
    for (unsigned l = 0; l < L; l++) {
        for (size_t i = 0; i < n; i++) {
            u64 *o = g + ((size_t)l * n + i) * D, tr16[D] = {0};
            memcpy(o, h[l] + i * D, D * sizeof(u64));
            tr16[0] = tau[l][i] % P;
            rmul_acc(o, s[0], tr16);
            rmul_acc(o, s[1], mtau[l] + i * D);
            rmul_acc(o, s[2], f[l] + i * D);
        }
        for (unsigned i = 0; i < kappa; i++) {
            u64 *o = cm_g + ((size_t)l * kappa + i) * D;
            memcpy(o, comh + ((size_t)l * kappa + i) * D, D * sizeof(u64));
            rmul_acc(o, s[0], fcoms[l] + ((size_t)1 * kappa + i) * D);     /* C_Mf */
            rmul_acc(o, s[1], fcoms[l] + ((size_t)2 * kappa + i) * D);     /* cm_mtau */
            rmul_acc(o, s[2], fcoms[l] + ((size_t)0 * kappa + i) * D);     /* cm_f */
        }
        for (unsigned q = 0; q < 1 + nM; q++)
            for (int pass = 0; pass < 2; pass++) {
                const u64 *e4 = (pass ? eb : ea) + ((size_t)l * per + 4 * q) * D;
                u64 *o = vo + (((size_t)l * (1 + nM) + q) * 2 + pass) * D;
                memcpy(o, e4 + 3 * D, D * sizeof(u64));
                rmul_acc(o, s[0], e4); rmul_acc(o, s[1], e4 + D); rmul_acc(o, s[2], e4 + 2 * D);
            }
        free(h[l]);
    }
    free(h); free(sp); free(t0); free(t1);
    return 0;
}
/* CmProof::verify (cm.rs:349-543).  0 = accepted (cm_g / ro / vo are then the ComX of cm.rs:545-580, recomputed from the proof);
 * -1..-5 range check, -6 a sumcheck, -8 final evaluation of a sumchecker */
int lfp_cm_verify(lfp_tr *tr, unsigned nvars, unsigned L, unsigned k, unsigned ell, unsigned kappa, unsigned nM, const u64 *const *fcoms, const u64 *msgs,
                  const u64 *e, const u64 *b, const u64 *v, const u64 *a, const u64 *bb, const u64 *c, const u64 *comh, const u64 *pa, const u64 *pb,
                  const u64 *ea, const u64 *eb, u64 *cm_g, u64 *ro, u64 *vo) {
    size_t n = (size_t)1 << nvars;
    u64 *r = (u64 *)malloc(nvars * sizeof(u64));
    int rc = lfp_range_check_verify(tr, nvars, L, k, nM, msgs, e, b, v, a, bb, c, r);
    if (rc) { free(r); return rc; }
    u64 s[3][D], *sp = (u64 *)malloc((size_t)k * D * D * sizeof(u64));
    for (int i = 0; i < 3; i++) lfp_short_challenge(tr, s[i]);
    for (unsigned i = 0; i < k * D; i++) lfp_short_challenge(tr, sp + (size_t)i * D);
    lfp_tr_absorb(tr, comh, (size_t)L * kappa);
    unsigned logk = 0;
    while (((unsigned)1 << logk) < kappa) logk++;
    u64 cz[2][32];
    for (int z = 0; z < 2; z++) for (unsigned j = 0; j < logk; j++) cz[z][j] = lfp_tr_challenge(tr);
    /* u[l][q] = sum over the instance's k 16 columns of e[q][..] * s'  (cm.rs:383-401) */
    unsigned per = 4 + 4 * nM, z_idx = L * per;
    u64 *u = (u64 *)calloc((size_t)L * (1 + nM) * D, sizeof(u64));
    for (unsigned l = 0; l < L; l++)
        for (unsigned q = 0; q < 1 + nM; q++)
            for (unsigned j = 0; j < k * D; j++) rmul_acc(u + ((size_t)l * (1 + nM) + q) * D, e + (((size_t)q * L * k + (size_t)l * k) * D + j) * D, sp + (size_t)j * D);
    size_t tl = (size_t)1 << logk;
    u64 *tc0 = (u64 *)malloc(tl * sizeof(u64)), *tc1 = (u64 *)malloc(tl * sizeof(u64));
    lfp_tensor(cz[0], logk, tc0); lfp_tensor(cz[1], logk, tc1);
    u64 *tcch = (u64 *)calloc((size_t)2 * L * D, sizeof(u64));
    for (unsigned l = 0; l < L; l++)
        for (unsigned i = 0; i < kappa && i < tl; i++) {
            rscale_add(tcch + ((size_t)0 * L + l) * D, comh + ((size_t)l * kappa + i) * D, tc0[i]);
            rscale_add(tcch + ((size_t)1 * L + l) * D, comh + ((size_t)l * kappa + i) * D, tc1[i]);
        }
    u64 *t0 = (u64 *)malloc(n * D * sizeof(u64)), *t1 = (u64 *)malloc(n * D * sizeof(u64));
    if (calc_t_z(cz[0], logk, sp, k * D, ell, n, t0) || calc_t_z(cz[1], logk, sp, k * D, ell, n, t1)) rc = -7;
    for (int pass = 0; pass < 2 && !rc; pass++) {
        u64 rcv = lfp_tr_challenge(tr), *rcps = (u64 *)malloc((z_idx + 2) * sizeof(u64));
        rcps[0] = 1;
        for (unsigned i = 1; i < z_idx + 2; i++) rcps[i] = fmul(rcps[i - 1], rcv);
        u64 claim[D] = {0};
        for (unsigned l = 0; l < L; l++) {
            unsigned l_idx = l * per;
            for (unsigned q = 0; q < 1 + nM; q++) {
                unsigned idx = l_idx + 4 * q;
                claim[0] = fadd(claim[0], fmul(a[(size_t)l * (1 + nM) + q] % P, rcps[idx]));
                rscale_add(claim, bb + ((size_t)l * (1 + nM) + q) * D, rcps[idx + 1]);
                rscale_add(claim, c + ((size_t)l * (1 + nM) + q) * D, rcps[idx + 2]);
                rscale_add(claim, u + ((size_t)l * (1 + nM) + q) * D, rcps[idx + 3]);
            }
            rscale_add(claim, tcch + ((size_t)0 * L + l) * D, rcps[z_idx]);
            rscale_add(claim, tcch + ((size_t)1 * L + l) * D, rcps[z_idx + 1]);
        }
        const u64 *proof = pass ? pb : pa, *evs = pass ? eb : ea;
        u64 *rop = ro + (size_t)pass * nvars, expected[D];
        if (ring_sumcheck_verify(tr, nvars, 2, claim, proof, rop, expected)) { rc = -6; free(rcps); break; }
        u64 *eqo = build_eq(rop, nvars), t0r[D], t1r[D];
        mle_eval_ring(t0, n, eqo, t0r);
        mle_eval_ring(t1, n, eqo, t1r);
        free(eqo);
        lfp_tr_absorb(tr, evs, (size_t)L * per);
        u64 eq = eq_eval(r, rop, nvars), ev16[D] = {0};
        for (unsigned l = 0; l < L; l++) {
            const u64 *el = evs + (size_t)l * per * D;
            u64 inner[D] = {0};
            for (unsigned j = 0; j < per; j++) rscale_add(inner, el + (size_t)j * D, rcps[l * per + j]);
            rscale_add(ev16, inner, eq);
            u64 t[D];
            lfp_ring_mul(t0r, el, t); rscale_add(ev16, t, rcps[z_idx]);
            lfp_ring_mul(t1r, el, t); rscale_add(ev16, t, rcps[z_idx + 1]);
        }
        if (memcmp(ev16, expected, sizeof(ev16))) rc = -8;
        free(rcps);
    }
    if (!rc) {   /* CmProof::x */
        unsigned q, pass;
        for (unsigned l = 0; l < L; l++) {
            for (unsigned i = 0; i < kappa; i++) {
                u64 *o = cm_g + ((size_t)l * kappa + i) * D;
                memcpy(o, comh + ((size_t)l * kappa + i) * D, D * sizeof(u64));
                rmul_acc(o, s[0], fcoms[l] + ((size_t)1 * kappa + i) * D);
                rmul_acc(o, s[1], fcoms[l] + ((size_t)2 * kappa + i) * D);
                rmul_acc(o, s[2], fcoms[l] + ((size_t)0 * kappa + i) * D);
            }
            for (q = 0; q < 1 + nM; q++)
                for (pass = 0; pass < 2; pass++) {
                    const u64 *e4 = (pass ? eb : ea) + ((size_t)l * per + 4 * q) * D;
                    u64 *o = vo + (((size_t)l * (1 + nM) + q) * 2 + pass) * D;
                    memcpy(o, e4 + 3 * D, D * sizeof(u64));
                    rmul_acc(o, s[0], e4); rmul_acc(o, s[1], e4 + D); rmul_acc(o, s[2], e4 + 2 * D);
                }
        }
    }
    free(r); free(sp); free(u); free(tc0); free(tc1); free(tcch); free(t0); free(t1);
    return rc;
}

/* ---- ComR1CS::linearize / ComR1CSProof::verify (r1cs.rs:76-162) ----------------------------------------------------------------------------- */
/* comb_fn of r1cs.rs:95: vals = [eq | ga | gb | gc] -> eq (ga gb - gc), ring products */
static void r1cs_comb(const u64 *vals, void *ctx, u64 *out) {
    (void)ctx;
    u64 t[D];
    lfp_ring_mul(vals + D, vals + 2 * D, t);
    for (int i = 0; i < D; i++) t[i] = fsub(t[i], vals[3 * D + i]);
    lfp_ring_mul(vals, t, out);
}
/* f: n = 2^nvars ring elements; the three R1CS matrices n x n in CSR form.  Outputs: msgs nvars x 4 ring elements (degree 3), ro nvars words, evals = v | va | vb
 * | vc (4 ring elements: f, A f, B f, C f at ro) */
int lfp_r1cs_linearize(lfp_tr *tr, unsigned nvars, const u64 *f, const uint32_t *const *rowptr, const uint32_t *const *col, const u64 *const *val, u64 *msgs,
                       u64 *ro, u64 *evals) {
    size_t n = (size_t)1 << nvars;
    u64 *g[3], *tab[4], r[64];
    for (int q = 0; q < 3; q++) {
        csr m = {n, rowptr[q], col[q], val[q]};
        g[q] = (u64 *)malloc(n * D * sizeof(u64));
        spmv_ring(&m, f, g[q]);
    }
    for (unsigned j = 0; j < nvars; j++) r[j] = lfp_tr_challenge(tr);
    u64 *eq = build_eq(r, nvars);
    tab[0] = (u64 *)calloc(n * D, sizeof(u64));
    for (size_t i = 0; i < n; i++) tab[0][i * D] = eq[i];
    free(eq);
    for (int q = 0; q < 3; q++) { tab[1 + q] = (u64 *)malloc(n * D * sizeof(u64)); memcpy(tab[1 + q], g[q], n * D * sizeof(u64)); }
    ring_sumcheck_prove(tr, tab, 4, nvars, 3, r1cs_comb, NULL, msgs, ro);
    u64 *eqo = build_eq(ro, nvars);
    mle_eval_ring(f, n, eqo, evals);
    for (int q = 0; q < 3; q++) mle_eval_ring(g[q], n, eqo, evals + (size_t)(1 + q) * D);
    free(eqo);
    lfp_tr_absorb(tr, evals, 4);
    for (int q = 0; q < 3; q++) free(g[q]);
    for (int t = 0; t < 4; t++) free(tab[t]);
    return 0;
}
/* 0 accepted; -1 a sumcheck round; -2 e (va vb - vc) != s (r1cs.rs:159, an assert_eq in the reference) */
int lfp_r1cs_verify(lfp_tr *tr, unsigned nvars, const u64 *msgs, const u64 *evals, u64 *ro) {
    u64 r[64], zero[D] = {0}, s[D], t[D], want[D];
    for (unsigned j = 0; j < nvars; j++) r[j] = lfp_tr_challenge(tr);
    if (ring_sumcheck_verify(tr, nvars, 3, zero, msgs, ro, s)) return -1;
    lfp_tr_absorb(tr, evals, 4);
    u64 e = eq_eval(r, ro, nvars);
    lfp_ring_mul(evals + D, evals + 2 * D, t);
    for (int i = 0; i < D; i++) want[i] = fmul(e, fsub(t[i], evals[3 * D + i] % P));
    return memcmp(want, s, sizeof(want)) ? -2 : 0;
}
/* DecompProof::verify (decomp.rs:101-123): C0 + B C1 = cm_f and v0 + B v1 = v (count pairs); 0 / -1 commitment / -2 evaluations */
int lfp_decomp_verify(const u64 *C0, const u64 *C1, unsigned kappa, const u64 *v0, const u64 *v1, unsigned count, const u64 *cm_f, const u64 *v, u64 B) {
    for (size_t i = 0; i < (size_t)kappa * D; i++)
        if (fadd(C0[i] % P, fmul(B % P, C1[i] % P)) != cm_f[i] % P) return -1;
    for (size_t i = 0; i < (size_t)count * 2 * D; i++)
        if (fadd(v0[i] % P, fmul(B % P, v1[i] % P)) != v[i] % P) return -2;
    return 0;
}

/* lfo_field.h -- ORACLE (test infrastructure only): F_p and F_{p^tau} arithmetic for the ring selected
 * in lfo.h (Goldilocks tau = 3, BabyBear tau = 9).  Restates ark-ff 0.4.2 Fp64 / extension-field
 * semantics on canonical residues (the reference stores Montgomery form internally; every KAT and
 * Display shows canonical values).  F_{p^tau} = F_p[Y]/(Y^tau - NONRES). */
#ifndef LFO_FIELD_H
#define LFO_FIELD_H
#include "lfo.h"
#include <string.h>

#define TAU LFO_TAU
#define RE LFO_D /* words per ring element */
#define NSLOT LFO_SLOTS

typedef unsigned __int128 u128;
typedef struct { u64 c[TAU]; } fqe;

extern u64 lfo_NONRES; /* F_{p^tau} = F_p[Y]/(Y^tau - NONRES) */
/* general mode (lfo_set_ring_general): structure constants of F_{p^tau} in an arbitrary F_p-basis e_0 = 1, e_1, ..:
 * e_i * e_j = sum_k TENSOR[(i*tau + j)*tau + k] e_k.  NULL = the binomial basis above. */
extern const u64 *lfo_TENSOR;

static inline u64 fq_add(u64 a, u64 b) {
    u64 r = a + b;
    if (r < a || r >= LFO_P) r -= LFO_P;
    return r;
}
static inline u64 fq_sub(u64 a, u64 b) { return a >= b ? a - b : a + (LFO_P - b); }
static inline u64 fq_neg(u64 a) { return a ? LFO_P - a : 0; }
#ifdef LFO_RING_BABYBEAR
static inline u64 fq_mul(u64 a, u64 b) { return (a * b) % LFO_P; } /* a, b < 2^31 */
#else
/* x mod p using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p); cross-checked against `%` in tests */
static inline u64 fq_reduce128(u128 x) {
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= 0xFFFFFFFFULL; /* + p (mod 2^64) */
    u64 t1 = hl * 0xFFFFFFFFULL;
    u64 r = t0 + t1;
    if (r < t1) r += 0xFFFFFFFFULL; /* - p (mod 2^64) */
    if (r >= LFO_P) r -= LFO_P;
    return r;
}
static inline u64 fq_mul(u64 a, u64 b) { return fq_reduce128((u128)a * b); }
#endif
static inline u64 fq_mul_slow(u64 a, u64 b) { return (u64)(((u128)a * b) % LFO_P); }
static inline u64 fq_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = fq_mul(r, a);
        a = fq_mul(a, a);
        e >>= 1;
    }
    return r;
}
static inline u64 fq_inv(u64 a) { return fq_pow(a, LFO_P - 2); }
static inline u64 fq_from_i64(int64_t v) { return v >= 0 ? (u64)v % LFO_P : (LFO_P - ((u64)(-v) % LFO_P)) % LFO_P; }

static inline fqe fqe_zero(void) { fqe r; memset(&r, 0, sizeof(r)); return r; }
static inline fqe fqe_from_fq(u64 a) { fqe r = fqe_zero(); r.c[0] = a; return r; }
static inline fqe fqe_one(void) { return fqe_from_fq(1); }
static inline fqe fqe_load(const u64 *w) { fqe r; memcpy(r.c, w, sizeof(r.c)); return r; }
static inline void fqe_store(u64 *w, fqe a) { memcpy(w, a.c, sizeof(a.c)); }
static inline int fqe_is_zero(fqe a) { for (int i = 0; i < TAU; i++) if (a.c[i]) return 0; return 1; }
static inline fqe fqe_add(fqe a, fqe b) { fqe r; for (int i = 0; i < TAU; i++) r.c[i] = fq_add(a.c[i], b.c[i]); return r; }
static inline fqe fqe_sub(fqe a, fqe b) { fqe r; for (int i = 0; i < TAU; i++) r.c[i] = fq_sub(a.c[i], b.c[i]); return r; }
/* schoolbook product modulo Y^tau = NONRES */
static inline fqe fqe_mul(fqe a, fqe b) {
    if (lfo_TENSOR) {
        fqe r = fqe_zero();
        for (int i = 0; i < TAU; i++) {
            if (!a.c[i]) continue;
            for (int j = 0; j < TAU; j++) {
                if (!b.c[j]) continue;
                u64 pr = fq_mul(a.c[i], b.c[j]);
                const u64 *t = lfo_TENSOR + (size_t)(i * TAU + j) * TAU;
                for (int k = 0; k < TAU; k++)
                    if (t[k]) r.c[k] = fq_add(r.c[k], fq_mul(pr, t[k]));
            }
        }
        return r;
    }
    u64 lo[TAU], hi[TAU];
    memset(lo, 0, sizeof(lo));
    memset(hi, 0, sizeof(hi));
    for (int i = 0; i < TAU; i++)
        for (int j = 0; j < TAU; j++) {
            u64 pr = fq_mul(a.c[i], b.c[j]);
            if (i + j < TAU) lo[i + j] = fq_add(lo[i + j], pr);
            else hi[i + j - TAU] = fq_add(hi[i + j - TAU], pr);
        }
    fqe r;
    for (int k = 0; k < TAU; k++) r.c[k] = fq_add(lo[k], fq_mul(lfo_NONRES, hi[k]));
    return r;
}
static inline fqe fqe_mul_fq(fqe a, u64 s) { fqe r; for (int i = 0; i < TAU; i++) r.c[i] = fq_mul(a.c[i], s); return r; }

/* ---- ring elements in NTT form: 8 slots of fqe, stored slot-major in RE words ---------- */
static inline fqe rq_slot(const u64 *e, int k) { return fqe_load(e + TAU * k); }
static inline void rq_set_slot(u64 *e, int k, fqe v) { fqe_store(e + TAU * k, v); }
static inline void rq_zero(u64 *e) { memset(e, 0, RE * sizeof(u64)); }
static inline void rq_copy(u64 *d, const u64 *s) { memcpy(d, s, RE * sizeof(u64)); }
static inline int rq_is_zero(const u64 *e) { for (int i = 0; i < RE; i++) if (e[i]) return 0; return 1; }
static inline int rq_eq(const u64 *a, const u64 *b) { return memcmp(a, b, RE * sizeof(u64)) == 0; }
static inline void rq_add(u64 *d, const u64 *a, const u64 *b) { for (int i = 0; i < RE; i++) d[i] = fq_add(a[i], b[i]); }
static inline void rq_sub(u64 *d, const u64 *a, const u64 *b) { for (int i = 0; i < RE; i++) d[i] = fq_sub(a[i], b[i]); }
static inline void rq_mul(u64 *d, const u64 *a, const u64 *b) { /* NTT form: slot-wise */
    u64 t[RE];
    for (int k = 0; k < NSLOT; k++) rq_set_slot(t, k, fqe_mul(rq_slot(a, k), rq_slot(b, k)));
    rq_copy(d, t);
}
static inline void rq_mul_fqe(u64 *d, const u64 *a, fqe s) { /* times diagonal embedding of s */
    for (int k = 0; k < NSLOT; k++) rq_set_slot(d, k, fqe_mul(rq_slot(a, k), s));
}
static inline void rq_from_fqe(u64 *d, fqe s) { for (int k = 0; k < NSLOT; k++) rq_set_slot(d, k, s); } /* R::from(BaseRing) */
static inline void rq_from_u64(u64 *d, u64 v) { rq_from_fqe(d, fqe_from_fq(v % LFO_P)); }              /* R::from(u128) */
/* is the element a diagonal embedding?  if so return the fqe */
static inline int rq_is_diag(const u64 *e, fqe *out) {
    for (int k = 1; k < NSLOT; k++)
        if (memcmp(e + TAU * k, e, TAU * sizeof(u64))) return 0;
    if (out) *out = rq_slot(e, 0);
    return 1;
}
#endif

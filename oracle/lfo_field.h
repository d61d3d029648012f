/* lfo_field.h -- ORACLE (test infrastructure only): Goldilocks F_p and F_{p^3} arithmetic.
 * Restates ark-ff 0.4.2 Fp64 / Fp3 semantics on canonical residues (the reference stores
 * Montgomery form internally; every KAT and Display shows canonical values). */
#ifndef LFO_FIELD_H
#define LFO_FIELD_H
#include "lfo.h"
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { u64 c[3]; } fq3;

extern u64 lfo_NONRES; /* F_{p^3} = F_p[Y]/(Y^3 - NONRES) */

static inline u64 fq_add(u64 a, u64 b) {
    u64 r = a + b;
    if (r < a || r >= LFO_P) r -= LFO_P;
    return r;
}
static inline u64 fq_sub(u64 a, u64 b) { return a >= b ? a - b : a + (LFO_P - b); }
static inline u64 fq_neg(u64 a) { return a ? LFO_P - a : 0; }
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
static inline u64 fq_from_i64(int64_t v) { return v >= 0 ? (u64)v % LFO_P : LFO_P - ((u64)(-v) % LFO_P); }

static inline fq3 fq3_zero(void) { fq3 r = {{0, 0, 0}}; return r; }
static inline fq3 fq3_one(void) { fq3 r = {{1, 0, 0}}; return r; }
static inline fq3 fq3_from_fq(u64 a) { fq3 r = {{a, 0, 0}}; return r; }
static inline int fq3_is_zero(fq3 a) { return !(a.c[0] | a.c[1] | a.c[2]); }
static inline fq3 fq3_add(fq3 a, fq3 b) {
    fq3 r = {{fq_add(a.c[0], b.c[0]), fq_add(a.c[1], b.c[1]), fq_add(a.c[2], b.c[2])}};
    return r;
}
static inline fq3 fq3_sub(fq3 a, fq3 b) {
    fq3 r = {{fq_sub(a.c[0], b.c[0]), fq_sub(a.c[1], b.c[1]), fq_sub(a.c[2], b.c[2])}};
    return r;
}
/* schoolbook product modulo Y^3 = NONRES */
static inline fq3 fq3_mul(fq3 a, fq3 b) {
    u64 nr = lfo_NONRES;
    fq3 r;
    r.c[0] = fq_add(fq_mul(a.c[0], b.c[0]), fq_mul(nr, fq_add(fq_mul(a.c[1], b.c[2]), fq_mul(a.c[2], b.c[1]))));
    r.c[1] = fq_add(fq_add(fq_mul(a.c[0], b.c[1]), fq_mul(a.c[1], b.c[0])), fq_mul(nr, fq_mul(a.c[2], b.c[2])));
    r.c[2] = fq_add(fq_add(fq_mul(a.c[0], b.c[2]), fq_mul(a.c[1], b.c[1])), fq_mul(a.c[2], b.c[0]));
    return r;
}
static inline fq3 fq3_mul_fq(fq3 a, u64 s) {
    fq3 r = {{fq_mul(a.c[0], s), fq_mul(a.c[1], s), fq_mul(a.c[2], s)}};
    return r;
}

/* ---- ring elements in NTT form: 8 slots of fq3, stored slot-major in 24 words ---------- */
static inline fq3 rq_slot(const u64 *e, int k) { fq3 r = {{e[3 * k], e[3 * k + 1], e[3 * k + 2]}}; return r; }
static inline void rq_set_slot(u64 *e, int k, fq3 v) { e[3 * k] = v.c[0]; e[3 * k + 1] = v.c[1]; e[3 * k + 2] = v.c[2]; }
static inline void rq_zero(u64 *e) { memset(e, 0, 24 * sizeof(u64)); }
static inline void rq_copy(u64 *d, const u64 *s) { memcpy(d, s, 24 * sizeof(u64)); }
static inline int rq_is_zero(const u64 *e) { for (int i = 0; i < 24; i++) if (e[i]) return 0; return 1; }
static inline int rq_eq(const u64 *a, const u64 *b) { return memcmp(a, b, 24 * sizeof(u64)) == 0; }
static inline void rq_add(u64 *d, const u64 *a, const u64 *b) { for (int i = 0; i < 24; i++) d[i] = fq_add(a[i], b[i]); }
static inline void rq_sub(u64 *d, const u64 *a, const u64 *b) { for (int i = 0; i < 24; i++) d[i] = fq_sub(a[i], b[i]); }
static inline void rq_mul(u64 *d, const u64 *a, const u64 *b) { /* NTT form: slot-wise */
    u64 t[24];
    for (int k = 0; k < 8; k++) rq_set_slot(t, k, fq3_mul(rq_slot(a, k), rq_slot(b, k)));
    rq_copy(d, t);
}
static inline void rq_mul_fq3(u64 *d, const u64 *a, fq3 s) { /* times diagonal embedding of s */
    for (int k = 0; k < 8; k++) rq_set_slot(d, k, fq3_mul(rq_slot(a, k), s));
}
static inline void rq_from_fq3(u64 *d, fq3 s) { for (int k = 0; k < 8; k++) rq_set_slot(d, k, s); } /* R::from(BaseRing) */
static inline void rq_from_u64(u64 *d, u64 v) { rq_from_fq3(d, fq3_from_fq(v % LFO_P)); }           /* R::from(u128) */
/* is the element a diagonal embedding?  if so return the fq3 */
static inline int rq_is_diag(const u64 *e, fq3 *out) {
    for (int k = 1; k < 8; k++)
        if (e[3 * k] != e[0] || e[3 * k + 1] != e[1] || e[3 * k + 2] != e[2]) return 0;
    if (out) *out = rq_slot(e, 0);
    return 1;
}
#endif

// lf_field.cuh -- Goldilocks F_p (p = 2^64 - 2^32 + 1) and F_{p^3} = F_p[Y]/(Y^3 - NU) arithmetic for
// gfx950, shared by device kernels and the host driver.
//
// Replaces ark-ff 0.4.2 `Fp64<MontBackend>` / `Fp3` as used through stark-rings (reference call sites:
// crates/cyclotomic-rings/src/rings/goldilocks.rs:1-25).  Values are canonical residues in [0,p) at every
// kernel boundary; inside kernels "loose" values in [0,2^64) are allowed where noted.
//
// gfx950 has no 64x64->128 multiply: a product is four v_mad_u64_u32 (quarter rate, measured 15.7 T lane-op/s,
// profiles/r01_microbench.txt) plus a shift/add reduction that uses 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

#define LF_P 0xFFFFFFFF00000001ULL
#define LF_EPS 0xFFFFFFFFULL /* 2^32 - 1 = 2^64 mod p */

#define LF_HD __host__ __device__ __forceinline__

namespace lf {

// ---- F_p ------------------------------------------------------------------------------------------------
LF_HD u64 fq_canon(u64 a) { return a >= LF_P ? a - LF_P : a; }  // loose -> canonical
LF_HD u64 fq_add(u64 a, u64 b) {                                 // canonical in, canonical out
    u64 r = a + b;
    if (r < a || r >= LF_P) r -= LF_P;
    return r;
}
LF_HD u64 fq_sub(u64 a, u64 b) { return a >= b ? a - b : a + (LF_P - b); }
LF_HD u64 fq_neg(u64 a) { return a ? LF_P - a : 0; }

// (hi:lo) mod p, result loose (in [0,2^64)); one conditional subtract canonicalises.
LF_HD u64 fq_reduce128_loose(u64 lo, u64 hi) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= LF_EPS;            // borrow: + p (mod 2^64)
    u32 nz = hl != 0;
    u64 t1 = ((u64)(hl - nz) << 32) | (u32)(0u - hl);  // hl * (2^32 - 1) without a multiply
    u64 r = t0 + t1;
    if (r < t1) r += LF_EPS;              // carry: - p (mod 2^64)
    return r;
}
LF_HD void mul64wide(u64 a, u64 b, u64 &lo, u64 &hi) {
#if !defined(__HIP_DEVICE_COMPILE__)
    unsigned __int128 pr = (unsigned __int128)a * b;  // host: one mulq
    lo = (u64)pr;
    hi = (u64)(pr >> 64);
    return;
#endif
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    hi = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    lo = (p10 << 32) | (u32)p00;
}
LF_HD u64 fq_mul_loose(u64 a, u64 b) {  // any u64 inputs, loose output
    u64 lo, hi;
    mul64wide(a, b, lo, hi);
    return fq_reduce128_loose(lo, hi);
}
LF_HD u64 fq_mul(u64 a, u64 b) { return fq_canon(fq_mul_loose(a, b)); }
// a * 2^40 mod p (NU = 2^40 fast path): (a << 40) is a 104-bit value
LF_HD u64 fq_mul_2p40(u64 a) { return fq_canon(fq_reduce128_loose(a << 40, a >> 24)); }
// small signed integer -> canonical
LF_HD u64 fq_from_i64(int64_t v) { return v >= 0 ? (u64)v : LF_P - (u64)(-v); }  // |v| < p assumed

LF_HD u64 fq_pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = fq_mul(r, a);
        a = fq_mul(a, a);
        e >>= 1;
    }
    return r;
}
LF_HD u64 fq_inv(u64 a) { return fq_pow(a, LF_P - 2); }

// ---- F_{p^3} ----------------------------------------------------------------------------------------------
struct Fq3 {
    u64 c[3];
};
LF_HD Fq3 fq3_make(u64 a, u64 b, u64 c) { Fq3 r; r.c[0] = a; r.c[1] = b; r.c[2] = c; return r; }
LF_HD Fq3 fq3_zero() { return fq3_make(0, 0, 0); }
LF_HD Fq3 fq3_one() { return fq3_make(1, 0, 0); }
LF_HD Fq3 fq3_add(Fq3 a, Fq3 b) { return fq3_make(fq_add(a.c[0], b.c[0]), fq_add(a.c[1], b.c[1]), fq_add(a.c[2], b.c[2])); }
LF_HD Fq3 fq3_sub(Fq3 a, Fq3 b) { return fq3_make(fq_sub(a.c[0], b.c[0]), fq_sub(a.c[1], b.c[1]), fq_sub(a.c[2], b.c[2])); }
LF_HD Fq3 fq3_neg(Fq3 a) { return fq3_make(fq_neg(a.c[0]), fq_neg(a.c[1]), fq_neg(a.c[2])); }
LF_HD bool fq3_eq(Fq3 a, Fq3 b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2]; }

// multiply by the non-residue.  NU2P40 = true: NU = 2^40 (shift); else generic runtime constant.
template <bool NU2P40>
LF_HD u64 fq_mul_nu(u64 a, u64 nu) { return NU2P40 ? fq_mul_2p40(a) : fq_mul(a, nu); }

// 128-bit + carry accumulator for sums of up to 3 products < 2^128
struct Acc {
    u64 lo, hi;
    u32 ov;
};
LF_HD void acc_set(Acc &s, u64 a, u64 b) { mul64wide(a, b, s.lo, s.hi); s.ov = 0; }
LF_HD void acc_mad(Acc &s, u64 a, u64 b) {
    u64 lo, hi;
    mul64wide(a, b, lo, hi);
    u64 nlo = s.lo + lo;
    u64 c = nlo < lo;
    u64 nhi = s.hi + hi;
    u32 c2 = nhi < hi;
    nhi += c;
    c2 += (nhi < c);
    s.lo = nlo; s.hi = nhi; s.ov += c2;
}
// (ov:hi:lo) mod p, canonical.  2^128 = 2^64 * 2^64 = (2^32-1)^2 = 2^64 - 2^33 + 1 = -2^32 (mod p)
LF_HD u64 acc_reduce(const Acc &s) {
    u64 r = fq_canon(fq_reduce128_loose(s.lo, s.hi));
    if (s.ov) r = fq_sub(r, (u64)s.ov << 32);
    return r;
}

// ---- partial-product accumulator --------------------------------------------------------------------------
// A sum of 64x64 products is kept as three 64-bit sums of 32x32 partial products (weights 2^0, 2^32, 2^64) plus
// carry counters.  On gfx950 one partial product costs exactly two instructions: v_mad_u64_u32 (multiply-add with
// carry-out in VCC) + v_addc_co_u32 into the counter -- no shifting/recombination, no reduction until the end.
struct AccP {
    u64 s00, s01, s11;
    u32 c00, c01, c11;
};
LF_HD void accp_zero(AccP &a) { a.s00 = a.s01 = a.s11 = 0; a.c00 = a.c01 = a.c11 = 0; }
LF_HD void mad_cc(u64 &acc, u32 &cnt, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // carry through a compiler-allocated SGPR pair (VOP3 forms) instead of VCC: independent accumulator chains do not
    // serialise on the single VCC register and can be interleaved by the scheduler
    unsigned long long cy;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2" : "+v"(acc), "+v"(cnt), "=&s"(cy) : "v"(a), "v"(b));
#else
    u64 p = (u64)a * b, n = acc + p;
    cnt += (n < p);
    acc = n;
#endif
}
// first product of a sum: no carries can occur yet (compiler keeps untouched counters as constants)
LF_HD void accp_set(AccP &s, u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    s.s00 = (u64)a0 * b0;
    s.s01 = (u64)a0 * b1;
    s.s11 = (u64)a1 * b1;
    s.c00 = 0; s.c01 = 0; s.c11 = 0;
    mad_cc(s.s01, s.c01, a1, b0);
}
LF_HD void accp_mad(AccP &s, u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(LF_ACCP_SPLIT_ASM)
    // the four partial products of one 64x64 product in ONE asm statement: the compiler pads every asm statement that defines an SGPR
    // with an s_nop (hazard recogniser), which cost 4 s_nop per product with one statement per partial product
    unsigned long long cy;
    asm("v_mad_u64_u32 %0, %6, %7, %9, %0\n\tv_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "v_mad_u64_u32 %1, %6, %7, %10, %1\n\tv_addc_co_u32_e64 %4, %6, 0, %4, %6\n\t"
        "v_mad_u64_u32 %1, %6, %8, %9, %1\n\tv_addc_co_u32_e64 %4, %6, 0, %4, %6\n\t"
        "v_mad_u64_u32 %2, %6, %8, %10, %2\n\tv_addc_co_u32_e64 %5, %6, 0, %5, %6"
        : "+v"(s.s00), "+v"(s.s01), "+v"(s.s11), "+v"(s.c00), "+v"(s.c01), "+v"(s.c11), "=&s"(cy)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
#else
    mad_cc(s.s00, s.c00, a0, b0);
    mad_cc(s.s01, s.c01, a0, b1);
    mad_cc(s.s01, s.c01, a1, b0);
    mad_cc(s.s11, s.c11, a1, b1);
#endif
}
// value mod p, canonical:  s00 + s01*2^32 + s11*2^64 + c00*2^64 + c01*2^96 + c11*2^128
// with 2^64 = 2^32-1, 2^96 = -1, 2^128 = -2^32 (mod p)
LF_HD u64 accp_reduce(const AccP &s) {
    u64 r = fq_canon(s.s00);
    r = fq_add(r, fq_canon(fq_reduce128_loose(s.s01 << 32, s.s01 >> 32)));
    r = fq_add(r, fq_canon(fq_reduce128_loose(0, s.s11)));
    r = fq_add(r, ((u64)s.c00 << 32) - s.c00);          // c00 * (2^32 - 1) < p
    r = fq_sub(r, (u64)s.c01);
    r = fq_sub(r, (u64)s.c11 << 32);
    return r;
}

// ---- fast F_{p^3} product for NU = 2^40 ---------------------------------------------------------------------
// A column sum S (AccP) is congruent to L + 2^32 H with small signed L, H:
//   S = a0 + 2^32 a1 + 2^32 b0 + 2^64 (b1 + d0 + c00) + 2^96 (d1 + c01) + 2^128 c11     (32-bit halves of s00,s01,s11)
//     = L + 2^32 H,  L = a0 - T - U,  H = a1 + b0 + T - c11,  T = b1 + d0 + c00,  U = d1 + c01
// using 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32 (mod p).  Multiplying by NU = 2^40 is a shift of that linear form:
//   2^40 (L + 2^32 H) = 2^40 (L + H) - 2^8 H     (2^72 = 2^40 - 2^8).
struct LH {
    int64_t l, h;
};
LF_HD LH accp_lh(const AccP &s) {
    int64_t T = (int64_t)(s.s01 >> 32) + (int64_t)(u32)s.s11 + (int64_t)s.c00;
    int64_t U = (int64_t)(s.s11 >> 32) + (int64_t)s.c01;
    LH r;
    r.l = (int64_t)(u32)s.s00 - T - U;
    r.h = (int64_t)(s.s00 >> 32) + (int64_t)(u32)s.s01 + T - (int64_t)s.c11;
    return r;
}
// signed value  lo64 + 2^64 * hi  (|hi| small) -> canonical residue
LF_HD u64 fq_from_s128(u64 lo, int64_t hi) {
    // 2^64 = eps: add hi*eps to lo, fix the single possible wrap in either direction
    int64_t t = (int64_t)((u64)hi << 32) - hi;  // hi * (2^32 - 1), |hi| < 2^30 -> no overflow
    u64 s = lo + (u64)t;
    if (t >= 0) {
        if (s < lo) s += LF_EPS;       // wrapped past 2^64: -p
    } else {
        if (s > lo) s -= LF_EPS;       // wrapped below 0: +p
    }
    return fq_canon(s);
}
// value = base + 2^32 * h32 + 2^40 * h40 as a signed 128-bit integer, reduced (all inputs small signed, < 2^40)
LF_HD u64 fq_from_lin(int64_t base, int64_t h32, int64_t h40) {
    typedef __int128 i128;
    i128 v = (i128)base + ((i128)h32 << 32) + ((i128)h40 << 40);
    return fq_from_s128((u64)v, (int64_t)(v >> 64));
}
LF_HD Fq3 fq3_from_columns_2p40(const AccP *s) {
    LH c0 = accp_lh(s[0]), c1 = accp_lh(s[1]), c2 = accp_lh(s[2]), c3 = accp_lh(s[3]), c4 = accp_lh(s[4]);
    Fq3 r;
    r.c[0] = fq_from_lin(c0.l - (c3.h << 8), c0.h, c3.l + c3.h);
    r.c[1] = fq_from_lin(c1.l - (c4.h << 8), c1.h, c4.l + c4.h);
    r.c[2] = fq_from_lin(c2.l, c2.h, 0);
    return r;
}
LF_HD Fq3 fq3_mul_2p40(Fq3 a, Fq3 b) {
    AccP s[5];
    accp_set(s[0], a.c[0], b.c[0]);
    accp_set(s[1], a.c[0], b.c[1]); accp_mad(s[1], a.c[1], b.c[0]);
    accp_set(s[2], a.c[0], b.c[2]); accp_mad(s[2], a.c[1], b.c[1]); accp_mad(s[2], a.c[2], b.c[0]);
    accp_set(s[3], a.c[1], b.c[2]); accp_mad(s[3], a.c[2], b.c[1]);
    accp_set(s[4], a.c[2], b.c[2]);
    return fq3_from_columns_2p40(s);
}

// Lazy sum of F_{p^3} products for NU = 2^40: each product's five columns are folded to their (L,H) linear forms and
// added as plain 64-bit integers (|L|,|H| < 2^36 per product -> 2^27 products fit); one reduction at the very end.
struct LH5 {
    LH c[5];
};
LF_HD void lh5_zero(LH5 &a) {
#pragma unroll
    for (int i = 0; i < 5; i++) { a.c[i].l = 0; a.c[i].h = 0; }
}
LF_HD void lh5_mac(LH5 &acc, Fq3 a, Fq3 b) {
    AccP s[5];
    accp_set(s[0], a.c[0], b.c[0]);
    accp_set(s[1], a.c[0], b.c[1]); accp_mad(s[1], a.c[1], b.c[0]);
    accp_set(s[2], a.c[0], b.c[2]); accp_mad(s[2], a.c[1], b.c[1]); accp_mad(s[2], a.c[2], b.c[0]);
    accp_set(s[3], a.c[1], b.c[2]); accp_mad(s[3], a.c[2], b.c[1]);
    accp_set(s[4], a.c[2], b.c[2]);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        LH t = accp_lh(s[i]);
        acc.c[i].l += t.l;
        acc.c[i].h += t.h;
    }
}
// acc += a b + c d: the partial products of BOTH F_{p^3} products go into the same five column sums before they are folded to their (L, H) forms -- one
// fold (the larger half of a lazy product's instructions: zero-extensions and 64-bit adds) per two products
LF_HD void lh5_mac2(LH5 &acc, Fq3 a, Fq3 b, Fq3 c, Fq3 d) {
    AccP s[5];
    accp_set(s[0], a.c[0], b.c[0]);
    accp_set(s[1], a.c[0], b.c[1]); accp_mad(s[1], a.c[1], b.c[0]);
    accp_set(s[2], a.c[0], b.c[2]); accp_mad(s[2], a.c[1], b.c[1]); accp_mad(s[2], a.c[2], b.c[0]);
    accp_set(s[3], a.c[1], b.c[2]); accp_mad(s[3], a.c[2], b.c[1]);
    accp_set(s[4], a.c[2], b.c[2]);
    accp_mad(s[0], c.c[0], d.c[0]);
    accp_mad(s[1], c.c[0], d.c[1]); accp_mad(s[1], c.c[1], d.c[0]);
    accp_mad(s[2], c.c[0], d.c[2]); accp_mad(s[2], c.c[1], d.c[1]); accp_mad(s[2], c.c[2], d.c[0]);
    accp_mad(s[3], c.c[1], d.c[2]); accp_mad(s[3], c.c[2], d.c[1]);
    accp_mad(s[4], c.c[2], d.c[2]);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        LH t = accp_lh(s[i]);
        acc.c[i].l += t.l;
        acc.c[i].h += t.h;
    }
}
// acc += sum_{i < N} a[i] b[i]: N products per fold of the column sums (N = 2: lh5_mac2; larger N where the operands fit in registers)
template <int N>
LF_HD void lh5_macn(LH5 &acc, const Fq3 (&a)[N], const Fq3 (&b)[N]) {
    AccP s[5];
    accp_set(s[0], a[0].c[0], b[0].c[0]);
    accp_set(s[1], a[0].c[0], b[0].c[1]); accp_mad(s[1], a[0].c[1], b[0].c[0]);
    accp_set(s[2], a[0].c[0], b[0].c[2]); accp_mad(s[2], a[0].c[1], b[0].c[1]); accp_mad(s[2], a[0].c[2], b[0].c[0]);
    accp_set(s[3], a[0].c[1], b[0].c[2]); accp_mad(s[3], a[0].c[2], b[0].c[1]);
    accp_set(s[4], a[0].c[2], b[0].c[2]);
#pragma unroll
    for (int i = 1; i < N; i++) {
        accp_mad(s[0], a[i].c[0], b[i].c[0]);
        accp_mad(s[1], a[i].c[0], b[i].c[1]); accp_mad(s[1], a[i].c[1], b[i].c[0]);
        accp_mad(s[2], a[i].c[0], b[i].c[2]); accp_mad(s[2], a[i].c[1], b[i].c[1]); accp_mad(s[2], a[i].c[2], b[i].c[0]);
        accp_mad(s[3], a[i].c[1], b[i].c[2]); accp_mad(s[3], a[i].c[2], b[i].c[1]);
        accp_mad(s[4], a[i].c[2], b[i].c[2]);
    }
#pragma unroll
    for (int i = 0; i < 5; i++) {
        LH t = accp_lh(s[i]);
        acc.c[i].l += t.l;
        acc.c[i].h += t.h;
    }
}
// signed wide value base + 2^32 h32 + 2^40 h40 with |terms| up to ~2^62: split before shifting
LF_HD u64 fq_from_lin_wide(int64_t base, int64_t h32, int64_t h40) {
    typedef __int128 i128;
    i128 v = (i128)base + ((i128)h32 << 32) + ((i128)h40 << 40);   // |v| < 2^103
    // fold the high part: v = lo + 2^64 hi, hi up to 2^39 -> hi*eps needs 71 bits: fold twice
    u64 lo = (u64)v;
    i128 hi = v >> 64;
    i128 w = (i128)lo + (hi << 32) - hi;                            // |w| < 2^72
    return fq_from_s128((u64)w, (int64_t)(w >> 64));
}
LF_HD Fq3 lh5_finish(const LH5 &a) {
    Fq3 r;
    r.c[0] = fq_from_lin_wide(a.c[0].l - (a.c[3].h << 8), a.c[0].h, a.c[3].l + a.c[3].h);
    r.c[1] = fq_from_lin_wide(a.c[1].l - (a.c[4].h << 8), a.c[1].h, a.c[4].l + a.c[4].h);
    r.c[2] = fq_from_lin_wide(a.c[2].l, a.c[2].h, 0);
    return r;
}

// generic-NU product: schoolbook, 9 base multiplications, lazy 128-bit column sums, 5 reductions
template <bool NU2P40>
LF_HD Fq3 fq3_mul(Fq3 a, Fq3 b, u64 nu) {
    if (NU2P40) return fq3_mul_2p40(a, b);
    Acc s0, s1, s2, s3, s4;
    acc_set(s0, a.c[0], b.c[0]);
    acc_set(s1, a.c[0], b.c[1]); acc_mad(s1, a.c[1], b.c[0]);
    acc_set(s2, a.c[0], b.c[2]); acc_mad(s2, a.c[1], b.c[1]); acc_mad(s2, a.c[2], b.c[0]);
    acc_set(s3, a.c[1], b.c[2]); acc_mad(s3, a.c[2], b.c[1]);
    acc_set(s4, a.c[2], b.c[2]);
    Fq3 r;
    r.c[0] = fq_add(acc_reduce(s0), fq_mul_nu<NU2P40>(acc_reduce(s3), nu));
    r.c[1] = fq_add(acc_reduce(s1), fq_mul_nu<NU2P40>(acc_reduce(s4), nu));
    r.c[2] = acc_reduce(s2);
    return r;
}
// square
template <bool NU2P40>
LF_HD Fq3 fq3_sqr(Fq3 a, u64 nu) {
    if (NU2P40) return fq3_mul_2p40(a, a);
    u64 d01 = fq_add(a.c[0], a.c[0]), d1 = fq_add(a.c[1], a.c[1]);
    Acc s0, s1, s2, s3, s4;
    acc_set(s0, a.c[0], a.c[0]);
    acc_set(s1, d01, a.c[1]);
    acc_set(s2, d01, a.c[2]); acc_mad(s2, a.c[1], a.c[1]);
    acc_set(s3, d1, a.c[2]);
    acc_set(s4, a.c[2], a.c[2]);
    Fq3 r;
    r.c[0] = fq_add(acc_reduce(s0), fq_mul_nu<NU2P40>(acc_reduce(s3), nu));
    r.c[1] = fq_add(acc_reduce(s1), fq_mul_nu<NU2P40>(acc_reduce(s4), nu));
    r.c[2] = acc_reduce(s2);
    return r;
}
LF_HD Fq3 fq3_mul_fq(Fq3 a, u64 s) { return fq3_make(fq_mul(a.c[0], s), fq_mul(a.c[1], s), fq_mul(a.c[2], s)); }
// times a small signed integer |k| < 2^31
LF_HD Fq3 fq3_mul_small(Fq3 a, int k) {
    u64 m = (u64)(k < 0 ? -k : k);
    Fq3 r = fq3_make(fq_mul(a.c[0], m), fq_mul(a.c[1], m), fq_mul(a.c[2], m));
    return k < 0 ? fq3_neg(r) : r;
}

template <bool NU2P40>
LF_HD Fq3 fq3_inv(Fq3 a, u64 nu) {
    // norm-based inverse of a cubic extension element
    u64 t0 = fq_sub(fq_mul(a.c[0], a.c[0]), fq_mul_nu<NU2P40>(fq_mul(a.c[1], a.c[2]), nu));
    u64 t1 = fq_sub(fq_mul_nu<NU2P40>(fq_mul(a.c[2], a.c[2]), nu), fq_mul(a.c[0], a.c[1]));
    u64 t2 = fq_sub(fq_mul(a.c[1], a.c[1]), fq_mul(a.c[0], a.c[2]));
    u64 n = fq_add(fq_mul(a.c[0], t0), fq_mul_nu<NU2P40>(fq_add(fq_mul(a.c[2], t1), fq_mul(a.c[1], t2)), nu));
    u64 ni = fq_inv(n);
    return fq3_make(fq_mul(t0, ni), fq_mul(t1, ni), fq_mul(t2, ni));
}

}  // namespace lf

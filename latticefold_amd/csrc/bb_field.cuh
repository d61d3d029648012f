// bb_field.cuh -- BabyBear F_p (p = 15*2^27 + 1, 31 bits) and F_{p^9} = F_p[Y]/(Y^9 - nu) arithmetic for host and gfx950.
//
// Device representation ("alt-prime 31-bit Montgomery path", BASELINE configs[2]): a residue x is stored as the CENTRED
// Montgomery word  x~ = centre(x * 2^32 mod p)  in [-H, H], H = (p-1)/2 < 2^30.  Two centred words multiply to < 2^59.82,
// so the NINE products of one column of an F_{p^9} schoolbook product (9 * H^2 < 2^63) accumulate exactly in one signed
// 64-bit register with v_mad_i64_i32 and are Montgomery-reduced once per column.  gfx950 has no 64x64 multiplier and
// v_mad_*64_*32 runs at quarter rate, so the cost model is "mads per product"; see DESIGN.md (BabyBear section).
//
// Reference semantics: stark-rings BabyBear Fq / Fq9 (absent dependency) on canonical residues.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BB_HD __host__ __device__ __forceinline__
#else
#define BB_HD inline
#endif

namespace lfbb {

typedef int32_t fe;       // centred Montgomery residue
typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;

constexpr u32 BB_P = 2013265921u;            // 15 * 2^27 + 1
constexpr int32_t BB_H = (int32_t)((BB_P - 1) / 2);
constexpr u32 BB_PINV = 0x88000001u;          // p^-1 mod 2^32  ((1+a)(1-a) = 1 - a^2, a = 15*2^27, a^2 = 0 mod 2^32)
constexpr int D = 72;                         // ring degree, Phi_216 = X^72 - X^36 + 1 (cyclotomic-rings/src/rings/babybear.rs:19)
constexpr int TAU = 9;                        // extension degree of a slot
constexpr int SLOTS = 8;
constexpr int RE = 72;                        // words per ring element

constexpr u64 cmod(u64 a) { return a % BB_P; }
constexpr u64 BB_R = cmod(1ull << 32);                     // 2^32 mod p
constexpr u64 BB_R2 = cmod(BB_R * BB_R);                    // 2^64 mod p
constexpr int32_t ccentre(u64 c) { return c > (u64)BB_H ? (int32_t)((i64)c - (i64)BB_P) : (int32_t)c; }
constexpr fe BB_ONE = ccentre(BB_R);                        // Montgomery form of 1
constexpr fe BB_R2C = ccentre(BB_R2);                       // multiply by this (then reduce) to enter Montgomery form

BB_HD fe centre(int32_t s) {   // s in (-1.5p, 1.5p) -> [-H, H]
    s -= (s > BB_H) ? (int32_t)BB_P : 0;
    s += (s < -BB_H) ? (int32_t)BB_P : 0;
    return s;
}
BB_HD int32_t mulhi_i32(int32_t a, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __mulhi(a, b);
#else
    return (int32_t)(((i64)a * (i64)b) >> 32);
#endif
}
// Montgomery reduction of a signed 64-bit T with |T| <= 9*H^2 (< 2^63): returns T * 2^-32 mod p, centred.
BB_HD fe mred(i64 T) {
    int32_t thi = (int32_t)(T >> 32);
    u32 tlo = (u32)T;
    int32_t m = (int32_t)(tlo * BB_PINV);         // m*p = tlo (mod 2^32)
    int32_t q = mulhi_i32(m, (int32_t)BB_P);      // floor(m*p / 2^32); T - m*p = (thi - q) * 2^32 exactly
    thi = centre(thi);                            // thi*2^32 and (thi -+ p)*2^32 agree mod p; |thi| <= H afterwards
    return centre(thi - q);                       // |q| <= p/2
}
BB_HD fe fmul(fe a, fe b) { return mred((i64)a * (i64)b); }
BB_HD fe fadd(fe a, fe b) { return centre(a + b); }
BB_HD fe fsub(fe a, fe b) { return centre(a - b); }
BB_HD fe fneg(fe a) { return -a; }
BB_HD fe from_canon(u64 x) { return mred((i64)ccentre(x % BB_P) * (i64)BB_R2C); }
BB_HD fe from_int(i64 s) { return mred((s % (i64)BB_P) * (i64)BB_R2C); }   // any integer (host); small ints on device
BB_HD fe from_small(int32_t s) { return mred((i64)s * (i64)BB_R2C); }       // |s| < 2^31 on device without the %
BB_HD u32 to_canon(fe x) {
    int32_t t = mred((i64)x);
    return (u32)(t < 0 ? t + (int32_t)BB_P : t);
}
// x (centred, Montgomery) * small integer s, |s| <= 2^31 / ... : product fits, plain reduction is NOT Montgomery, so go
// through the Montgomery form of s
BB_HD fe fmul_small(fe a, int32_t s) { return fmul(a, from_small(s)); }

// ---- F_{p^9} ---------------------------------------------------------------------------------------------------
struct E9 { fe c[TAU]; };
BB_HD E9 e9_zero() { E9 r; for (int i = 0; i < TAU; i++) r.c[i] = 0; return r; }
BB_HD E9 e9_from_fe(fe a) { E9 r = e9_zero(); r.c[0] = a; return r; }
BB_HD E9 e9_add(const E9 &a, const E9 &b) { E9 r; for (int i = 0; i < TAU; i++) r.c[i] = fadd(a.c[i], b.c[i]); return r; }
BB_HD E9 e9_sub(const E9 &a, const E9 &b) { E9 r; for (int i = 0; i < TAU; i++) r.c[i] = fsub(a.c[i], b.c[i]); return r; }
BB_HD E9 e9_neg(const E9 &a) { E9 r; for (int i = 0; i < TAU; i++) r.c[i] = -a.c[i]; return r; }
BB_HD E9 e9_mul_fe(const E9 &a, fe s) { E9 r; for (int i = 0; i < TAU; i++) r.c[i] = fmul(a.c[i], s); return r; }
// b pre-multiplied by nu (hoist when b is loop invariant).  The default tables use nu = 2 (a non-cube mod p, so Y^9 - 2 is
// irreducible): the pre-multiplication is then a doubling + centring instead of a Montgomery product.  `nu` is a kernel
// argument, so the test is a uniform scalar branch; any other (data) nu takes the generic path.
constexpr fe BB_TWO = ccentre((2 * BB_R) % BB_P);   // Montgomery form of 2
template <bool NU2>
BB_HD E9 e9_times_nu_t(const E9 &b, fe nu) {
    E9 r;
    if (NU2) {
        for (int i = 0; i < TAU; i++) r.c[i] = centre(2 * b.c[i]);
    } else {
        for (int i = 0; i < TAU; i++) r.c[i] = fmul(b.c[i], nu);
    }
    return r;
}
BB_HD E9 e9_times_nu(const E9 &b, fe nu) { return e9_times_nu_t<false>(b, nu); }
// a * b with bn = nu * b:  c_k = sum_{i<=k} a_i b_{k-i} + sum_{i>k} a_i bn_{k+9-i}   (9 terms per column, one reduction)
BB_HD E9 e9_mul_pre(const E9 &a, const E9 &b, const E9 &bn) {
    E9 r;
#pragma unroll
    for (int k = 0; k < TAU; k++) {
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < TAU; i++) acc += (i64)a.c[i] * (i64)(i <= k ? b.c[k - i] : bn.c[k + TAU - i]);
        r.c[k] = mred(acc);
    }
    return r;
}
// the nine un-reduced column sums of a * b (each |T_k| <= 9 H^2); callers that add many products keep the high and low
// 32-bit halves of T_k in separate 64-bit sums and reduce once (mred(hi * 2^32 + lo) = hi + lo * 2^-32)
BB_HD void e9_mul_cols(const E9 &a, const E9 &b, const E9 &bn, i64 (&T)[TAU]) {
#pragma unroll
    for (int k = 0; k < TAU; k++) {
        i64 acc = 0;
#pragma unroll
        for (int i = 0; i < TAU; i++) acc += (i64)a.c[i] * (i64)(i <= k ? b.c[k - i] : bn.c[k + TAU - i]);
        T[k] = acc;
    }
}
BB_HD E9 e9_mul(const E9 &a, const E9 &b, fe nu) { return e9_mul_pre(a, b, e9_times_nu(b, nu)); }
// a^2 with an = nu * a: symmetric terms are formed once and doubled (45 instead of 81 mads); the column bound is unchanged
// (every ordered pair (i,j) still contributes |a_i a_j| once)
BB_HD E9 e9_sqr_pre(const E9 &a, const E9 &an) {
    E9 r;
#pragma unroll
    for (int k = 0; k < TAU; k++) {
        i64 off = 0, diag = 0;
#pragma unroll
        for (int i = 0; i < TAU; i++) {
            // ordered pairs (i, j) with i + j = k (j = k - i >= 0) or i + j = k + 9 (j = k + 9 - i <= 8)
            int j = i <= k ? k - i : k + TAU - i;
            fe bj = i <= k ? a.c[j] : an.c[j];
            if (i < j) off += (i64)a.c[i] * (i64)bj;
            else if (i == j) diag += (i64)a.c[i] * (i64)bj;
        }
        r.c[k] = mred(2 * off + diag);
    }
    return r;
}
BB_HD E9 e9_sqr(const E9 &a, fe nu) { return e9_sqr_pre(a, e9_times_nu(a, nu)); }
template <bool NU2> BB_HD E9 e9_mul_t(const E9 &a, const E9 &b, fe nu) { return e9_mul_pre(a, b, e9_times_nu_t<NU2>(b, nu)); }
template <bool NU2> BB_HD E9 e9_sqr_t(const E9 &a, fe nu) { return e9_sqr_pre(a, e9_times_nu_t<NU2>(a, nu)); }
// a * s where s is a plain small integer
BB_HD E9 e9_mul_small(const E9 &a, int32_t s) { return e9_mul_fe(a, from_small(s)); }
BB_HD bool e9_eq(const E9 &a, const E9 &b) { for (int i = 0; i < TAU; i++) if (a.c[i] != b.c[i]) return false; return true; }

struct E9Pre { E9 v, vn; };   // a constant together with nu * constant
BB_HD E9Pre e9_pre(const E9 &b, fe nu) { E9Pre r; r.v = b; r.vn = e9_times_nu(b, nu); return r; }
BB_HD E9 e9_mul(const E9 &a, const E9Pre &b) { return e9_mul_pre(a, b.v, b.vn); }

}  // namespace lfbb

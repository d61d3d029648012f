// lf_poseidon_simd.cc -- see lf_poseidon_simd.h.  Plain host C++ (compiled with the AVX-512 IFMA target for this file only;
// every entry point is reached through the run-time check psimd::supported()).
//
// Arithmetic: a 64-bit word a = a0 + 2^52 a1 (a1 < 2^12).  vpmadd52{l,h}uq multiply the low 52 bits of their operands, so the
// un-split word serves as a0.  A product a*b is the sum of
//      weight 2^0   : lo52(a0 b0)
//      weight 2^52  : hi52(a0 b0) + lo52(a0 b1) + lo52(a1 b0)
//      weight 2^104 : hi52(a0 b1) + hi52(a1 b0) + lo52(a1 b1)
// and sums of up to 24 products stay below 2^60 per weight class: seven IFMAs per product, no carry handling.  The value
// W0 + 2^52 W52 + 2^104 W104 is reduced once with 2^64 = 2^32 - 1, 2^96 = -1 (so 2^104 = -2^8) mod p.
#include "lf_poseidon_simd.h"

#include <immintrin.h>
#include <string.h>

namespace lf {
namespace psimd {

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef __m512i V;

namespace {
constexpr int W = 24, RF = 8, RP = 22;
constexpr u64 P = 0xFFFFFFFF00000001ULL, EPS = 0xFFFFFFFFULL;

struct Tables {
    alignas(64) u64 mds0[W][W], mds1[W][W];     // [j][i] = M[i][j] and its top 12 bits
    alignas(64) u64 post0[W][W], post1[W][W];   // the deferred factor embedded as diag(1, post)
    alignas(64) u64 arkf[RF][W];                // constants of the full rounds
    alignas(64) u64 cst[RP][W];                 // partial-round constants, lane 0 cleared (kept in cst0)
    alignas(64) u64 row0[RP][W], row1[RP][W];   // lane 0 cleared (e00 is applied on the scalar side)
    alignas(64) u64 col0[RP][W], col1[RP][W];   // lane 0 cleared
    u64 cst0[RP], e00[RP];
};
Tables T;

inline u64 canon(u64 a) { return a >= P ? a - P : a; }
inline u64 reduce128(u64 lo, u64 hi) {   // canonical
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    u64 t1 = (hl << 32) - hl;
    u64 r = t0 + t1;
    if (r < t1) r += EPS;
    return canon(r);
}
inline u64 mulmod(u64 a, u64 b) {
    u128 pr = (u128)a * b;
    return reduce128((u64)pr, (u64)(pr >> 64));
}
inline u64 addmod(u64 a, u64 b) {
    u64 r = a + b;
    if (r < a || r >= P) r -= P;
    return r;
}
inline u64 submod(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
inline u64 sbox(u64 x) {
    u64 x2 = mulmod(x, x), x3 = mulmod(x2, x), x4 = mulmod(x2, x2);
    return mulmod(x4, x3);
}

// W0 (< 2^60) + 2^52 W52 (W52 < 2^60) - 2^8 W104 (W104 < 2^30)  ->  canonical residue
inline V reduce(V w0, V w52, V w104) {
    const V eps = _mm512_set1_epi64((long long)EPS), pp = _mm512_set1_epi64((long long)P), one = _mm512_set1_epi64(1);
    V sh = _mm512_slli_epi64(w52, 52);
    V lo = _mm512_add_epi64(w0, sh);
    __mmask8 c1 = _mm512_cmplt_epu64_mask(lo, sh);
    V hi = _mm512_srli_epi64(w52, 12);
    hi = _mm512_mask_add_epi64(hi, c1, hi, one);            // < 2^48 + 1
    V hh = _mm512_srli_epi64(hi, 32), hl = _mm512_and_si512(hi, eps);
    __mmask8 b = _mm512_cmplt_epu64_mask(lo, hh);
    V t0 = _mm512_sub_epi64(lo, hh);
    t0 = _mm512_mask_sub_epi64(t0, b, t0, eps);             // borrow: the wrap added 2^64 = eps
    V t1 = _mm512_sub_epi64(_mm512_slli_epi64(hl, 32), hl); // hl * (2^32 - 1)
    V r = _mm512_add_epi64(t0, t1);
    __mmask8 c = _mm512_cmplt_epu64_mask(r, t1);
    r = _mm512_mask_add_epi64(r, c, r, eps);
    V s = _mm512_slli_epi64(w104, 8);
    __mmask8 b2 = _mm512_cmplt_epu64_mask(r, s);
    r = _mm512_sub_epi64(r, s);
    r = _mm512_mask_sub_epi64(r, b2, r, eps);
    __mmask8 g = _mm512_cmpge_epu64_mask(r, pp);
    return _mm512_mask_sub_epi64(r, g, r, pp);
}
inline V vmul(V a, V b) {
    const V z = _mm512_setzero_si512();
    V a1 = _mm512_srli_epi64(a, 52), b1 = _mm512_srli_epi64(b, 52);
    V w0 = _mm512_madd52lo_epu64(z, a, b);
    V w52 = _mm512_madd52hi_epu64(z, a, b);
    V w52b = _mm512_madd52lo_epu64(z, a, b1);
    V w52c = _mm512_madd52lo_epu64(z, a1, b);
    V w104 = _mm512_madd52hi_epu64(z, a, b1);
    V w104b = _mm512_madd52hi_epu64(z, a1, b);
    V w104c = _mm512_madd52lo_epu64(z, a1, b1);
    return reduce(w0, _mm512_add_epi64(_mm512_add_epi64(w52, w52b), w52c), _mm512_add_epi64(_mm512_add_epi64(w104, w104b), w104c));
}
inline V vadd(V a, V b) {   // canonical + canonical -> canonical
    const V eps = _mm512_set1_epi64((long long)EPS), pp = _mm512_set1_epi64((long long)P);
    V r = _mm512_add_epi64(a, b);
    __mmask8 c = _mm512_cmplt_epu64_mask(r, a);
    __mmask8 g = _mm512_cmpge_epu64_mask(r, pp);
    r = _mm512_mask_add_epi64(r, c, r, eps);                // wrapped: + 2^64 - p
    return _mm512_mask_sub_epi64(r, (__mmask8)(g & ~c), r, pp);
}

// x <- M x for a 24 x 24 matrix given column-wise (t0[j] = column j, t1[j] = its top 12 bits)
inline void matvec(const u64 (*t0)[W], const u64 (*t1)[W], V x[3]) {
    alignas(64) u64 xl[W], xh[W];
    for (int g = 0; g < 3; g++) {
        _mm512_store_si512((void *)(xl + 8 * g), x[g]);
        _mm512_store_si512((void *)(xh + 8 * g), _mm512_srli_epi64(x[g], 52));
    }
    const V z = _mm512_setzero_si512();
    V a0[3], a52[3], a52b[3], a52c[3], a104[3], a104b[3], a104c[3];
    for (int g = 0; g < 3; g++) a0[g] = a52[g] = a52b[g] = a52c[g] = a104[g] = a104b[g] = a104c[g] = z;
    for (int j = 0; j < W; j++) {
        V b = _mm512_set1_epi64((long long)xl[j]), b1 = _mm512_set1_epi64((long long)xh[j]);
#pragma GCC unroll 3
        for (int g = 0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(t0[j] + 8 * g)), m1 = _mm512_load_si512((const void *)(t1[j] + 8 * g));
            a0[g] = _mm512_madd52lo_epu64(a0[g], m, b);
            a52[g] = _mm512_madd52hi_epu64(a52[g], m, b);
            a52b[g] = _mm512_madd52lo_epu64(a52b[g], m, b1);
            a52c[g] = _mm512_madd52lo_epu64(a52c[g], m1, b);
            a104[g] = _mm512_madd52hi_epu64(a104[g], m, b1);
            a104b[g] = _mm512_madd52hi_epu64(a104b[g], m1, b);
            a104c[g] = _mm512_madd52lo_epu64(a104c[g], m1, b1);
        }
    }
    for (int g = 0; g < 3; g++)
        x[g] = reduce(a0[g], _mm512_add_epi64(_mm512_add_epi64(a52[g], a52b[g]), a52c[g]),
                      _mm512_add_epi64(_mm512_add_epi64(a104[g], a104b[g]), a104c[g]));
}

inline void full_round(V x[3], const u64 *ark) {
    V t[3], x2[3], x3[3], x4[3];
    for (int g = 0; g < 3; g++) t[g] = vadd(x[g], _mm512_load_si512((const void *)(ark + 8 * g)));
    for (int g = 0; g < 3; g++) x2[g] = vmul(t[g], t[g]);
    for (int g = 0; g < 3; g++) x3[g] = vmul(x2[g], t[g]);
    for (int g = 0; g < 3; g++) x4[g] = vmul(x2[g], x2[g]);
    for (int g = 0; g < 3; g++) x[g] = vmul(x4[g], x3[g]);
    matvec(T.mds0, T.mds1, x);
}
}  // namespace

bool supported() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq");
    return ok;
}

void build(const u64 *ark, const u64 *mds, const u64 *cst, const u64 *e00, const u64 *row, const u64 *col, const u64 *post) {
    memset(&T, 0, sizeof(T));
    for (int i = 0; i < W; i++)
        for (int j = 0; j < W; j++) {
            T.mds0[j][i] = mds[i * W + j];
            T.mds1[j][i] = mds[i * W + j] >> 52;
            u64 e = (i == 0 || j == 0) ? (u64)(i == j) : post[(i - 1) * (W - 1) + (j - 1)];
            T.post0[j][i] = e;
            T.post1[j][i] = e >> 52;
        }
    for (int r = 0; r < RF; r++) {
        int src = r < RF / 2 ? r : RP + r;
        memcpy(T.arkf[r], ark + (size_t)src * W, W * 8);
    }
    for (int r = 0; r < RP; r++) {
        T.cst0[r] = cst[r * W];
        T.e00[r] = e00[r];
        for (int i = 1; i < W; i++) {
            T.cst[r][i] = cst[r * W + i];
            T.row0[r][i] = row[r * (W - 1) + i - 1];
            T.row1[r][i] = T.row0[r][i] >> 52;
            T.col0[r][i] = col[r * (W - 1) + i - 1];
            T.col1[r][i] = T.col0[r][i] >> 52;
        }
    }
}

void permute(u64 st[24]) {
    V x[3];
    for (int g = 0; g < 3; g++) x[g] = _mm512_loadu_si512((const void *)(st + 8 * g));
    for (int r = 0; r < RF / 2; r++) full_round(x, T.arkf[r]);
    // partial rounds: word 0 lives in a scalar register, lane 0 of x[0] is kept at zero
    u64 s0 = (u64)_mm_cvtsi128_si64(_mm512_castsi512_si128(x[0]));
    x[0] = _mm512_maskz_mov_epi64(0xFE, x[0]);
    const V z = _mm512_setzero_si512(), m52 = _mm512_set1_epi64((long long)((1ULL << 52) - 1));
    for (int r = 0; r < RP; r++) {
        V xs[3], xh[3];
        for (int g = 0; g < 3; g++) {
            xs[g] = vadd(x[g], _mm512_load_si512((const void *)(T.cst[r] + 8 * g)));
            xh[g] = _mm512_srli_epi64(xs[g], 52);
        }
        // row . xs (lanes 1..23), independent of the S-box of word 0
        V d0 = z, d52 = z, d52b = z, d52c = z, d104 = z, d104b = z, d104c = z;
        for (int g = 0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(T.row0[r] + 8 * g)), m1 = _mm512_load_si512((const void *)(T.row1[r] + 8 * g));
            d0 = _mm512_madd52lo_epu64(d0, m, xs[g]);
            d52 = _mm512_madd52hi_epu64(d52, m, xs[g]);
            d52b = _mm512_madd52lo_epu64(d52b, m, xh[g]);
            d52c = _mm512_madd52lo_epu64(d52c, m1, xs[g]);
            d104 = _mm512_madd52hi_epu64(d104, m, xh[g]);
            d104b = _mm512_madd52hi_epu64(d104b, m1, xs[g]);
            d104c = _mm512_madd52lo_epu64(d104c, m1, xh[g]);
        }
        u64 w0 = (u64)_mm512_reduce_add_epi64(d0);
        u64 w52 = (u64)_mm512_reduce_add_epi64(_mm512_add_epi64(_mm512_add_epi64(d52, d52b), d52c));
        u64 w104 = (u64)_mm512_reduce_add_epi64(_mm512_add_epi64(_mm512_add_epi64(d104, d104b), d104c));
        u128 dv = (u128)w0 + ((u128)w52 << 52);
        u64 dot = submod(reduce128((u64)dv, (u64)(dv >> 64)), w104 << 8);
        // S-box of word 0, then y0 = e00 x0 + dot, y_i = col_i x0 + xs_i
        u64 x0 = sbox(addmod(s0, T.cst0[r]));
        V b = _mm512_set1_epi64((long long)x0), b1 = _mm512_set1_epi64((long long)(x0 >> 52));
        for (int g = 0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(T.col0[r] + 8 * g)), m1 = _mm512_load_si512((const void *)(T.col1[r] + 8 * g));
            V a0 = _mm512_madd52lo_epu64(_mm512_and_si512(xs[g], m52), m, b);
            V a52 = _mm512_madd52hi_epu64(xh[g], m, b);
            V a52b = _mm512_madd52lo_epu64(z, m, b1);
            V a52c = _mm512_madd52lo_epu64(z, m1, b);
            V a104 = _mm512_madd52hi_epu64(z, m, b1);
            V a104b = _mm512_madd52hi_epu64(z, m1, b);
            V a104c = _mm512_madd52lo_epu64(z, m1, b1);
            x[g] = reduce(a0, _mm512_add_epi64(_mm512_add_epi64(a52, a52b), a52c), _mm512_add_epi64(_mm512_add_epi64(a104, a104b), a104c));
        }
        s0 = addmod(mulmod(T.e00[r], x0), dot);
    }
    x[0] = _mm512_mask_set1_epi64(x[0], 0x01, (long long)s0);
    matvec(T.post0, T.post1, x);
    for (int r = RF / 2; r < RF; r++) full_round(x, T.arkf[r]);
    for (int g = 0; g < 3; g++) _mm512_storeu_si512((void *)(st + 8 * g), x[g]);
}

}  // namespace psimd
}  // namespace lf

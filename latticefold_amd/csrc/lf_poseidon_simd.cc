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

constexpr int NX = W + RP;   // columns of the closing map: 24 state words + 22 S-box outputs
struct Tables {
    alignas(64) u64 mds0[W][W], mds1[W][W];     // [j][i] = M[i][j] and its top 12 bits
    alignas(64) u64 arkf[RF][W];                // constants of the full rounds
    // partial rounds in scalar form (see permute): D = SX x, s0_{r+1} = D_r + K_r + sum_{i<=r} G[r][i] X_i,
    // closing map  state' = FIN [x ; X] + FK
    alignas(64) u64 sx0[W][W], sx1[W][W];       // [j][r] = coefficient of state word j in D_r (column 0 and lanes >= 22 zero)
    alignas(64) u64 fin0[NX][W], fin1[NX][W];   // [j][i]: columns 0..23 state words, 24..45 the S-box outputs X_r; lane 0 zero
    alignas(64) u64 fk0[W], fk1[W];             // constant of the closing map (52-bit / top-12-bit halves)
    u64 cst0[RP], K[RP], G[RP][RP];
};
Tables T;

// scalar helpers.  The word-0 chain of the partial rounds is latency bound, so its products are reduced only "loosely" (any
// value below 2^64 is a valid multiplicand) with the shortest dependent sequence: the borrow of lo - hh is predictably absent
// (it needs lo < 2^32) and is a branch, the carry of the final add is data dependent and handled with a mask.
inline u64 canon(u64 a) { return a - (P & (0 - (u64)(a >= P))); }
inline u64 reduce128_loose(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & EPS;
    u64 t0 = lo - hh, r;
    if (__builtin_expect(lo < hh, 0)) t0 -= EPS;
    u64 t1 = (hl << 32) - hl;
    u64 c = __builtin_add_overflow(t0, t1, &r);
    return r + (EPS & (0 - c));
}
inline u64 reduce128(u64 lo, u64 hi) { return canon(reduce128_loose(lo, hi)); }
inline u64 mulmod(u64 a, u64 b) {   // canonical
    u128 pr = (u128)a * b;
    return reduce128((u64)pr, (u64)(pr >> 64));
}
inline u64 mul_loose(u64 a, u64 b) {   // any a, b; result below 2^64
    u128 pr = (u128)a * b;
    return reduce128_loose((u64)pr, (u64)(pr >> 64));
}
inline u64 addmod(u64 a, u64 b) {   // canonical operands
    u64 r;
    u64 c = __builtin_add_overflow(a, b, &r);
    return r - (P & (0 - (c | (u64)(r >= P))));
}
inline u64 add_loose(u64 a, u64 b) {   // a below 2^64, b canonical; result below 2^64
    u64 r;
    u64 c = __builtin_add_overflow(a, b, &r);
    return r + (EPS & (0 - c));
}
inline u64 submod(u64 a, u64 b) { return a - b + (P & (0 - (u64)(a < b))); }
inline u64 sbox_loose(u64 x) {
    u64 x2 = mul_loose(x, x), x3 = mul_loose(x2, x), x4 = mul_loose(x2, x2);
    return mul_loose(x4, x3);
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

// out = sum_j col_j * x_j (+ seed) for ncols columns given column-wise (t0[j] = column j, t1[j] = its top 12 bits);
// xl/xh = the words x_j and their top 12 bits
inline void matvec_n(const u64 (*t0)[W], const u64 (*t1)[W], int ncols, const u64 *xl, const u64 *xh, const V *seed0, const V *seed52, V out[3]) {
    const V z = _mm512_setzero_si512();
    V a0[3], a52[3], a52b[3], a52c[3], a104[3], a104b[3], a104c[3];
    for (int g = 0; g < 3; g++) {
        a0[g] = seed0 ? seed0[g] : z;
        a52[g] = seed52 ? seed52[g] : z;
        a52b[g] = a52c[g] = a104[g] = a104b[g] = a104c[g] = z;
    }
    for (int j = 0; j < ncols; j++) {
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
        out[g] = reduce(a0[g], _mm512_add_epi64(_mm512_add_epi64(a52[g], a52b[g]), a52c[g]),
                        _mm512_add_epi64(_mm512_add_epi64(a104[g], a104b[g]), a104c[g]));
}
inline void split_words(const V x[3], u64 *xl, u64 *xh) {
    for (int g = 0; g < 3; g++) {
        _mm512_store_si512((void *)(xl + 8 * g), x[g]);
        _mm512_store_si512((void *)(xh + 8 * g), _mm512_srli_epi64(x[g], 52));
    }
}
// x <- M x for a 24 x 24 matrix
inline void matvec(const u64 (*t0)[W], const u64 (*t1)[W], V x[3]) {
    alignas(64) u64 xl[W], xh[W];
    split_words(x, xl, xh);
    matvec_n(t0, t1, W, xl, xh, nullptr, nullptr, x);
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
        }
    for (int r = 0; r < RF; r++) {
        int src = r < RF / 2 ? r : RP + r;
        memcpy(T.arkf[r], ark + (size_t)src * W, W * 8);
    }
    // Symbolic run of the 22 sparse partial rounds.  Every state word 1..23 is an affine form over
    //   [ x_1..x_23 (words on entry) | X_0..X_21 (S-box outputs of word 0) | 1 ]
    // because a partial round is  xs = state[1..] + cst_r,  X_r = sbox(s0 + c0_r),  s0' = e00_r X_r + row_r . xs,
    // state'[1..] = xs + col_r X_r -- linear except for the S-box.  Collecting coefficients turns the rounds into
    //   D = SX x (one mat-vec up front),  s0_{r+1} = D_r + K_r + sum_{i<=r} G[r][i] X_i (scalar chain),
    //   state' = diag(1, post) [s0_22 ; x + CX X + ck] (one closing mat-vec over [x ; X]).
    const int n = W - 1, NB = n + RP + 1;   // basis size
    static u64 form[W - 1][W - 1 + RP + 1];
    memset(form, 0, sizeof(form));
    for (int i = 0; i < n; i++) { form[i][i] = 1; form[i][NB - 1] = cst[0 * W + 1 + i]; }
    for (int r = 0; r < RP; r++) {
        T.cst0[r] = cst[r * W];
        u64 dotf[W - 1 + RP + 1];
        for (int b = 0; b < NB; b++) {
            u64 a = 0;
            for (int i = 0; i < n; i++) a = addmod(a, mulmod(row[r * n + i], form[i][b]));
            dotf[b] = a;
        }
        for (int j = 0; j < n; j++) { T.sx0[1 + j][r] = dotf[j]; T.sx1[1 + j][r] = dotf[j] >> 52; }
        for (int i = 0; i < r; i++) T.G[r][i] = dotf[n + i];
        T.G[r][r] = e00[r];
        T.K[r] = dotf[NB - 1];
        for (int i = 0; i < n; i++) {
            form[i][n + r] = addmod(form[i][n + r], col[r * n + i]);
            if (r + 1 < RP) form[i][NB - 1] = addmod(form[i][NB - 1], cst[(r + 1) * W + 1 + i]);
        }
    }
    // closing map: words 1..23 = post * form
    for (int i = 0; i < n; i++)
        for (int b = 0; b < NB; b++) {
            u64 a = 0;
            for (int k = 0; k < n; k++) a = addmod(a, mulmod(post[i * n + k], form[k][b]));
            if (b < n) { T.fin0[1 + b][1 + i] = a; T.fin1[1 + b][1 + i] = a >> 52; }
            else if (b < n + RP) { T.fin0[W + (b - n)][1 + i] = a; T.fin1[W + (b - n)][1 + i] = a >> 52; }
            else { T.fk0[1 + i] = a & ((1ULL << 52) - 1); T.fk1[1 + i] = a >> 52; }
        }
}

void permute(u64 st[24]) {
    V x[3];
    for (int g = 0; g < 3; g++) x[g] = _mm512_loadu_si512((const void *)(st + 8 * g));
    for (int r = 0; r < RF / 2; r++) full_round(x, T.arkf[r]);
    // Partial rounds (tables: see build).  The only sequential part is the scalar chain of word 0: three dependent multiplies
    // of the S-box and one more per round; the contributions of X_r to the later rounds are pushed into their lazy
    // 192-bit accumulators off the critical path.
    alignas(64) u64 xl[NX], xh[NX], d[W];
    split_words(x, xl, xh);
    V dv[3];
    matvec_n(T.sx0, T.sx1, W, xl, xh, nullptr, nullptr, dv);
    for (int g = 0; g < 3; g++) _mm512_store_si512((void *)(d + 8 * g), dv[g]);
    u64 lo[RP], mid[RP], hi[RP];
    for (int r = 0; r < RP; r++) {
        u128 t = (u128)d[r] + T.K[r];
        lo[r] = (u64)t; mid[r] = (u64)(t >> 64); hi[r] = 0;
    }
    u64 s0 = xl[0];
    for (int r = 0; r < RP; r++) {
        u64 X = sbox_loose(add_loose(s0, T.cst0[r]));
        xl[W + r] = X; xh[W + r] = X >> 52;
        {   // this round's own term closes s0_{r+1}
            u128 pr = (u128)T.G[r][r] * X;
            u128 t = (u128)lo[r] + (u64)pr;
            u128 t2 = (u128)mid[r] + (u64)(pr >> 64) + (u64)(t >> 64);
            u64 h = hi[r] + (u64)(t2 >> 64);
            u64 v = reduce128_loose((u64)t, (u64)t2), sh = h << 32;   // minus h * 2^32: 2^128 = -2^32 (mod p)
            s0 = v - sh - (EPS & (0 - (u64)(v < sh)));                 // borrow: the wrap added 2^64 = eps
        }
        for (int q = r + 1; q < RP; q++) {
            u128 pr = (u128)T.G[q][r] * X;
            u128 t = (u128)lo[q] + (u64)pr;
            lo[q] = (u64)t;
            t = (u128)mid[q] + (u64)(pr >> 64) + (u64)(t >> 64);
            mid[q] = (u64)t;
            hi[q] += (u64)(t >> 64);
        }
    }
    V seed0[3], seed52[3];
    for (int g = 0; g < 3; g++) {
        seed0[g] = _mm512_load_si512((const void *)(T.fk0 + 8 * g));
        seed52[g] = _mm512_load_si512((const void *)(T.fk1 + 8 * g));
    }
    matvec_n(T.fin0, T.fin1, NX, xl, xh, seed0, seed52, x);      // lane 0 of every column is zero
    x[0] = _mm512_mask_set1_epi64(x[0], 0x01, (long long)canon(s0));
    for (int r = RF / 2; r < RF; r++) full_round(x, T.arkf[r]);
    for (int g = 0; g < 3; g++) _mm512_storeu_si512((void *)(st + 8 * g), x[g]);
}

}  // namespace psimd
}  // namespace lf

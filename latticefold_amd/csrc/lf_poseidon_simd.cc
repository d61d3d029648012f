// lf_poseidon_simd.cc -- see lf_poseidon_simd.h.  Plain host C++ (compiled with the AVX-512 IFMA target for this file only;
// every entry point is reached through the run-time check psimd::supported()).
//
// Arithmetic.  vpmadd52{l,h}uq multiply the low 52 bits of their operands.  A product a * b with a = a0 + 2^52 a1 (a1 < 2^13,
// the un-split word serves as a0) and b = bl + 2^32 bh (32-bit halves) is
//      a * bl = lo52(a0 bl) + 2^52 (hi52(a0 bl) + a1 bl)          (a1 bl < 2^46: its low part is the whole product)
// and the same for bh at weight 2^32: SIX IFMAs, four weight classes
//      value = A0 + 2^52 A52 + 2^32 (B0 + 2^52 B52),
// and sums of up to 46 products (+ a seed) stay below 2^58 (A0, B0) / 2^51 (A52, B52): no carry handling, one reduction per
// output word.  Measured on the EPYC 9575F of the GPU box (profiles/r05_host_poseidon.txt): IFMA issues 2 per cycle, and the
// permutation was bound by the LATENCY of what surrounds the products -- compare -> mask -> masked-add carry chains in the
// reduction (55 cycles per dependent product in the S-box layers) and the scalar word-0 chain of the partial rounds.  Hence:
//  * a field element lives as a PAIR (u, v), value = u + 2^32 v, u < 2^33, v < 2^32 + 137 (not canonical, not even below 2^64).
//    The reduction (reduce_uv) cuts the four classes into 32-bit chunks c_k of weight 2^(32k), folds them with
//    2^64 = 2^32 - 1, 2^96 = -1 (mod p) into U + 2^32 W, adds a multiple of p that makes both positive, and
//    propagates two small carries with shifts: 27 one-cycle operations, dependency depth 11, no compares.  (u, v) is exactly
//    the multiplier form (bl, bh) of the next product; the multiplicand form costs five more operations (prep_a);
//  * the partial rounds: scalar chain of word 0 with ONE product and one reduction per round next to the S-box; everything
//    else of round r+1's input (D, constants, sum_{i<r} G X_i) is prepared off the chain -- the triangle of cross terms is
//    accumulated by the vector unit (E, in memory, one column per round together with the closing map's column F) and read
//    back one lane per round.
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
    alignas(64) u64 ark0[RF][W], ark1[RF][W];   // ... as the seed of the mat-vec before them: low 52 bits / top 12 bits
    // partial rounds in scalar form (see permute): D = SX x, s0_{r+1} = D_r + K_r + sum_{i<=r} G[r][i] X_i,
    // closing map  state' = FIN [x ; X] + FK
    alignas(64) u64 sx0[W][W], sx1[W][W];       // [j][r] = coefficient of state word j in D_r (column 0 and lanes >= 22 zero)
    alignas(64) u64 fin0[NX][W], fin1[NX][W];   // [j][i]: columns 0..23 state words, 24..45 the S-box outputs X_r; lane 0 zero
    alignas(64) u64 sxm0[W][W], sxm1[W][W];     // SX M (rows 0..21) and row 0 of M in lane 22: D and word 0 straight from the S-box outputs of the last full round
    alignas(64) u64 finm0[W][W], finm1[W][W];   // (state columns of the closing map) M
    alignas(64) u64 fk0[W], fk1[W];             // constant of the closing map (52-bit / top-12-bit halves), + the constants of the full round behind it
    alignas(64) u64 e0[RP][W], e1[RP][W];       // [r][q] = G[q][r] for q >= r + 2 (the cross terms the vector unit accumulates), else 0
    u64 cst0[RP], K[RP], G[RP][RP];
    u64 Kc[RP];                                 // K_q + cst0[q + 1] (the next round's constant of word 0 rides along)
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
inline u64 sbox_loose(u64 x) {
    u64 x2 = mul_loose(x, x), x3 = mul_loose(x2, x), x4 = mul_loose(x2, x2);
    return mul_loose(x4, x3);
}

struct UV { V u, v; };   // value = u + 2^32 v  (u < 2^33, v < 2^32 + 137)

inline V vsrl(V a, int n) { return _mm512_srli_epi64(a, n); }
inline V vsll(V a, int n) { return _mm512_slli_epi64(a, n); }
inline V vadd64(V a, V b) { return _mm512_add_epi64(a, b); }
inline V vsub64(V a, V b) { return _mm512_sub_epi64(a, b); }
inline V vand(V a, V b) { return _mm512_and_si512(a, b); }

// A0 + 2^52 A52 + 2^32 (B0 + 2^52 B52)  (A0, B0 < 2^58, A52, B52 < 2^51)  ->  (u, v), u in [2^32 - 137, 2^33), v < 2^32 + 137.
// 27 operations, dependency depth 11.  Chunks of weight 2^0, 2^32, 2^64, 2^96 (the upper ones are left up to 39 bits wide):
//   c0 = A0 mod 2^32,  c1 = (A0 >> 32) + 2^20 (A52 mod 2^12) + (B0 mod 2^32),  c2 = (A52 >> 12) + (B0 >> 32) + 2^20 (B52 mod 2^12),  c3 = B52 >> 12
// and c0 + 2^32 c1 + 2^64 c2 + 2^96 c3 = (c0 - c2 - c3) + 2^32 (c1 + c2)  (mod p).  Adding p = (2^41 + 1) + 2^32 (2^32 - 513) makes both
// parts positive; U = c0 - c2 - c3 + 2^41 + 1 < 2^42 passes its upper bits to W, W's upper bits wh <= 137 come back through
// 2^64 wh = 2^32 wh - wh: - wh to u (which borrows 2^32 from v so that it cannot go negative), + wh to v.
inline UV reduce_uv(V A0, V A52, V B0, V B52) {
    const V eps = _mm512_set1_epi64((long long)EPS), m12 = _mm512_set1_epi64(0xfff);
    const V biasU = _mm512_set1_epi64((1LL << 41) + 1), biasW = _mm512_set1_epi64((1LL << 32) - 513);
    const V two32 = _mm512_set1_epi64(1LL << 32), one = _mm512_set1_epi64(1);
    V c1 = vadd64(vadd64(vsrl(A0, 32), vsll(vand(A52, m12), 20)), vand(B0, eps));
    V c2 = vadd64(vadd64(vsrl(A52, 12), vsrl(B0, 32)), vsll(vand(B52, m12), 20));
    V c3 = vsrl(B52, 12);
    V U = vsub64(vsub64(vadd64(vand(A0, eps), biasU), c2), c3);    // (2^41 - 2^40.01, 2^41 + 2^32 + 1)
    V W1 = vadd64(vadd64(vadd64(c1, biasW), c2), vsrl(U, 32));     // [2^32 - 259, 2^39.1)
    V wh = vsrl(W1, 32);                                            // 0 .. 137 (0 only with W1 >= 2^32 - 259)
    UV r;
    r.u = vsub64(_mm512_ternarylogic_epi64(U, eps, two32, 0xEA), wh);   // (U mod 2^32) + 2^32 - wh
    r.v = vadd64(vand(W1, eps), vsub64(wh, one));                       // (W1 mod 2^32) + wh - 1
    return r;
}
// multiplicand form of (u, v): the word whose low 52 bits are the value's, and the value's bits from 52 up
inline void prep_a(const UV &x, V &wa, V &a1) {
    wa = vadd64(x.u, vsll(x.v, 32));
    a1 = vsrl(vadd64(x.v, vsrl(x.u, 32)), 20);
}
inline UV vmul(V wa, V a1, const UV &b) {
    const V z = _mm512_setzero_si512();
    V a0 = _mm512_madd52lo_epu64(z, wa, b.u);
    V a52 = _mm512_madd52lo_epu64(_mm512_madd52hi_epu64(z, wa, b.u), a1, b.u);
    V b0 = _mm512_madd52lo_epu64(z, wa, b.v);
    V b52 = _mm512_madd52lo_epu64(_mm512_madd52hi_epu64(z, wa, b.v), a1, b.v);
    return reduce_uv(a0, a52, b0, b52);
}
inline V vadd(V a, V b) {   // canonical + canonical -> canonical
    const V eps = _mm512_set1_epi64((long long)EPS), pp = _mm512_set1_epi64((long long)P);
    V r = _mm512_add_epi64(a, b);
    __mmask8 c = _mm512_cmplt_epu64_mask(r, a);
    __mmask8 g = _mm512_cmpge_epu64_mask(r, pp);
    r = _mm512_mask_add_epi64(r, c, r, eps);                // wrapped: + 2^64 - p
    return _mm512_mask_sub_epi64(r, (__mmask8)(g & ~c), r, pp);
}
// (u, v) -> canonical word
inline V to_canon(const UV &x) {
    const V eps = _mm512_set1_epi64((long long)EPS), pp = _mm512_set1_epi64((long long)P);
    V vv = vadd64(x.v, vsrl(x.u, 32));                      // value = (u mod 2^32) + 2^32 vv, vv < 2^32 + 9
    V lo = vadd64(vand(x.u, eps), vsll(vv, 32));            // mod 2^64; the part above is (vv >> 32) 2^64 = (vv >> 32) eps
    V t = _mm512_mullo_epi64(vsrl(vv, 32), eps);            // 0 or eps (vv >> 32 is 0 or 1: one multiply is cheaper to write than a mask here)
    V r = vadd64(lo, t);
    __mmask8 c = _mm512_cmplt_epu64_mask(r, t);
    r = _mm512_mask_add_epi64(r, c, r, eps);
    __mmask8 g = _mm512_cmpge_epu64_mask(r, pp);
    return _mm512_mask_sub_epi64(r, g, r, pp);
}

// lazy accumulators of a mat-vec with 24 output words: six classes x three registers
struct Acc {
    V a0[3], a52[3], a52b[3], b0[3], b52[3], b52b[3];
    inline void init(const u64 *seed0, const u64 *seed52) {   // seed = seed0 + 2^52 seed52 (both given per output word), or none
        const V z = _mm512_setzero_si512();
        for (int g = 0; g < 3; g++) {
            a0[g] = seed0 ? _mm512_load_si512((const void *)(seed0 + 8 * g)) : z;
            a52[g] = seed52 ? _mm512_load_si512((const void *)(seed52 + 8 * g)) : z;
            a52b[g] = b0[g] = b52[g] = b52b[g] = z;
        }
    }
    // += column * x; col0 = the column's words (their low 52 bits are used), col1 = their top 12 bits; (xu, xv) = x
    inline void col(const u64 *col0, const u64 *col1, u64 xu, u64 xv) {
        V bl = _mm512_set1_epi64((long long)xu), bh = _mm512_set1_epi64((long long)xv);
#pragma GCC unroll 3
        for (int g = 0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(col0 + 8 * g)), m1 = _mm512_load_si512((const void *)(col1 + 8 * g));
            a0[g] = _mm512_madd52lo_epu64(a0[g], m, bl);
            a52[g] = _mm512_madd52hi_epu64(a52[g], m, bl);
            a52b[g] = _mm512_madd52lo_epu64(a52b[g], m1, bl);
            b0[g] = _mm512_madd52lo_epu64(b0[g], m, bh);
            b52[g] = _mm512_madd52hi_epu64(b52[g], m, bh);
            b52b[g] = _mm512_madd52lo_epu64(b52b[g], m1, bh);
        }
    }
    inline void finish(UV out[3]) const {
        for (int g = 0; g < 3; g++) out[g] = reduce_uv(a0[g], vadd64(a52[g], a52b[g]), b0[g], vadd64(b52[g], b52b[g]));
    }
};
// the same in memory, four classes (the two parts of the 2^52 classes share an accumulator): the cross terms of the partial rounds
struct AccMem {
    alignas(64) u64 a0[W], a52[W], b0[W], b52[W];
    inline void clear() { memset(this, 0, sizeof(*this)); }
    inline void col(const u64 *col0, const u64 *col1, u64 xu, u64 xv, int g0) {
        V bl = _mm512_set1_epi64((long long)xu), bh = _mm512_set1_epi64((long long)xv);
        for (int g = g0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(col0 + 8 * g)), m1 = _mm512_load_si512((const void *)(col1 + 8 * g));
            V x0 = _mm512_load_si512((const void *)(a0 + 8 * g)), x52 = _mm512_load_si512((const void *)(a52 + 8 * g));
            V y0 = _mm512_load_si512((const void *)(b0 + 8 * g)), y52 = _mm512_load_si512((const void *)(b52 + 8 * g));
            x0 = _mm512_madd52lo_epu64(x0, m, bl);
            x52 = _mm512_madd52lo_epu64(_mm512_madd52hi_epu64(x52, m, bl), m1, bl);
            y0 = _mm512_madd52lo_epu64(y0, m, bh);
            y52 = _mm512_madd52lo_epu64(_mm512_madd52hi_epu64(y52, m, bh), m1, bh);
            _mm512_store_si512((void *)(a0 + 8 * g), x0);
            _mm512_store_si512((void *)(a52 + 8 * g), x52);
            _mm512_store_si512((void *)(b0 + 8 * g), y0);
            _mm512_store_si512((void *)(b52 + 8 * g), y52);
        }
    }
};
inline void store_uv(const UV x[3], u64 *xu, u64 *xv) {
    for (int g = 0; g < 3; g++) {
        _mm512_store_si512((void *)(xu + 8 * g), x[g].u);
        _mm512_store_si512((void *)(xv + 8 * g), x[g].v);
    }
}
// x <- M x (+ seed) for a 24 x 24 matrix
inline void matvec(const u64 (*t0)[W], const u64 (*t1)[W], UV x[3], const u64 *seed0, const u64 *seed52) {
    alignas(64) u64 xu[W], xv[W];
    store_uv(x, xu, xv);
    Acc A;
    A.init(seed0, seed52);
    for (int j = 0; j < W; j++) A.col(t0[j], t1[j], xu[j], xv[j]);
    A.finish(x);
}

// S-box layer and MDS of a full round; the round constants were added by the producer of x (as the seed of its mat-vec), the
// constants of the NEXT full round are this mat-vec's seed
inline void sbox_layer(UV x[3]) {
    UV x2[3], x3[3], x4[3];
    V wa[3], a1[3];
    for (int g = 0; g < 3; g++) prep_a(x[g], wa[g], a1[g]);
    for (int g = 0; g < 3; g++) x2[g] = vmul(wa[g], a1[g], x[g]);
    for (int g = 0; g < 3; g++) prep_a(x2[g], wa[g], a1[g]);
    for (int g = 0; g < 3; g++) x3[g] = vmul(wa[g], a1[g], x[g]);
    for (int g = 0; g < 3; g++) x4[g] = vmul(wa[g], a1[g], x2[g]);
    for (int g = 0; g < 3; g++) prep_a(x4[g], wa[g], a1[g]);
    for (int g = 0; g < 3; g++) x[g] = vmul(wa[g], a1[g], x3[g]);
}
inline void full_round(UV x[3], const u64 *seed0, const u64 *seed52) {
    sbox_layer(x);
    matvec(T.mds0, T.mds1, x, seed0, seed52);
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
        for (int i = 0; i < W; i++) { T.ark0[r][i] = T.arkf[r][i] & ((1ULL << 52) - 1); T.ark1[r][i] = T.arkf[r][i] >> 52; }
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
            else {   // the closing map's constant plus the constants of the full round that follows it
                a = addmod(a, T.arkf[RF / 2][1 + i]);
                T.fk0[1 + i] = a & ((1ULL << 52) - 1); T.fk1[1 + i] = a >> 52;
            }
        }
    for (int r = 0; r < RP; r++) {
        T.Kc[r] = r + 1 < RP ? addmod(T.K[r], T.cst0[r + 1]) : T.K[r];
        for (int q = r + 2; q < RP; q++) { T.e0[r][q] = T.G[q][r]; T.e1[r][q] = T.G[q][r] >> 52; }
    }
    // The mat-vec of the full round in front of the partial rounds is folded into what consumes its output: x = M s, D = SX x = (SX M) s,
    // closing-map part = FIN_x x = (FIN_x M) s, word 0 = (row 0 of M) s (lane 22 of the D table): two mat-vecs over s instead of three
    for (int j = 0; j < W; j++) {
        for (int r = 0; r < W; r++) {
            u64 a = 0, b = 0;
            for (int i = 0; i < W; i++) {
                a = addmod(a, mulmod(T.sx0[i][r], mds[i * W + j]));
                b = addmod(b, mulmod(T.fin0[i][r], mds[i * W + j]));
            }
            if (r == RP) a = mds[0 * W + j];
            T.sxm0[j][r] = a; T.sxm1[j][r] = a >> 52;
            T.finm0[j][r] = b; T.finm1[j][r] = b >> 52;
        }
    }
}

void permute(u64 st[24]) {
    const V eps = _mm512_set1_epi64((long long)EPS);
    UV x[3];
    for (int g = 0; g < 3; g++) {
        V w = vadd(_mm512_loadu_si512((const void *)(st + 8 * g)), _mm512_load_si512((const void *)(T.arkf[0] + 8 * g)));
        x[g].u = vand(w, eps);
        x[g].v = vsrl(w, 32);
    }
    for (int r = 0; r + 1 < RF / 2; r++) full_round(x, T.ark0[r + 1], T.ark1[r + 1]);
    sbox_layer(x);      // the last full round of the first half: its mat-vec is folded into the tables below
    // Partial rounds (tables: see build):  X_r = sbox(s_r),  s_{r+1} = base_r + G[r][r] X_r  on the scalar chain, with
    //   base_r = D_r + K_r + cst0_{r+1} + sum_{i <= r-2} G[r][i] X_i + G[r][r-1] X_{r-1}
    // prepared while the S-box of round r runs: the sum over i <= r-2 is lane r of the vector accumulator E (one column per
    // round, issued a round before it is read), the last cross term one scalar product.  X_r's column of the closing map is
    // accumulated in the same place (F).
    alignas(64) u64 xu[W], xv[W], du[W], dv[W];
    store_uv(x, xu, xv);
    {
        Acc D;
        D.init(nullptr, nullptr);
        for (int j = 0; j < W; j++) D.col(T.sxm0[j], T.sxm1[j], xu[j], xv[j]);
        UV d[3];
        D.finish(d);
        store_uv(d, du, dv);
    }
    Acc F;
    F.init(T.fk0, T.fk1);
    for (int j = 0; j < W; j++) F.col(T.finm0[j], T.finm1[j], xu[j], xv[j]);  // lane 0 of every column is zero
    AccMem E;
    E.clear();
    auto loose = [](u128 t) { return reduce128_loose((u64)t, (u64)(t >> 64)); };
    u64 s = add_loose(loose((u128)du[RP] + ((u128)dv[RP] << 32)), T.cst0[0]);      // word 0 of the state: lane 22 of the D table
    u64 base = loose((u128)du[0] + ((u128)dv[0] << 32) + T.Kc[0]);
    for (int r = 0; r < RP; r++) {
        // next round's base without its X_r term: lane r + 1 of E is complete (its last term came from X_{r-1}, stored a round ago) -- read BEFORE this
        // round's column is added, so that the read does not wait for this round's stores; 2^84 B52 = 2^84 (B52 mod 2^12) - (B52 >> 12)
        u64 lp = 0;
        if (r + 1 < RP) {
            const int q = r + 1;
            u128 part = (u128)E.a0[q] + ((u128)E.a52[q] << 52) + ((u128)E.b0[q] << 32) + ((u128)(E.b52[q] & 0xfff) << 84) + ((u128)P << 40)
                        - (E.b52[q] >> 12) + du[q] + ((u128)dv[q] << 32) + T.Kc[q];
            lp = loose(part);
        }
        const u64 X = sbox_loose(s);
        s = loose((u128)T.G[r][r] * X + base);                      // (2^64 - 1)^2 + 2^64 - 1 < 2^128
        if (r + 1 < RP) base = loose((u128)T.G[r + 1][r] * X + lp);
        const u64 Xu = X & EPS, Xv = X >> 32;
        F.col(T.fin0[W + r], T.fin1[W + r], Xu, Xv);
        if (r + 2 < RP) E.col(T.e0[r], T.e1[r], Xu, Xv, (r + 2) >> 3);
    }
    F.finish(x);
    {   // word 0 of the closing map is the chain's last value; the constants of the next full round ride along
        u64 s0 = add_loose(s, T.arkf[RF / 2][0]);
        x[0].u = _mm512_mask_set1_epi64(x[0].u, 0x01, (long long)(s0 & EPS));
        x[0].v = _mm512_mask_set1_epi64(x[0].v, 0x01, (long long)(s0 >> 32));
    }
    for (int r = RF / 2; r < RF; r++) full_round(x, r + 1 < RF ? T.ark0[r + 1] : nullptr, r + 1 < RF ? T.ark1[r + 1] : nullptr);
    for (int g = 0; g < 3; g++) _mm512_storeu_si512((void *)(st + 8 * g), to_canon(x[g]));
}

}  // namespace psimd
}  // namespace lf

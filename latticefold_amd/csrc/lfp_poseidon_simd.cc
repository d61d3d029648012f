// lfp_poseidon_simd.cc -- see lfp_poseidon_simd.h.  Plain host C++ (compiled with the AVX-512 IFMA target for this file only; every entry
// point is reached through the run-time check lfp_psimd::supported()).
//
// The Frog prime p = 15912092521325583641 (~2^63.8) has no special form, so the lanes work on MONTGOMERY words with R = 2^104:
//   * a 64-bit word a = a0 + 2^52 a1 (a1 < 2^12); vpmadd52{l,h}uq multiply the low 52 bits of their operands, so the un-split word serves as a0.
//     A product a*b is  lo52(a0 b0)  +  2^52 [hi52(a0 b0) + lo52(a0 b1) + lo52(a1 b0)]  +  2^104 [hi52(a0 b1) + hi52(a1 b0) + lo52(a1 b1)]:
//     seven IFMAs, no carries; sums of up to 46 products stay below 2^60 per weight class;
//   * one reduction per OUTPUT word: two radix-2^52 Montgomery steps (q = -V p^-1 mod 2^52, V <- (V + q p) / 2^52) turn
//     V = W0 + 2^52 W52 + 2^104 W104 into V 2^-104 mod p below 2p -- 14 vector operations, no more than the special-form reduction of the
//     Goldilocks lanes (lf_poseidon_simd.cc).
//   * the mat-vecs split the MULTIPLIER into 32-bit halves instead (x = xl + 2^32 xh: a0 xl, a1 xl < 2^44 whole, the same at weight 2^32): six IFMAs
//     per product, four weight classes that are put back on the 52-bit grid once per output word (eight cheap operations).
// The 22 partial rounds are collapsed by linearity exactly as there: D = SX x (one mat-vec), the scalar chain over word 0
// (s0_{r+1} = D_r + K_r + sum_{i<=r} G[r][i] X_i; scalar Montgomery words with R = 2^64, the factors 2^+-40 between the two domains are folded
// into the SX / closing tables), a closing map over [x ; X].  The chain carries three dependent products per round (G x^7 = ((G x) x^2) x^4
// next to x^7) and one addition; the rest of s0_{r+1} is prepared while the S-box of round r runs: the cross terms sum_{i<=r-2} G[r][i] X_i
// and the closing map's columns are accumulated by the vector unit (E in memory, F in registers, one column per round), the last cross
// term is one scalar product, one 192-bit reduction per round off the chain.
#include "lfp_poseidon_simd.h"

#include <immintrin.h>
#include <string.h>

namespace lfp_psimd {

typedef uint64_t u64;
typedef unsigned __int128 u128;
typedef __m512i V;

namespace {
constexpr int W = 24, RF = 8, RP = 22, NX = W + RP;
constexpr u64 M52 = (1ULL << 52) - 1;

struct Tables {
    u64 p, pinv52, pinv64n, c64;                // the prime, -p^-1 mod 2^52, -p^-1 mod 2^64, 2^64 - p
    u64 r2_104;                                 // 2^208 mod p: vmul(x, r2_104) = x 2^104
    u64 two24, two104;                          // 2^24, 2^104 mod p (domain changes of word 0 through the scalar Montgomery product)
    alignas(64) u64 mds0[W][W], mds1[W][W];     // [j][i] = M[i][j] 2^104 and its top 12 bits
    alignas(64) u64 arkf[RF][W];                // constants of the full rounds, 2^104 form
    alignas(64) u64 sx0[W][W], sx1[W][W];       // [j][r]: coefficient of state word j in D_r, times 2^-40 (2^104 form): D comes out in 2^64 form
    alignas(64) u64 fin0[NX][W], fin1[NX][W];   // closing map: columns 0..23 state words (2^104 form), 24..45 the S-box outputs X_r (given in 2^64 form: times 2^40)
    alignas(64) u64 sxm0[W][W], sxm1[W][W];     // SX M (rows 0..21) and row 0 of M in lane 22, times 2^-40: D and word 0 (2^64 form) straight from the S-box outputs of the last full round
    alignas(64) u64 finm0[W][W], finm1[W][W];   // (state columns of the closing map) M
    alignas(64) u64 fk[W];                      // constant of the closing map (2^104 form), added after the reduction
    alignas(64) u64 e0[RP][W], e1[RP][W];       // [r][q] = G[q][r] for q >= r + 2 (the cross terms the vector unit accumulates, raw 2^64-form words), else 0
    u64 cst0[RP], K[RP], G[RP][RP];             // scalar chain, 2^64 form
    u64 Kc[RP];                                 // K_q + cst0[q + 1]
};
Tables T;

inline u64 addmod(u64 a, u64 b) {   // canonical operands (p > 2^63: the sum may wrap); branch-free
    u64 r;
    const u64 c = __builtin_add_overflow(a, b, &r);
    return r - (T.p & (0 - (c | (u64)(r >= T.p))));
}
inline u64 mulmod(u64 a, u64 b) { return (u64)(((u128)a * b) % T.p); }
inline u64 shl_mod(u64 a, int k) { while (k > 0) { int s = k > 60 ? 60 : k; a = (u64)((((u128)a) << s) % T.p); k -= s; } return a; }
u64 powmod(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; } return r; }

// scalar Montgomery (R = 2^64): a b 2^-64 mod p, canonical.  Branch-free: whether the final subtraction happens is a coin flip per product, and the chain
// of the partial rounds is seven of these per round (a mispredicted branch costs more than the product)
inline u64 mm(u64 a, u64 b) {
    const u128 t = (u128)a * b;
    const u64 m = (u64)t * T.pinv64n;
    const u128 mp = (u128)m * T.p;
    const u128 s = (u128)(u64)(t >> 64) + (u64)(mp >> 64) + ((u64)t != 0);      // (t + m p) / 2^64 < 2p
    const u64 r = (u64)s;
    const u64 use = (u64)(s >> 64) | (u64)(r >= T.p);
    return r - (T.p & (0 - use));
}
inline u64 sbox(u64 x) { const u64 x2 = mm(x, x), x3 = mm(x2, x), x4 = mm(x2, x2); return mm(x4, x3); }
// (lo + 2^64 mid + 2^128 hi) 2^-64 mod p, canonical; hi small (the carries of at most 24 terms)
inline u64 redc192(u64 lo, u64 mid, u64 hi) {
    const u64 m = lo * T.pinv64n;
    const u128 mp = (u128)m * T.p;
    const u128 s = (u128)mid + (u64)(mp >> 64) + (lo != 0);
    const u64 sh = hi + (u64)(s >> 64);
    u128 t = (u128)sh * T.c64 + (u64)s;                  // 2^64 = c64 (mod p)
    t = (u128)(u64)(t >> 64) * T.c64 + (u64)t;           // < 2^64 + 2^63.3
    u64 r = (u64)t + (T.c64 & (0 - (u64)(t >> 64)));     // (the low word is < 2^63.3 when the high one is set: no wrap); r < 2^64 < 2p + 2^62
    r -= T.p & (0 - (u64)(r >= T.p));
    r -= T.p & (0 - (u64)(r >= T.p));
    return r;
}

// (W0 + 2^52 W52 + 2^104 W104) 2^-104 mod p, canonical.  W0, W52 < 2^60, W104 < 2^32
inline V reduce(V w0, V w52, V w104) {
    const V z = _mm512_setzero_si512(), pp = _mm512_set1_epi64((long long)T.p), p1 = _mm512_set1_epi64((long long)(T.p >> 52)),
            pi = _mm512_set1_epi64((long long)T.pinv52);
    V q = _mm512_madd52lo_epu64(z, w0, pi);
    V t = _mm512_madd52lo_epu64(w0, q, pp);              // low 52 bits vanish
    V u0 = _mm512_add_epi64(w52, _mm512_srli_epi64(t, 52));
    u0 = _mm512_madd52hi_epu64(u0, q, pp);
    u0 = _mm512_madd52lo_epu64(u0, q, p1);
    V u1 = _mm512_madd52hi_epu64(w104, q, p1);
    q = _mm512_madd52lo_epu64(z, u0, pi);
    t = _mm512_madd52lo_epu64(u0, q, pp);
    V v0 = _mm512_add_epi64(u1, _mm512_srli_epi64(t, 52));
    v0 = _mm512_madd52hi_epu64(v0, q, pp);
    v0 = _mm512_madd52lo_epu64(v0, q, p1);
    V v1 = _mm512_madd52hi_epu64(z, q, p1);
    V r = _mm512_add_epi64(v0, _mm512_slli_epi64(v1, 52));   // < p + 2^33 < 2^64
    __mmask8 g = _mm512_cmpge_epu64_mask(r, pp);
    return _mm512_mask_sub_epi64(r, g, r, pp);
}
inline V vmul(V a, V b) {
    const V z = _mm512_setzero_si512();
    V a1 = _mm512_srli_epi64(a, 52), b1 = _mm512_srli_epi64(b, 52);
    V w0 = _mm512_madd52lo_epu64(z, a, b);
    V w52 = _mm512_madd52hi_epu64(z, a, b);
    w52 = _mm512_madd52lo_epu64(w52, a, b1);
    w52 = _mm512_madd52lo_epu64(w52, a1, b);
    V w104 = _mm512_madd52hi_epu64(z, a, b1);
    w104 = _mm512_madd52hi_epu64(w104, a1, b);
    w104 = _mm512_madd52lo_epu64(w104, a1, b1);
    return reduce(w0, w52, w104);
}
inline V vadd(V a, V b) {   // canonical + canonical -> canonical (p < 2^64 - 2^61: the sum may wrap)
    const V pp = _mm512_set1_epi64((long long)T.p);
    V r = _mm512_add_epi64(a, b);
    __mmask8 c = _mm512_cmplt_epu64_mask(r, a);
    __mmask8 g = _mm512_cmpge_epu64_mask(r, pp);
    return _mm512_mask_sub_epi64(r, (__mmask8)(c | g), r, pp);
}

// lazy accumulators of a mat-vec with 24 output words (multiplier split into 32-bit halves): value = A0 + 2^52 A52 + 2^32 (B0 + 2^52 B52)
struct Acc {
    V a0[3], a52[3], a52b[3], b0[3], b52[3], b52b[3];
    inline void init() {
        const V z = _mm512_setzero_si512();
        for (int g = 0; g < 3; g++) a0[g] = a52[g] = a52b[g] = b0[g] = b52[g] = b52b[g] = z;
    }
    // += column * x; col0 = the column's words (their low 52 bits are used), col1 = their top 12 bits; xl / xh = the 32-bit halves of x
    inline void col(const u64 *col0, const u64 *col1, u64 xl, u64 xh) {
        V bl = _mm512_set1_epi64((long long)xl), bh = _mm512_set1_epi64((long long)xh);
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
    // back on the 52-bit grid: 2^32 B0 = 2^32 (B0 mod 2^20) + 2^52 (B0 >> 20), 2^84 B52 = 2^52 2^32 (B52 mod 2^20) + 2^104 (B52 >> 20); then 2^-104 mod p
    inline void finish(V out[3]) const {
        const V m20 = _mm512_set1_epi64((1 << 20) - 1);
        for (int g = 0; g < 3; g++) {
            V A52 = _mm512_add_epi64(a52[g], a52b[g]), B52 = _mm512_add_epi64(b52[g], b52b[g]);
            V w0 = _mm512_add_epi64(a0[g], _mm512_slli_epi64(_mm512_and_si512(b0[g], m20), 32));                       // < 2^58
            V w52 = _mm512_add_epi64(_mm512_add_epi64(A52, _mm512_srli_epi64(b0[g], 20)), _mm512_slli_epi64(_mm512_and_si512(B52, m20), 32));   // < 2^53
            out[g] = reduce(w0, w52, _mm512_srli_epi64(B52, 20));
        }
    }
};
// the same in memory, four classes (the two parts of the 2^52 classes share an accumulator): the cross terms of the partial rounds
struct AccMem {
    alignas(64) u64 a0[W], a52[W], b0[W], b52[W];
    inline void clear() { memset(this, 0, sizeof(*this)); }
    inline void col(const u64 *col0, const u64 *col1, u64 xl, u64 xh, int g0) {
        V bl = _mm512_set1_epi64((long long)xl), bh = _mm512_set1_epi64((long long)xh);
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
// the 32-bit halves of the state words, for the broadcasts of the next mat-vec
inline void split_words(const V x[3], u64 *xl, u64 *xh) {
    const V eps = _mm512_set1_epi64(0xffffffffLL);
    for (int g = 0; g < 3; g++) {
        _mm512_store_si512((void *)(xl + 8 * g), _mm512_and_si512(x[g], eps));
        _mm512_store_si512((void *)(xh + 8 * g), _mm512_srli_epi64(x[g], 32));
    }
}
inline void sbox_layer(V x[3], const u64 *ark) {
    V t[3], x2[3], x3[3], x4[3];
    for (int g = 0; g < 3; g++) t[g] = vadd(x[g], _mm512_load_si512((const void *)(ark + 8 * g)));
    for (int g = 0; g < 3; g++) x2[g] = vmul(t[g], t[g]);
    for (int g = 0; g < 3; g++) x3[g] = vmul(x2[g], t[g]);
    for (int g = 0; g < 3; g++) x4[g] = vmul(x2[g], x2[g]);
    for (int g = 0; g < 3; g++) x[g] = vmul(x4[g], x3[g]);
}
inline void full_round(V x[3], const u64 *ark) {
    sbox_layer(x, ark);
    alignas(64) u64 xl[W], xh[W];
    split_words(x, xl, xh);
    Acc A;
    A.init();
    for (int j = 0; j < W; j++) A.col(T.mds0[j], T.mds1[j], xl[j], xh[j]);
    A.finish(x);
}
}  // namespace

bool supported() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq");
    return ok;
}

void build(u64 p, const u64 *ark, const u64 *mds, const u64 *cst, const u64 *e00, const u64 *row, const u64 *col, const u64 *post) {
    memset(&T, 0, sizeof(T));
    T.p = p;
    u64 x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - p * x;          // p^-1 mod 2^64
    T.pinv64n = 0 - x;
    T.pinv52 = (0 - x) & M52;
    T.c64 = 0 - p;
    T.r2_104 = shl_mod(1, 208);
    T.two24 = shl_mod(1, 24);
    T.two104 = shl_mod(1, 104);
    const u64 inv2 = (p + 1) / 2, i40 = powmod(inv2, 40);                      // 2^-40
    auto to104 = [&](u64 a) { return shl_mod(a % p, 104); };
    auto to64 = [&](u64 a) { return shl_mod(a % p, 64); };
    for (int i = 0; i < W; i++)
        for (int j = 0; j < W; j++) {
            const u64 m = to104(mds[i * W + j]);
            T.mds0[j][i] = m;
            T.mds1[j][i] = m >> 52;
        }
    for (int r = 0; r < RF; r++) {
        const int src = r < RF / 2 ? r : RP + r;
        for (int i = 0; i < W; i++) T.arkf[r][i] = to104(ark[(size_t)src * W + i]);
    }
    // Symbolic run of the 22 sparse partial rounds (as lf_poseidon_simd.cc): every state word 1..23 is an affine form over
    //   [ x_1..x_23 (words on entry) | X_0..X_21 (S-box outputs of word 0) | 1 ]
    const int n = W - 1, NB = n + RP + 1;
    static u64 form[W - 1][W - 1 + RP + 1], sxc[W][W], finc[W][W];
    memset(form, 0, sizeof(form));
    memset(sxc, 0, sizeof(sxc));
    memset(finc, 0, sizeof(finc));
    for (int i = 0; i < n; i++) { form[i][i] = 1; form[i][NB - 1] = cst[0 * W + 1 + i] % p; }
    for (int r = 0; r < RP; r++) {
        T.cst0[r] = to64(cst[r * W]);
        u64 dotf[W - 1 + RP + 1];
        for (int b = 0; b < NB; b++) {
            u64 a = 0;
            for (int i = 0; i < n; i++) a = addmod(a, mulmod(row[r * n + i] % p, form[i][b]));
            dotf[b] = a;
        }
        for (int j = 0; j < n; j++) {
            const u64 v = to104(mulmod(dotf[j], i40));
            sxc[1 + j][r] = dotf[j];
            T.sx0[1 + j][r] = v;
            T.sx1[1 + j][r] = v >> 52;
        }
        for (int i = 0; i < r; i++) T.G[r][i] = to64(dotf[n + i]);
        T.G[r][r] = to64(e00[r]);
        T.K[r] = to64(dotf[NB - 1]);
        for (int i = 0; i < n; i++) {
            form[i][n + r] = addmod(form[i][n + r], col[r * n + i] % p);
            if (r + 1 < RP) form[i][NB - 1] = addmod(form[i][NB - 1], cst[(r + 1) * W + 1 + i] % p);
        }
    }
    for (int i = 0; i < n; i++)
        for (int b = 0; b < NB; b++) {
            u64 a = 0;
            for (int k = 0; k < n; k++) a = addmod(a, mulmod(post[i * n + k] % p, form[k][b]));
            if (b < n) { const u64 v = to104(a); T.fin0[1 + b][1 + i] = v; T.fin1[1 + b][1 + i] = v >> 52; finc[1 + b][1 + i] = a; }
            else if (b < n + RP) { const u64 v = shl_mod(a, 144); T.fin0[W + (b - n)][1 + i] = v; T.fin1[W + (b - n)][1 + i] = v >> 52; }   // x 2^104 x 2^40
            else T.fk[1 + i] = to104(a);
        }
    for (int r = 0; r < RP; r++) {
        T.Kc[r] = r + 1 < RP ? addmod(T.K[r], T.cst0[r + 1]) : T.K[r];
        for (int q = r + 2; q < RP; q++) { T.e0[r][q] = T.G[q][r]; T.e1[r][q] = T.G[q][r] >> 52; }
    }
    // The mat-vec of the full round in front of the partial rounds is folded into what consumes its output: x = M s, D = (SX M) s, closing-map part
    // (FIN_x M) s, word 0 = (row 0 of M) s in lane 22 of the D table (so it arrives in 2^64 form like D): two mat-vecs over s instead of three
    for (int j = 0; j < W; j++)
        for (int r = 0; r < W; r++) {
            u64 a = 0, b = 0;
            for (int i = 0; i < W; i++) {
                a = addmod(a, mulmod(sxc[i][r], mds[i * W + j] % p));
                b = addmod(b, mulmod(finc[i][r], mds[i * W + j] % p));
            }
            if (r == RP) a = mds[0 * W + j] % p;
            const u64 va = to104(mulmod(a, i40)), vb = to104(b);
            T.sxm0[j][r] = va; T.sxm1[j][r] = va >> 52;
            T.finm0[j][r] = vb; T.finm1[j][r] = vb >> 52;
        }
}

void permute(u64 st[24]) {
    V x[3];
    const V r2 = _mm512_set1_epi64((long long)T.r2_104);
    for (int g = 0; g < 3; g++) x[g] = vmul(_mm512_loadu_si512((const void *)(st + 8 * g)), r2);      // -> 2^104 form
    for (int r = 0; r + 1 < RF / 2; r++) full_round(x, T.arkf[r]);
    sbox_layer(x, T.arkf[RF / 2 - 1]);      // the last full round of the first half: its mat-vec is folded into the tables below
    // Partial rounds (tables: see build), everything on the chain in 2^64 form:  X_r = sbox(s_r),  s_{r+1} = base_r + G[r][r] X_r  with
    //   base_r = D_r + K_r + cst0_{r+1} + sum_{i <= r-2} G[r][i] X_i + G[r][r-1] X_{r-1}
    // prepared while the S-box of round r runs: the sum over i <= r-2 is lane r of the vector accumulator E (raw products, one column per
    // round, issued a round before it is read), the last cross term one scalar product, one 192-bit Montgomery reduction for all of it.
    alignas(64) u64 xl[W], xh[W], d[W];
    split_words(x, xl, xh);
    {
        Acc D;
        D.init();
        for (int j = 0; j < W; j++) D.col(T.sxm0[j], T.sxm1[j], xl[j], xh[j]);      // D_r (and word 0 in lane 22) in 2^64 form (the tables carry 2^-40)
        V dv[3];
        D.finish(dv);
        for (int g = 0; g < 3; g++) _mm512_store_si512((void *)(d + 8 * g), dv[g]);
    }
    Acc F;
    F.init();
    for (int j = 0; j < W; j++) F.col(T.finm0[j], T.finm1[j], xl[j], xh[j]);        // lane 0 of every column is zero
    AccMem E;
    E.clear();
    u64 s = addmod(d[RP], T.cst0[0]);                                                // word 0 (lane 22 of the D table, 2^64 form) + its first constant
    u64 base = addmod(d[0], T.Kc[0]);
    for (int r = 0; r < RP; r++) {
        // next round's base without its X_r term: lane r + 1 of E is complete (its last term came from X_{r-1}, stored a round ago) -- read before this
        // round's column is added.  E = a0 + 2^52 a52 + 2^32 b0 + 2^84 b52 as (lo, mid:hi); (D + Kc) enters at weight 2^64 (the reduction divides by 2^64)
        u64 plo = 0;
        u128 pmh = 0;
        if (r + 1 < RP) {
            const int q = r + 1;
            const u128 t1 = (u128)E.a0[q] + ((u128)E.a52[q] << 52) + ((u128)E.b0[q] << 32);
            plo = (u64)t1;
            pmh = (u128)(u64)(t1 >> 64) + ((u128)E.b52[q] << 20) + d[q] + T.Kc[q];
        }
        // three dependent products on the chain: G x^7 = ((G x) x^2) x^4 next to x^7 = x^3 x^4 (which only the columns need)
        const u64 x2 = mm(s, s), gx = mm(T.G[r][r], s);
        const u64 x3 = mm(x2, s), x4 = mm(x2, x2), gx3 = mm(gx, x2);
        s = addmod(mm(gx3, x4), base);
        const u64 X = mm(x3, x4);
        if (r + 1 < RP) {
            const u128 pr = (u128)T.G[r + 1][r] * X;
            const u128 t = (u128)plo + (u64)pr;
            const u128 t2 = pmh + (u64)(pr >> 64) + (u64)(t >> 64);
            base = redc192((u64)t, (u64)t2, (u64)(t2 >> 64));
        }
        const u64 Xl = X & 0xffffffffULL, Xh = X >> 32;
        F.col(T.fin0[W + r], T.fin1[W + r], Xl, Xh);
        if (r + 2 < RP) E.col(T.e0[r], T.e1[r], Xl, Xh, (r + 2) >> 3);
    }
    F.finish(x);
    for (int g = 0; g < 3; g++) x[g] = vadd(x[g], _mm512_load_si512((const void *)(T.fk + 8 * g)));
    x[0] = _mm512_mask_set1_epi64(x[0], 0x01, (long long)mm(s, T.two104));           // word 0 back to 2^104 form
    for (int r = RF / 2; r < RF; r++) full_round(x, T.arkf[r]);
    const V one = _mm512_set1_epi64(1);
    for (int g = 0; g < 3; g++) _mm512_storeu_si512((void *)(st + 8 * g), vmul(x[g], one));            // -> canonical words
}

}  // namespace lfp_psimd

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
// The 22 partial rounds are collapsed by linearity exactly as there: D = SX x (one mat-vec), the scalar chain over word 0
// (s0_{r+1} = D_r + K_r + sum_{i<=r} G[r][i] X_i with lazy 192-bit sums for the cross terms; scalar Montgomery words with R = 2^64, the
// factors 2^+-40 between the two domains are folded into the SX / closing tables), one closing mat-vec over [x ; X].
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
    alignas(64) u64 fk[W];                      // constant of the closing map (2^104 form), added after the reduction
    u64 cst0[RP], K[RP], G[RP][RP];             // scalar chain, 2^64 form
};
Tables T;

inline u64 addmod(u64 a, u64 b) { u128 s = (u128)a + b; return (u64)(s >= T.p ? s - T.p : s); }
inline u64 mulmod(u64 a, u64 b) { return (u64)(((u128)a * b) % T.p); }
inline u64 shl_mod(u64 a, int k) { while (k > 0) { int s = k > 60 ? 60 : k; a = (u64)((((u128)a) << s) % T.p); k -= s; } return a; }
u64 powmod(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = mulmod(r, a); a = mulmod(a, a); e >>= 1; } return r; }

// scalar Montgomery (R = 2^64): a b 2^-64 mod p, canonical
inline u64 mm(u64 a, u64 b) {
    const u128 t = (u128)a * b;
    const u64 m = (u64)t * T.pinv64n;
    const u128 mp = (u128)m * T.p;
    const u128 s = (u128)(u64)(t >> 64) + (u64)(mp >> 64) + ((u64)t != 0);      // (t + m p) / 2^64 < 2p
    const u64 r = (u64)s;
    return (s >> 64) || r >= T.p ? r - T.p : r;
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
    u64 r = (u64)t;
    if ((u64)(t >> 64)) r += T.c64;                      // (r < 2^63.3 here: no wrap)
    while (r >= T.p) r -= T.p;
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

// out = (sum_j col_j * x_j) 2^-104 for ncols columns given column-wise (t0[j] = column j, t1[j] = its top 12 bits); xl / xh = the words x_j and their top 12 bits
inline void matvec_n(const u64 (*t0)[W], const u64 (*t1)[W], int ncols, const u64 *xl, const u64 *xh, V out[3]) {
    const V z = _mm512_setzero_si512();
    V a0[3], a52[3], a52b[3], a52c[3], a104[3], a104b[3], a104c[3];
    for (int g = 0; g < 3; g++) a0[g] = a52[g] = a52b[g] = a52c[g] = a104[g] = a104b[g] = a104c[g] = z;
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
inline void full_round(V x[3], const u64 *ark) {
    V t[3], x2[3], x3[3], x4[3];
    for (int g = 0; g < 3; g++) t[g] = vadd(x[g], _mm512_load_si512((const void *)(ark + 8 * g)));
    for (int g = 0; g < 3; g++) x2[g] = vmul(t[g], t[g]);
    for (int g = 0; g < 3; g++) x3[g] = vmul(x2[g], t[g]);
    for (int g = 0; g < 3; g++) x4[g] = vmul(x2[g], x2[g]);
    for (int g = 0; g < 3; g++) x[g] = vmul(x4[g], x3[g]);
    alignas(64) u64 xl[W], xh[W];
    split_words(x, xl, xh);
    matvec_n(T.mds0, T.mds1, W, xl, xh, x);
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
    static u64 form[W - 1][W - 1 + RP + 1];
    memset(form, 0, sizeof(form));
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
            if (b < n) { const u64 v = to104(a); T.fin0[1 + b][1 + i] = v; T.fin1[1 + b][1 + i] = v >> 52; }
            else if (b < n + RP) { const u64 v = shl_mod(a, 144); T.fin0[W + (b - n)][1 + i] = v; T.fin1[W + (b - n)][1 + i] = v >> 52; }   // x 2^104 x 2^40
            else T.fk[1 + i] = to104(a);
        }
}

void permute(u64 st[24]) {
    V x[3];
    const V r2 = _mm512_set1_epi64((long long)T.r2_104);
    for (int g = 0; g < 3; g++) x[g] = vmul(_mm512_loadu_si512((const void *)(st + 8 * g)), r2);      // -> 2^104 form
    for (int r = 0; r < RF / 2; r++) full_round(x, T.arkf[r]);
    alignas(64) u64 xl[NX], xh[NX], d[W];
    split_words(x, xl, xh);
    V dv[3];
    matvec_n(T.sx0, T.sx1, W, xl, xh, dv);                      // D_r in 2^64 form (the tables carry 2^-40)
    for (int g = 0; g < 3; g++) _mm512_store_si512((void *)(d + 8 * g), dv[g]);
    u64 lo[RP], mid[RP], hi[RP];
    for (int r = 0; r < RP; r++) {                               // (D_r + K_r) at weight 2^64: the reduction divides by 2^64
        const u128 t = (u128)d[r] + T.K[r];
        lo[r] = 0; mid[r] = (u64)t; hi[r] = (u64)(t >> 64);
    }
    u64 s0 = mm(xl[0], T.two24);                                 // word 0: 2^104 form -> 2^64 form
    for (int r = 0; r < RP; r++) {
        const u64 X = sbox(addmod(s0, T.cst0[r]));
        xl[W + r] = X; xh[W + r] = X >> 52;
        {   // this round's own term closes s0_{r+1}
            const u128 pr = (u128)T.G[r][r] * X;
            const u128 t = (u128)lo[r] + (u64)pr;
            const u128 t2 = (u128)mid[r] + (u64)(pr >> 64) + (u64)(t >> 64);
            s0 = redc192((u64)t, (u64)t2, hi[r] + (u64)(t2 >> 64));
        }
        for (int q = r + 1; q < RP; q++) {
            const u128 pr = (u128)T.G[q][r] * X;
            u128 t = (u128)lo[q] + (u64)pr;
            lo[q] = (u64)t;
            t = (u128)mid[q] + (u64)(pr >> 64) + (u64)(t >> 64);
            mid[q] = (u64)t;
            hi[q] += (u64)(t >> 64);
        }
    }
    matvec_n(T.fin0, T.fin1, NX, xl, xh, x);                     // lane 0 of every column is zero
    for (int g = 0; g < 3; g++) x[g] = vadd(x[g], _mm512_load_si512((const void *)(T.fk + 8 * g)));
    x[0] = _mm512_mask_set1_epi64(x[0], 0x01, (long long)mm(s0, T.two104));     // word 0 back to 2^104 form
    for (int r = RF / 2; r < RF; r++) full_round(x, T.arkf[r]);
    const V one = _mm512_set1_epi64(1);
    for (int g = 0; g < 3; g++) _mm512_storeu_si512((void *)(st + 8 * g), vmul(x[g], one));            // -> canonical words
}

}  // namespace lfp_psimd

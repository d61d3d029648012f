// bb_poseidon_avx512.cc -- AVX-512 (IFMA) lanes for the BabyBear Poseidon permutation (width 24, 8 full + 22 partial rounds,
// alpha 7).  A BabyBear fold step needs ~5300 permutations (a ring element is 72 words), all on the host; this path is selected
// at run time when the CPU has avx512f/ifma/dq (bb_poseidon_simd.h, AVX2, otherwise; LF_POSEIDON_SCALAR=1 forces the scalar code).
//
// State: three zmm registers of eight Montgomery words (R = 2^32, values in [0, p)), one word per 64-bit lane.
//  * lane product: vpmuludq + two more for the Montgomery quotient (no even/odd shuffles);
//  * dense mat-vec: the 62-bit products x_j * M_ij are accumulated as 52-bit halves with vpmadd52luq / vpmadd52huq
//    (46 terms + a seed stay below 2^58), one Montgomery reduction per output word; the constants of the next full round enter
//    as the seed of the mat-vec (times R);
//  * the 22 partial rounds are collapsed by linearity (as in lf_poseidon_simd.cc): D = SX x (one mat-vec up front), a scalar
//    chain over word 0,  s_{r+1} = D_r + K_r + sum_{i<=r} G[r][i] X_i,  X_r = sbox(s_r + c_r),  and a closing map over
//    [x ; X].  The chain is the only sequential part and is kept at three dependent products per round: G[r][r] x^7 is formed as
//    ((G x) x^2) x^4 next to x^7 itself; everything else of s_{r+1} -- D, constants, the cross terms sum_{i<r} -- is prepared while
//    the S-box of round r runs.  The cross terms sum_{i<=r-2} G[r][i] X_i and the closing map's columns are accumulated by the
//    vector unit (E in memory, F in registers: one column per round), the last cross term is one scalar product.
// Plain host C++, compiled with the AVX-512 target for this file only.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

namespace lfbb {
namespace simd512 {

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;
typedef __m512i V;

namespace {
constexpr u32 P = 2013265921u;
constexpr u32 PINV = 0x88000001u;            // p^-1 mod 2^32
constexpr u32 NEGPINV = 0x77FFFFFFu;          // -p^-1 mod 2^32
constexpr u64 R1 = (1ull << 32) % P;
constexpr u64 R2 = (R1 * R1) % P;
constexpr int W = 24, RF = 8, RP = 22, NX = W + RP;

struct Tables {
    alignas(64) u64 mds[W][W];        // [j][i] = Montgomery form of M[i][j]
    alignas(64) u64 arkf[RF][W];      // constants of the full rounds (Montgomery form)
    alignas(64) u64 arks[RF][W];      // ... times R: the seed of the mat-vec in front of that round
    alignas(64) u64 sx[W][W];         // [j][r] = coefficient of state word j in D_r (column 0 and lanes >= 22 zero)
    alignas(64) u64 fin[NX][W];       // closing map, [j][i]: columns 0..23 state words, 24..45 the S-box outputs X_r; lane 0 zero
    alignas(64) u64 sxm[W][W];        // SX M (rows 0..21) and row 0 of M in lane 22: D and word 0 straight from the S-box outputs of the last full round
    alignas(64) u64 finm[W][W];       // (state columns of the closing map) M
    alignas(64) u64 fks[W];           // its constant + the constants of the full round behind it, times R
    alignas(64) u64 e[RP][W];         // [r][q] = G[q][r] for q >= r + 2 (the cross terms the vector unit accumulates), else 0
    u32 cst0[RP], Gd[RP], Gs[RP];     // constant of word 0, G[r][r], G[r][r-1]
    u32 Kc[RP];                       // K_q + cst0[q + 1] (the next round's constant of word 0 rides along)
    u32 ark40;                        // word 0 of the constants of full round RF/2
};
Tables T;

inline u32 smul(u32 a, u32 b) {   // scalar Montgomery product
    u64 pr = (u64)a * b;
    u32 q = (u32)pr * PINV;
    int64_t d = (int64_t)pr - (int64_t)((u64)q * P);
    int32_t t = (int32_t)(d >> 32);
    return (u32)(t < 0 ? t + (int32_t)P : t);
}
inline u32 sadd(u32 a, u32 b) { u32 s = a + b; return s >= P ? s - P : s; }
inline u32 to_mont(u64 x) { return smul((u32)(x % P), (u32)R2); }

inline V vP() { return _mm512_set1_epi64((long long)P); }
inline V vadd(V a, V b) {
    V s = _mm512_add_epi64(a, b);
    return _mm512_min_epu64(s, _mm512_sub_epi64(s, vP()));
}
inline V vmul(V a, V b) {   // Montgomery product of lanes in [0, p)
    const V mu = _mm512_set1_epi64((long long)PINV), p = vP();
    V pe = _mm512_mul_epu32(a, b);
    V q = _mm512_mul_epu32(pe, mu);
    V qp = _mm512_mul_epu32(q, p);
    V t = _mm512_srai_epi64(_mm512_sub_epi64(pe, qp), 32);        // low halves cancel; (-p, p)
    return _mm512_add_epi64(t, _mm512_and_si512(_mm512_srai_epi64(t, 63), p));
}
// Montgomery reduction of T = lo + 2^52 hi (lo < 2^58, hi < 2^16):  T 2^-32 = mred32(lo mod 2^32) + (lo >> 32) + 2^20 hi
inline V mred_wide(V lo, V hi) {
    const V mask = _mm512_set1_epi64(0xffffffffll), p = vP(), npinv = _mm512_set1_epi64((long long)NEGPINV);
    const V m31 = _mm512_set1_epi64(0x7fffffffll), c31 = _mm512_set1_epi64((long long)((1u << 27) - 1));
    V ll = _mm512_and_si512(lo, mask);
    V m = _mm512_mul_epu32(ll, npinv);
    V t = _mm512_srli_epi64(_mm512_add_epi64(ll, _mm512_mul_epu32(m, p)), 32);       // [0, p]
    V S = _mm512_add_epi64(_mm512_srli_epi64(lo, 32), _mm512_slli_epi64(hi, 20));     // < 2^37
    // 2^31 = 2^27 - 1 (mod p): fold twice
    S = _mm512_add_epi64(_mm512_and_si512(S, m31), _mm512_mul_epu32(_mm512_srli_epi64(S, 31), c31));   // < 2^31 + 2^33
    S = _mm512_add_epi64(_mm512_and_si512(S, m31), _mm512_mul_epu32(_mm512_srli_epi64(S, 31), c31));   // < 2^31 + 2^30
    S = _mm512_min_epu64(S, _mm512_sub_epi64(S, p));
    t = _mm512_min_epu64(t, _mm512_sub_epi64(t, p));
    V z = _mm512_add_epi64(S, t);
    return _mm512_min_epu64(z, _mm512_sub_epi64(z, p));
}
// lazy accumulators of a mat-vec with 24 output words: the 52-bit halves of the products, two sets (even / odd columns) so that
// consecutive columns do not wait for each other
struct Acc {
    V lo[3], hi[3], lo2[3], hi2[3];
    inline void init(const u64 *seed) {
        const V z = _mm512_setzero_si512();
        for (int g = 0; g < 3; g++) {
            lo[g] = seed ? _mm512_load_si512((const void *)(seed + 8 * g)) : z;
            hi[g] = lo2[g] = hi2[g] = z;
        }
    }
    inline void col2(const u64 *c0, const u64 *c1, u64 x0, u64 x1) {
        V b = _mm512_set1_epi64((long long)x0), b2 = _mm512_set1_epi64((long long)x1);
#pragma GCC unroll 3
        for (int g = 0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(c0 + 8 * g)), m2 = _mm512_load_si512((const void *)(c1 + 8 * g));
            lo[g] = _mm512_madd52lo_epu64(lo[g], m, b);
            hi[g] = _mm512_madd52hi_epu64(hi[g], m, b);
            lo2[g] = _mm512_madd52lo_epu64(lo2[g], m2, b2);
            hi2[g] = _mm512_madd52hi_epu64(hi2[g], m2, b2);
        }
    }
    inline void col(const u64 *c0, u64 x0) {
        V b = _mm512_set1_epi64((long long)x0);
#pragma GCC unroll 3
        for (int g = 0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(c0 + 8 * g));
            lo[g] = _mm512_madd52lo_epu64(lo[g], m, b);
            hi[g] = _mm512_madd52hi_epu64(hi[g], m, b);
        }
    }
    inline void finish(V x[3]) const {
        for (int g = 0; g < 3; g++) x[g] = mred_wide(_mm512_add_epi64(lo[g], lo2[g]), _mm512_add_epi64(hi[g], hi2[g]));
    }
};
// the same in memory: the cross terms of the partial rounds (one column per round, one lane read back per round)
struct AccMem {
    alignas(64) u64 lo[W], hi[W];
    inline void clear() { memset(this, 0, sizeof(*this)); }
    inline void col(const u64 *c0, u64 x0, int g0) {
        V b = _mm512_set1_epi64((long long)x0);
        for (int g = g0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(c0 + 8 * g));
            V l = _mm512_load_si512((const void *)(lo + 8 * g)), h = _mm512_load_si512((const void *)(hi + 8 * g));
            _mm512_store_si512((void *)(lo + 8 * g), _mm512_madd52lo_epu64(l, m, b));
            _mm512_store_si512((void *)(hi + 8 * g), _mm512_madd52hi_epu64(h, m, b));
        }
    }
};
inline void matvec(const u64 (*M)[W], V x[3], const u64 *seed) {
    alignas(64) u64 xs[W];
    for (int g = 0; g < 3; g++) _mm512_store_si512((void *)(xs + 8 * g), x[g]);
    Acc A;
    A.init(seed);
    for (int j = 0; j < W; j += 2) A.col2(M[j], M[j + 1], xs[j], xs[j + 1]);
    A.finish(x);
}
// S-box layer and MDS of a full round; the round constants were added by the producer of x (as the seed of its mat-vec), the
// constants of the NEXT full round are this mat-vec's seed
inline void sbox_layer(V x[3]) {
    V x2[3], x3[3], x4[3];
    for (int g = 0; g < 3; g++) x2[g] = vmul(x[g], x[g]);
    for (int g = 0; g < 3; g++) x3[g] = vmul(x2[g], x[g]);
    for (int g = 0; g < 3; g++) x4[g] = vmul(x2[g], x2[g]);
    for (int g = 0; g < 3; g++) x[g] = vmul(x4[g], x3[g]);
}
inline void full_round(V x[3], const u64 *seed) {
    sbox_layer(x);
    matvec(T.mds, x, seed);
}
// canonical host arithmetic for build()
inline u64 hadd(u64 a, u64 b) { u64 s = a + b; return s >= P ? s - P : s; }
inline u64 hmul(u64 a, u64 b) { return a * b % P; }
}  // namespace

bool supported() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq");
    return ok;
}

// canonical parameter tables (the numbers bb_host.cpp uses): ark[30*24], mds[24*24] row-major, cst[22*24], e00[22], row[22*23],
// col[22*23], post[23*23] row-major
void build(const u64 *ark, const u64 *mds, const u64 *cst, const u64 *e00, const u64 *row, const u64 *col, const u64 *post) {
    memset(&T, 0, sizeof(T));
    for (int i = 0; i < W; i++)
        for (int j = 0; j < W; j++) T.mds[j][i] = to_mont(mds[i * W + j]);
    for (int r = 0; r < RF; r++) {
        int src = r < RF / 2 ? r : RP + r;
        for (int i = 0; i < W; i++) {
            T.arkf[r][i] = to_mont(ark[(size_t)src * W + i]);
            T.arks[r][i] = to_mont(T.arkf[r][i]);
        }
    }
    T.ark40 = (u32)T.arkf[RF / 2][0];
    // Symbolic run of the 22 sparse partial rounds (canonical numbers).  Every state word 1..23 is an affine form over
    //   [ x_1..x_23 (words on entry) | X_0..X_21 (S-box outputs of word 0) | 1 ]
    // because a partial round is  xs = state[1..] + cst_r,  X_r = sbox(s0 + c0_r),  s0' = e00_r X_r + row_r . xs,
    // state'[1..] = xs + col_r X_r -- linear except for the S-box.  Collecting coefficients turns the rounds into
    //   D = SX x (one mat-vec up front),  s0_{r+1} = D_r + K_r + sum_{i<=r} G[r][i] X_i (scalar chain),
    //   state' = diag(1, post) [s0_22 ; x + CX X + ck] (one closing mat-vec over [x ; X]).
    const int n = W - 1, NB = n + RP + 1;   // basis size
    static u64 form[W - 1][W - 1 + RP + 1];
    static u64 G[RP][RP], K[RP], sxc[W][W], finc[W][W];
    memset(form, 0, sizeof(form));
    memset(G, 0, sizeof(G));
    memset(sxc, 0, sizeof(sxc));
    memset(finc, 0, sizeof(finc));
    for (int i = 0; i < n; i++) { form[i][i] = 1; form[i][NB - 1] = cst[0 * W + 1 + i] % P; }
    for (int r = 0; r < RP; r++) {
        T.cst0[r] = to_mont(cst[r * W]);
        u64 dotf[W - 1 + RP + 1];
        for (int b = 0; b < NB; b++) {
            u64 a = 0;
            for (int i = 0; i < n; i++) a = hadd(a, hmul(row[r * n + i] % P, form[i][b]));
            dotf[b] = a;
        }
        for (int j = 0; j < n; j++) { T.sx[1 + j][r] = to_mont(dotf[j]); sxc[1 + j][r] = dotf[j]; }
        for (int i = 0; i < r; i++) G[r][i] = dotf[n + i];
        G[r][r] = e00[r] % P;
        K[r] = dotf[NB - 1];
        for (int i = 0; i < n; i++) {
            form[i][n + r] = hadd(form[i][n + r], col[r * n + i] % P);
            if (r + 1 < RP) form[i][NB - 1] = hadd(form[i][NB - 1], cst[(r + 1) * W + 1 + i] % P);
        }
    }
    // closing map: words 1..23 = post * form
    for (int i = 0; i < n; i++)
        for (int b = 0; b < NB; b++) {
            u64 a = 0;
            for (int k = 0; k < n; k++) a = hadd(a, hmul(post[i * n + k] % P, form[k][b]));
            if (b < n) { T.fin[1 + b][1 + i] = to_mont(a); finc[1 + b][1 + i] = a; }
            else if (b < n + RP) T.fin[W + (b - n)][1 + i] = to_mont(a);
            else T.fks[1 + i] = to_mont(sadd(to_mont(a), (u32)T.arkf[RF / 2][1 + i]));   // (constant + next round's constant) R
        }
    for (int r = 0; r < RP; r++) {
        T.Gd[r] = to_mont(G[r][r]);
        T.Gs[r] = r ? to_mont(G[r][r - 1]) : 0;
        T.Kc[r] = sadd(to_mont(K[r]), r + 1 < RP ? T.cst0[r + 1] : 0);
        for (int q = r + 2; q < RP; q++) T.e[r][q] = to_mont(G[q][r]);
    }
    // The mat-vec of the full round in front of the partial rounds is folded into what consumes its output: x = M s, D = (SX M) s, closing-map part
    // (FIN_x M) s, word 0 = (row 0 of M) s in lane 22 of the D table: two mat-vecs over s instead of three
    for (int j = 0; j < W; j++)
        for (int r = 0; r < W; r++) {
            u64 a = 0, b = 0;
            for (int i = 0; i < W; i++) {
                a = hadd(a, hmul(sxc[i][r], mds[i * W + j] % P));
                b = hadd(b, hmul(finc[i][r], mds[i * W + j] % P));
            }
            if (r == RP) a = mds[0 * W + j] % P;
            T.sxm[j][r] = to_mont(a);
            T.finm[j][r] = to_mont(b);
        }
}

void permute(u64 st[24]) {
    const V r2 = _mm512_set1_epi64((long long)R2);
    V x[3];
    for (int g = 0; g < 3; g++)   // to Montgomery form, + the constants of the first round
        x[g] = vadd(vmul(_mm512_loadu_si512((const void *)(st + 8 * g)), r2), _mm512_load_si512((const void *)(T.arkf[0] + 8 * g)));
    for (int r = 0; r + 1 < RF / 2; r++) full_round(x, T.arks[r + 1]);
    sbox_layer(x);      // the last full round of the first half: its mat-vec is folded into the tables below
    alignas(64) u64 xs[W], d[W];
    for (int g = 0; g < 3; g++) _mm512_store_si512((void *)(xs + 8 * g), x[g]);
    {
        Acc D;
        D.init(nullptr);
        for (int j = 0; j < W; j += 2) D.col2(T.sxm[j], T.sxm[j + 1], xs[j], xs[j + 1]);
        V dv[3];
        D.finish(dv);
        for (int g = 0; g < 3; g++) _mm512_store_si512((void *)(d + 8 * g), dv[g]);
    }
    Acc F;
    F.init(T.fks);
    for (int j = 0; j < W; j += 2) F.col2(T.finm[j], T.finm[j + 1], xs[j], xs[j + 1]);   // lane 0 of every column is zero
    AccMem E;
    E.clear();
    // sum of Montgomery products (below 2^67) -> Montgomery form of the sum
    auto mred = [](u128 t) {
        u32 m = (u32)t * NEGPINV;
        u64 r = (u64)((t + (u128)m * P) >> 32);     // < 2^36
        return (u32)(r % P);
    };
    u32 s = sadd((u32)d[RP], T.cst0[0]);               // word 0 of the state: lane 22 of the D table
    u32 base = sadd((u32)d[0], T.Kc[0]);
    for (int r = 0; r < RP; r++) {
        // next round's base without its X_r term (lane r + 1 of E is complete: its last term came from X_{r-1}, stored a round ago)
        u32 lp = 0;
        if (r + 1 < RP) {
            const int q = r + 1;
            lp = sadd(mred((u128)E.lo[q] + ((u128)E.hi[q] << 52)), sadd((u32)d[q], T.Kc[q]));
        }
        // chain: three dependent products.  x^7 = x^3 x^4 for the columns, G x^7 = ((G x) x^2) x^4 for the next round
        const u32 x2 = smul(s, s), gx = smul(T.Gd[r], s);
        const u32 x3 = smul(x2, s), x4 = smul(x2, x2), gx3 = smul(gx, x2);
        s = sadd(smul(gx3, x4), base);
        const u32 X = smul(x3, x4);
        if (r + 1 < RP) base = sadd(smul(T.Gs[r + 1], X), lp);
        F.col(T.fin[W + r], X);
        if (r + 2 < RP) E.col(T.e[r], X, (r + 2) >> 3);
    }
    F.finish(x);
    x[0] = _mm512_mask_set1_epi64(x[0], 0x01, (long long)sadd(s, T.ark40));   // word 0 of the closing map is the chain's last value (+ the next round's constant)
    for (int r = RF / 2; r < RF; r++) full_round(x, r + 1 < RF ? T.arks[r + 1] : nullptr);
    const V one = _mm512_set1_epi64(1);
    for (int g = 0; g < 3; g++) _mm512_storeu_si512((void *)(st + 8 * g), vmul(x[g], one));        // back to canonical
}

}  // namespace simd512
}  // namespace lfbb

// bb_poseidon_simd.h -- AVX2 implementation of the BabyBear Poseidon permutation (width 24, 8 full + 22 partial rounds,
// alpha 7; same sparse partial-round factorisation as bb_host.cpp).  The transcript of a BabyBear fold step needs ~5300
// permutations (a ring element is 72 words = 3.6 permutations), all on the host and partly on the critical path.
//
// State: three __m256i of eight 32-bit Montgomery words (R = 2^32) in [0, p).  The dense mat-vecs broadcast one state word
// at a time against the transposed matrix held in 64-bit lanes and sum the 62-bit products four at a time before splitting
// them into 32-bit halves (no overflow: 4 (p-1)^2 < 2^64); everything else is 8-lane Montgomery arithmetic.
// Included only by bb_host.cpp (host pass, -march=x86-64-v3).
#pragma once
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

namespace lfbb {
namespace simd {

typedef uint64_t u64;
typedef uint32_t u32;
constexpr u32 P = 2013265921u;
constexpr u32 PINV = 0x88000001u;            // p^-1 mod 2^32
constexpr u32 NEGPINV = 0x77FFFFFFu;          // -p^-1 mod 2^32
constexpr u64 R1 = (1ull << 32) % P;          // Montgomery form of 1
constexpr u64 R2 = (R1 * R1) % P;

static inline __m256i vP() { return _mm256_set1_epi32((int)P); }
// canonical result of a + b, a, b in [0, p)
static inline __m256i vadd(__m256i a, __m256i b) {
    __m256i s = _mm256_add_epi32(a, b);
    return _mm256_min_epu32(s, _mm256_sub_epi32(s, vP()));
}
// 8-lane Montgomery product a * b * 2^-32 mod p, inputs and output in [0, p)
static inline __m256i vmul(__m256i a, __m256i b) {
    const __m256i mu = _mm256_set1_epi32((int)PINV), p = vP();
    __m256i ao = _mm256_srli_epi64(a, 32), bo = _mm256_srli_epi64(b, 32);
    __m256i pe = _mm256_mul_epu32(a, b), po = _mm256_mul_epu32(ao, bo);
    __m256i qe = _mm256_mul_epu32(pe, mu), qo = _mm256_mul_epu32(po, mu);
    __m256i qpe = _mm256_mul_epu32(qe, p), qpo = _mm256_mul_epu32(qo, p);
    __m256i de = _mm256_sub_epi64(pe, qpe), dd = _mm256_sub_epi64(po, qpo);   // low halves cancel; high halves in (-p, p)
    __m256i t = _mm256_blend_epi32(_mm256_srli_epi64(de, 32), dd, 0xAA);
    __m256i corr = _mm256_and_si256(_mm256_srai_epi32(t, 31), p);
    return _mm256_add_epi32(t, corr);
}
static inline u32 smul(u32 a, u32 b) {   // scalar Montgomery product
    u64 pr = (u64)a * b;
    u32 q = (u32)pr * PINV;
    int64_t d = (int64_t)pr - (int64_t)((u64)q * P);
    int32_t t = (int32_t)(d >> 32);
    return (u32)(t < 0 ? t + (int32_t)P : t);
}
static inline u32 to_mont(u64 x) { return smul((u32)(x % P), (u32)R2); }

struct Tables {
    // dense matrices in "even/odd" 64-bit-lane layout: for state vector k, E[k] holds lanes 8k+{0,2,4,6}, O[k] lanes 8k+{1,3,5,7}
    alignas(32) u64 mds[24][6][4];
    alignas(32) u64 post[24][6][4];      // diag(1, post) as a 24 x 24 map
    alignas(32) u32 ark_full[8][24];
    alignas(32) u32 cst[22][24];
    alignas(32) u32 row[22][24];         // lane 0 = e00, lanes 1.. = row
    alignas(32) u32 col[22][24];         // lane 0 = 0
};

// load canonical parameter tables (same numbers bb_host.cpp uses) into Montgomery vector form
static inline void build_tables(Tables &T, const u64 *ark /*30x24*/, const u64 *mds /*24x24*/, const u64 (*cst)[24], const u64 *e00,
                                const u64 (*row)[23], const u64 (*col)[23], const u64 (*post)[23]) {
    memset(&T, 0, sizeof(T));
    auto put = [](u64 (*dst)[6][4], int j, int i, u32 v) {   // output lane i of column j
        int k = i / 8, l = i % 8;
        dst[j][2 * k + (l & 1)][l >> 1] = v;
    };
    for (int i = 0; i < 24; i++)
        for (int j = 0; j < 24; j++) put(T.mds, j, i, to_mont(mds[i * 24 + j]));
    put(T.post, 0, 0, to_mont(1));
    for (int i = 0; i < 23; i++)
        for (int j = 0; j < 23; j++) put(T.post, 1 + j, 1 + i, to_mont(post[i][j]));
    for (int r = 0; r < 8; r++) {
        int rr = r < 4 ? r : 22 + r;   // full rounds 0..3 and 26..29
        for (int i = 0; i < 24; i++) T.ark_full[r][i] = to_mont(ark[rr * 24 + i]);
    }
    for (int r = 0; r < 22; r++) {
        for (int i = 0; i < 24; i++) T.cst[r][i] = to_mont(cst[r][i]);
        T.row[r][0] = to_mont(e00[r]);
        T.col[r][0] = 0;
        for (int i = 0; i < 23; i++) { T.row[r][1 + i] = to_mont(row[r][i]); T.col[r][1 + i] = to_mont(col[r][i]); }
    }
}

// s <- M s for a dense matrix in the even/odd layout
static inline void matvec(const u64 (*M)[6][4], __m256i s[3]) {
    alignas(32) u32 x[24];
    _mm256_store_si256((__m256i *)x, s[0]);
    _mm256_store_si256((__m256i *)(x + 8), s[1]);
    _mm256_store_si256((__m256i *)(x + 16), s[2]);
    const __m256i mask = _mm256_set1_epi64x(0xffffffffll);
    __m256i lo[6], hi[6];
    // two passes of three output vectors each: 3 x (acc, lo, hi) + broadcast + mask fit the 16 ymm registers
    for (int h = 0; h < 2; h++) {
        __m256i l0 = _mm256_setzero_si256(), l1 = l0, l2 = l0, h0 = l0, h1 = l0, h2 = l0;
        for (int j0 = 0; j0 < 24; j0 += 4) {
            __m256i a0 = _mm256_setzero_si256(), a1 = a0, a2 = a0;
#pragma GCC unroll 4
            for (int j = j0; j < j0 + 4; j++) {
                __m256i xb = _mm256_set1_epi64x((long long)x[j]);
                a0 = _mm256_add_epi64(a0, _mm256_mul_epu32(xb, _mm256_load_si256((const __m256i *)M[j][3 * h])));
                a1 = _mm256_add_epi64(a1, _mm256_mul_epu32(xb, _mm256_load_si256((const __m256i *)M[j][3 * h + 1])));
                a2 = _mm256_add_epi64(a2, _mm256_mul_epu32(xb, _mm256_load_si256((const __m256i *)M[j][3 * h + 2])));
            }
            l0 = _mm256_add_epi64(l0, _mm256_and_si256(a0, mask)); h0 = _mm256_add_epi64(h0, _mm256_srli_epi64(a0, 32));
            l1 = _mm256_add_epi64(l1, _mm256_and_si256(a1, mask)); h1 = _mm256_add_epi64(h1, _mm256_srli_epi64(a1, 32));
            l2 = _mm256_add_epi64(l2, _mm256_and_si256(a2, mask)); h2 = _mm256_add_epi64(h2, _mm256_srli_epi64(a2, 32));
        }
        lo[3 * h] = l0; lo[3 * h + 1] = l1; lo[3 * h + 2] = l2;
        hi[3 * h] = h0; hi[3 * h + 1] = h1; hi[3 * h + 2] = h2;
    }
    // T = hi * 2^32 + lo;  T * 2^-32 = (hi + (lo >> 32)) + mred32(lo & mask)
    const __m256i p64 = _mm256_set1_epi64x((long long)P), npinv = _mm256_set1_epi64x((long long)NEGPINV);
    const __m256i c31 = _mm256_set1_epi64x((long long)((1u << 27) - 1)), m31 = _mm256_set1_epi64x(0x7fffffffll);
    __m256i r[6];
    for (int v = 0; v < 6; v++) {
        __m256i S = _mm256_add_epi64(hi[v], _mm256_srli_epi64(lo[v], 32));          // < 2^35
        __m256i ll = _mm256_and_si256(lo[v], mask);
        __m256i m = _mm256_mul_epu32(ll, npinv);                                     // low 32 bits used below
        __m256i t = _mm256_srli_epi64(_mm256_add_epi64(ll, _mm256_mul_epu32(m, p64)), 32);   // mred32(ll) in [0, p]
        // S mod p: 2^31 = 2^27 - 1 (mod p)
        __m256i Sr = _mm256_add_epi64(_mm256_and_si256(S, m31), _mm256_mul_epu32(_mm256_srli_epi64(S, 31), c31));   // < 2^32
        Sr = _mm256_min_epu32(Sr, _mm256_sub_epi64(Sr, p64));
        Sr = _mm256_min_epu32(Sr, _mm256_sub_epi64(Sr, p64));                        // < p
        t = _mm256_min_epu32(t, _mm256_sub_epi64(t, p64));                           // [0, p)
        __m256i z = _mm256_add_epi64(Sr, t);
        r[v] = _mm256_min_epu32(z, _mm256_sub_epi64(z, p64));
    }
    for (int k = 0; k < 3; k++) s[k] = _mm256_or_si256(r[2 * k], _mm256_slli_epi64(r[2 * k + 1], 32));
}
static inline __m256i vpow7(__m256i x) {
    __m256i x2 = vmul(x, x), x3 = vmul(x2, x), x4 = vmul(x2, x2);
    return vmul(x4, x3);
}
static inline u32 spow7(u32 x) {
    u32 x2 = smul(x, x), x3 = smul(x2, x), x4 = smul(x2, x2);
    return smul(x4, x3);
}

static inline void permute(const Tables &T, u64 st[24]) {
    alignas(32) u32 w[24];
    for (int i = 0; i < 24; i++) w[i] = (u32)st[i];
    const __m256i r2 = _mm256_set1_epi32((int)R2);
    __m256i s[3];
    for (int k = 0; k < 3; k++) s[k] = vmul(_mm256_load_si256((const __m256i *)(w + 8 * k)), r2);   // to Montgomery form
    auto full = [&](int r) {
        for (int k = 0; k < 3; k++) s[k] = vpow7(vadd(s[k], _mm256_load_si256((const __m256i *)(T.ark_full[r] + 8 * k))));
        matvec(T.mds, s);
    };
    for (int r = 0; r < 4; r++) full(r);
    for (int r = 0; r < 22; r++) {
        for (int k = 0; k < 3; k++) s[k] = vadd(s[k], _mm256_load_si256((const __m256i *)(T.cst[r] + 8 * k)));
        u32 x0 = spow7((u32)_mm256_extract_epi32(s[0], 0));
        s[0] = _mm256_insert_epi32(s[0], (int)x0, 0);
        // y0 = (e00, row) . state ; state[1..] += col * x0
        __m256i pr0 = vmul(s[0], _mm256_load_si256((const __m256i *)(T.row[r])));
        __m256i pr1 = vmul(s[1], _mm256_load_si256((const __m256i *)(T.row[r] + 8)));
        __m256i pr2 = vmul(s[2], _mm256_load_si256((const __m256i *)(T.row[r] + 16)));
        // horizontal sum of 24 words < p: widen to 64-bit lanes
        const __m256i mask = _mm256_set1_epi64x(0xffffffffll);
        __m256i sum = _mm256_add_epi64(_mm256_add_epi64(_mm256_and_si256(pr0, mask), _mm256_srli_epi64(pr0, 32)),
                                       _mm256_add_epi64(_mm256_add_epi64(_mm256_and_si256(pr1, mask), _mm256_srli_epi64(pr1, 32)),
                                                        _mm256_add_epi64(_mm256_and_si256(pr2, mask), _mm256_srli_epi64(pr2, 32))));
        alignas(32) u64 hs[4];
        _mm256_store_si256((__m256i *)hs, sum);
        u32 y0 = (u32)((hs[0] + hs[1] + hs[2] + hs[3]) % P);
        __m256i xb = _mm256_set1_epi32((int)x0);
        s[0] = vadd(s[0], vmul(xb, _mm256_load_si256((const __m256i *)(T.col[r]))));
        s[1] = vadd(s[1], vmul(xb, _mm256_load_si256((const __m256i *)(T.col[r] + 8))));
        s[2] = vadd(s[2], vmul(xb, _mm256_load_si256((const __m256i *)(T.col[r] + 16))));
        s[0] = _mm256_insert_epi32(s[0], (int)y0, 0);
    }
    matvec(T.post, s);
    for (int r = 4; r < 8; r++) full(r);
    const __m256i one = _mm256_set1_epi32(1);
    for (int k = 0; k < 3; k++) _mm256_store_si256((__m256i *)(w + 8 * k), vmul(s[k], one));   // back to canonical
    for (int i = 0; i < 24; i++) st[i] = w[i];
}

}  // namespace simd
}  // namespace lfbb

// bb_poseidon_avx512.cc -- AVX-512 (IFMA) lanes for the BabyBear Poseidon permutation (width 24, 8 full + 22 partial rounds,
// alpha 7; the sparse partial-round factorisation of bb_host.cpp).  A BabyBear fold step needs ~5300 permutations (a ring
// element is 72 words), all on the host; this path is selected at run time when the CPU has avx512f/ifma/dq
// (bb_poseidon_simd.h, AVX2, otherwise; LF_POSEIDON_SCALAR=1 forces the scalar code).
//
// State: three zmm registers of eight Montgomery words (R = 2^32, values in [0, p)), one word per 64-bit lane.
//  * lane product: vpmuludq + two more for the Montgomery quotient (no even/odd shuffles);
//  * dense mat-vec: the 62-bit products x_j * M_ij are accumulated as 52-bit halves with vpmadd52luq / vpmadd52huq
//    (24 terms stay below 2^57), one Montgomery reduction per output word.
// Plain host C++, compiled with the AVX-512 target for this file only.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

namespace lfbb {
namespace simd512 {

typedef uint64_t u64;
typedef uint32_t u32;
typedef __m512i V;

namespace {
constexpr u32 P = 2013265921u;
constexpr u32 PINV = 0x88000001u;            // p^-1 mod 2^32
constexpr u32 NEGPINV = 0x77FFFFFFu;          // -p^-1 mod 2^32
constexpr u64 R1 = (1ull << 32) % P;
constexpr u64 R2 = (R1 * R1) % P;
constexpr int W = 24, RF = 8, RP = 22;

struct Tables {
    alignas(64) u64 mds[W][W];        // [j][i] = Montgomery form of M[i][j]
    alignas(64) u64 post[W][W];       // diag(1, post), same layout
    alignas(64) u64 arkf[RF][W];
    alignas(64) u64 cst[RP][W];       // lane 0 cleared (cst0)
    alignas(64) u64 row[RP][W];       // lane 0 cleared (e00)
    alignas(64) u64 col[RP][W];       // lane 0 cleared
    u32 cst0[RP], e00[RP];
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
inline u32 spow7(u32 x) {
    u32 x2 = smul(x, x), x3 = smul(x2, x), x4 = smul(x2, x2);
    return smul(x4, x3);
}

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
inline V vpow7(V x) {
    V x2 = vmul(x, x), x3 = vmul(x2, x), x4 = vmul(x2, x2);
    return vmul(x4, x3);
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
inline void matvec(const u64 (*M)[W], V x[3]) {
    alignas(64) u64 xs[W];
    for (int g = 0; g < 3; g++) _mm512_store_si512((void *)(xs + 8 * g), x[g]);
    const V z = _mm512_setzero_si512();
    V lo[3] = {z, z, z}, hi[3] = {z, z, z}, lo2[3] = {z, z, z}, hi2[3] = {z, z, z};
    for (int j = 0; j < W; j += 2) {
        V b = _mm512_set1_epi64((long long)xs[j]), b2 = _mm512_set1_epi64((long long)xs[j + 1]);
#pragma GCC unroll 3
        for (int g = 0; g < 3; g++) {
            V m = _mm512_load_si512((const void *)(M[j] + 8 * g)), m2 = _mm512_load_si512((const void *)(M[j + 1] + 8 * g));
            lo[g] = _mm512_madd52lo_epu64(lo[g], m, b);
            hi[g] = _mm512_madd52hi_epu64(hi[g], m, b);
            lo2[g] = _mm512_madd52lo_epu64(lo2[g], m2, b2);
            hi2[g] = _mm512_madd52hi_epu64(hi2[g], m2, b2);
        }
    }
    for (int g = 0; g < 3; g++) x[g] = mred_wide(_mm512_add_epi64(lo[g], lo2[g]), _mm512_add_epi64(hi[g], hi2[g]));
}
inline void full_round(V x[3], const u64 *ark) {
    V t[3];
    for (int g = 0; g < 3; g++) t[g] = vadd(x[g], _mm512_load_si512((const void *)(ark + 8 * g)));
    V x2[3], x3[3], x4[3];
    for (int g = 0; g < 3; g++) x2[g] = vmul(t[g], t[g]);
    for (int g = 0; g < 3; g++) x3[g] = vmul(x2[g], t[g]);
    for (int g = 0; g < 3; g++) x4[g] = vmul(x2[g], x2[g]);
    for (int g = 0; g < 3; g++) x[g] = vmul(x4[g], x3[g]);
    matvec(T.mds, x);
}
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
        for (int j = 0; j < W; j++) {
            T.mds[j][i] = to_mont(mds[i * W + j]);
            u64 e = (i == 0 || j == 0) ? (u64)(i == j) : post[(i - 1) * (W - 1) + (j - 1)];
            T.post[j][i] = to_mont(e);
        }
    for (int r = 0; r < RF; r++) {
        int src = r < RF / 2 ? r : RP + r;
        for (int i = 0; i < W; i++) T.arkf[r][i] = to_mont(ark[(size_t)src * W + i]);
    }
    for (int r = 0; r < RP; r++) {
        T.cst0[r] = to_mont(cst[r * W]);
        T.e00[r] = to_mont(e00[r]);
        for (int i = 1; i < W; i++) {
            T.cst[r][i] = to_mont(cst[r * W + i]);
            T.row[r][i] = to_mont(row[r * (W - 1) + i - 1]);
            T.col[r][i] = to_mont(col[r * (W - 1) + i - 1]);
        }
    }
}

void permute(u64 st[24]) {
    const V r2 = _mm512_set1_epi64((long long)R2);
    V x[3];
    for (int g = 0; g < 3; g++) x[g] = vmul(_mm512_loadu_si512((const void *)(st + 8 * g)), r2);   // to Montgomery form
    for (int r = 0; r < RF / 2; r++) full_round(x, T.arkf[r]);
    u32 s0 = (u32)_mm_cvtsi128_si64(_mm512_castsi512_si128(x[0]));
    x[0] = _mm512_maskz_mov_epi64(0xFE, x[0]);
    for (int r = 0; r < RP; r++) {
        V xs[3], pr[3];
        for (int g = 0; g < 3; g++) {
            xs[g] = vadd(x[g], _mm512_load_si512((const void *)(T.cst[r] + 8 * g)));
            pr[g] = vmul(xs[g], _mm512_load_si512((const void *)(T.row[r] + 8 * g)));
        }
        u64 dot = (u64)_mm512_reduce_add_epi64(_mm512_add_epi64(_mm512_add_epi64(pr[0], pr[1]), pr[2]));   // < 23 p
        u32 x0 = spow7(sadd(s0, T.cst0[r]));
        V xb = _mm512_set1_epi64((long long)x0);
        for (int g = 0; g < 3; g++) x[g] = vadd(xs[g], vmul(xb, _mm512_load_si512((const void *)(T.col[r] + 8 * g))));
        s0 = (u32)((dot + smul(T.e00[r], x0)) % P);
    }
    x[0] = _mm512_mask_set1_epi64(x[0], 0x01, (long long)s0);
    matvec(T.post, x);
    for (int r = RF / 2; r < RF; r++) full_round(x, T.arkf[r]);
    const V one = _mm512_set1_epi64(1);
    for (int g = 0; g < 3; g++) _mm512_storeu_si512((void *)(st + 8 * g), vmul(x[g], one));        // back to canonical
}

}  // namespace simd512
}  // namespace lfbb

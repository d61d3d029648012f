// lf_host.cpp -- ring tables, small host ring ops, Poseidon (Grain-generated constants), transcript.
// Reference anchors: cyclotomic-rings/src/rings.rs:28-40 (CRT splitting), rings/goldilocks.rs:36-68
// (challenge set), rings/poseidon/goldilocks.rs:7-1425 (Poseidon parameters: regenerated, not copied),
// latticefold/src/transcript/poseidon.rs:29-75 (transcript), ark-crypto-primitives 0.4.0 PoseidonSponge.
#include "lf_host.h"
#include "lf_poseidon_simd.h"

#include <string.h>

#include <mutex>
#include <utility>
#include <vector>
#include <stdlib.h>

namespace lf {

// ---------------------------------------------------------------------------------------------------------
void default_ring(u64 *nonres, u64 y[24]) {
    // Phi_72(X) = prod_{e in (Z/24)^*} (X^3 - z^e) with z = 2^40 of multiplicative order 24.
    // F_{p^3} = F_p[Y]/(Y^3 - z).  e = 1 (mod 3): X -> z^((e-1)/3) Y ; e = 2 (mod 3): X -> z^((e-2)/3) Y^2.
    static const int E[8] = {1, 5, 7, 11, 13, 17, 19, 23};
    const u64 z = 1ULL << 40;
    *nonres = z;
    memset(y, 0, 24 * sizeof(u64));
    for (int k = 0; k < 8; k++) {
        int e = E[k];
        if (e % 3 == 1) y[3 * k + 1] = fq_pow(z, (u64)(e - 1) / 3);
        else y[3 * k + 2] = fq_pow(z, (u64)(e - 2) / 3);
    }
}

static Fq3 mul3g(Fq3 a, Fq3 b, u64 nu) { return fq3_mul<false>(a, b, nu); }

int build_crt_tables(u64 nonres, const u64 y[24], CrtTables &T) {
    memset(&T, 0, sizeof(T));
    T.nu = nonres % LF_P;
    T.nu_is_2p40 = (T.nu == (1ULL << 40));
    u64 zeta[8];
    for (int k = 0; k < 8; k++) {
        T.y[k] = fq3_make(y[3 * k] % LF_P, y[3 * k + 1] % LF_P, y[3 * k + 2] % LF_P);
        Fq3 cube = mul3g(mul3g(T.y[k], T.y[k], T.nu), T.y[k], T.nu);
        if (cube.c[1] || cube.c[2]) return -1;
        zeta[k] = cube.c[0];
        u64 z4 = fq_pow(zeta[k], 4), z8 = fq_mul(z4, z4);
        if (fq_add(fq_sub(z8, z4), 1) != 0) return -1;  // must be a root of Phi_24
        for (int j = 0; j < k; j++)
            if (zeta[j] == zeta[k]) return -1;
        // y_k and y_k^2 must be monomials c*Y^m (they always are when y_k^3 lies in F_p and Y^3 = nu)
        if (T.y[k].c[0] != 0 || (T.y[k].c[1] != 0) == (T.y[k].c[2] != 0)) return -1;
    }
    // dense tables
    for (int k = 0; k < 8; k++) {
        Fq3 p = fq3_one();
        for (int c = 0; c < 24; c++) {
            T.ypow[k][c] = p;
            p = mul3g(p, T.y[k], T.nu);
        }
    }
    {
        static thread_local u64 M[24][48];
        for (int k = 0; k < 8; k++)
            for (int c = 0; c < 24; c++)
                for (int q = 0; q < 3; q++) M[3 * k + q][c] = T.ypow[k][c].c[q];
        for (int r = 0; r < 24; r++)
            for (int c = 0; c < 24; c++) M[r][24 + c] = (r == c);
        for (int col = 0; col < 24; col++) {
            int piv = -1;
            for (int r = col; r < 24; r++)
                if (M[r][col]) { piv = r; break; }
            if (piv < 0) return -1;
            if (piv != col)
                for (int c = 0; c < 48; c++) { u64 t = M[piv][c]; M[piv][c] = M[col][c]; M[col][c] = t; }
            u64 inv = fq_inv(M[col][col]);
            for (int c = 0; c < 48; c++) M[col][c] = fq_mul(M[col][c], inv);
            for (int r = 0; r < 24; r++) {
                if (r == col || !M[r][col]) continue;
                u64 f = M[r][col];
                for (int c = 0; c < 48; c++) M[r][c] = fq_sub(M[r][c], fq_mul(f, M[col][c]));
            }
        }
        for (int r = 0; r < 24; r++)
            for (int c = 0; c < 24; c++) T.icrt[r][c] = M[r][24 + c];
    }
    // butterfly tables relative to omega = zeta of slot 0
    u64 w = zeta[0];
    T.w1 = w; T.w2 = fq_pow(w, 2); T.w4 = fq_pow(w, 4); T.w5 = fq_pow(w, 5); T.w7 = fq_pow(w, 7);
    T.w10 = fq_pow(w, 10); T.w11 = fq_pow(w, 11);
    T.inv2 = fq_inv(2);
    T.inv_1m2w4 = fq_inv(fq_sub(1, fq_add(T.w4, T.w4)));
    T.i2w2 = fq_inv(fq_add(T.w2, T.w2)); T.i2w10 = fq_inv(fq_add(T.w10, T.w10));
    T.i2w1 = fq_inv(fq_add(T.w1, T.w1)); T.i2w7 = fq_inv(fq_add(T.w7, T.w7));
    T.i2w5 = fq_inv(fq_add(T.w5, T.w5)); T.i2w11 = fq_inv(fq_add(T.w11, T.w11));
    static const int ENAT[8] = {1, 13, 7, 19, 5, 17, 11, 23};  // butterfly output order (exponent of omega)
    for (int p = 0; p < 8; p++) {
        u64 root = fq_pow(w, (u64)ENAT[p]);
        int slot = -1;
        for (int k = 0; k < 8; k++)
            if (zeta[k] == root) slot = k;
        if (slot < 0) return -1;
        T.slot_of_pos[p] = slot;
        Fq3 y1 = T.y[slot], y2 = mul3g(y1, y1, T.nu);
        T.pos1[p] = y1.c[1] ? 1 : 2;
        T.tw1[p] = y1.c[T.pos1[p]];
        if (y2.c[0] != 0 || (y2.c[1] != 0) == (y2.c[2] != 0)) return -1;
        T.pos2[p] = y2.c[1] ? 1 : 2;
        T.tw2[p] = y2.c[T.pos2[p]];
        if (T.pos1[p] == T.pos2[p]) return -1;
        T.itw1[p] = fq_inv(T.tw1[p]);
        T.itw2[p] = fq_inv(T.tw2[p]);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
void HostRing::crt(const u64 *a, u64 *out) const {
    for (int k = 0; k < 8; k++) {
        Fq3 acc = fq3_zero();
        for (int c = 0; c < 24; c++)
            if (a[c]) acc = fq3_add(acc, fq3_mul_fq(T.ypow[k][c], a[c]));
        out[3 * k] = acc.c[0]; out[3 * k + 1] = acc.c[1]; out[3 * k + 2] = acc.c[2];
    }
}
void HostRing::icrt(const u64 *x, u64 *out) const {
    u64 r[24];
    for (int i = 0; i < 24; i++) {
        u64 acc = 0;
        for (int j = 0; j < 24; j++)
            if (x[j]) acc = fq_add(acc, fq_mul(T.icrt[i][j], x[j]));
        r[i] = acc;
    }
    memcpy(out, r, sizeof(r));
}
void HostRing::mul_ntt(const u64 *a, const u64 *b, u64 *out) const {
    u64 r[24];
    for (int k = 0; k < 8; k++) {
        Fq3 p = mul3(fq3_make(a[3 * k], a[3 * k + 1], a[3 * k + 2]), fq3_make(b[3 * k], b[3 * k + 1], b[3 * k + 2]));
        r[3 * k] = p.c[0]; r[3 * k + 1] = p.c[1]; r[3 * k + 2] = p.c[2];
    }
    memcpy(out, r, sizeof(r));
}
void HostRing::mul_fq3(const u64 *a, Fq3 s, u64 *out) const {
    for (int k = 0; k < 8; k++) {
        Fq3 p = mul3(fq3_make(a[3 * k], a[3 * k + 1], a[3 * k + 2]), s);
        out[3 * k] = p.c[0]; out[3 * k + 1] = p.c[1]; out[3 * k + 2] = p.c[2];
    }
}
void HostRing::add(const u64 *a, const u64 *b, u64 *out) { for (int i = 0; i < 24; i++) out[i] = fq_add(a[i], b[i]); }
void HostRing::sub(const u64 *a, const u64 *b, u64 *out) { for (int i = 0; i < 24; i++) out[i] = fq_sub(a[i], b[i]); }
void HostRing::from_u64(u64 v, u64 *out) {
    v %= LF_P;
    for (int k = 0; k < 8; k++) { out[3 * k] = v; out[3 * k + 1] = 0; out[3 * k + 2] = 0; }
}
void HostRing::from_fq3(Fq3 s, u64 *out) {
    for (int k = 0; k < 8; k++) { out[3 * k] = s.c[0]; out[3 * k + 1] = s.c[1]; out[3 * k + 2] = s.c[2]; }
}
bool HostRing::is_diag(const u64 *e, Fq3 *out) {
    for (int k = 1; k < 8; k++)
        if (e[3 * k] != e[0] || e[3 * k + 1] != e[1] || e[3 * k + 2] != e[2]) return false;
    if (out) *out = fq3_make(e[0], e[1], e[2]);
    return true;
}

// stark_rings::balanced_decomposition as recollected (source absent; convention is DATA-level "unpinned"):
// centred lift, truncating remainder, |rem| <= b/2 kept, otherwise rem -+ b with carry +-1, zero padded.
void balanced_digits(u64 v, u64 base, unsigned digits, int64_t *out, int mode) {
    __int128 b = (__int128)base, half = b / 2;
    __int128 cur = v <= (LF_P - 1) / 2 ? (__int128)v : (__int128)v - (__int128)LF_P;
    for (unsigned k = 0; k < digits; k++) {
        __int128 rem = cur % b, q = cur / b;
        if (mode == 1 && base > 2) {   // floor rule: digits in [-base/2, base/2)
            if (rem < 0) rem += b;
            if (rem >= half) rem -= b;
            q = (cur - rem) / b;
        } else {
            __int128 ar = rem < 0 ? -rem : rem;
            if (ar > half) {
                if (rem < 0) { rem += b; q -= 1; }
                else { rem -= b; q += 1; }
            }
        }
        out[k] = (int64_t)rem;
        cur = q;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Poseidon: width 24 (rate 20 + capacity 4), 8 full + 22 partial rounds, alpha = 7; round constants and the
// Cauchy MDS matrix come from the Poseidon Grain LFSR (n = 64, t = 24, R_F = 8, R_P = 22).
namespace {
constexpr int W = 24, RATE = 20, CAP = 4, RF = 8, RP = 22;
u64 g_ark[(RF + RP) * W];
u64 g_mds[W * W];
std::once_flag g_once;

struct Grain {
    unsigned char st[80];
    int head;
    int update() {
        int h = head;
        unsigned char nb = st[(h + 62) % 80] ^ st[(h + 51) % 80] ^ st[(h + 38) % 80] ^ st[(h + 23) % 80] ^ st[(h + 13) % 80] ^ st[h];
        st[h] = nb;
        head = (h + 1) % 80;
        return nb;
    }
    void put(int lo, int hi, u64 v) {
        for (int i = hi; i >= lo; i--) { st[i] = v & 1; v >>= 1; }
    }
    void init(u64 bits, u64 width, u64 rf, u64 rp) {
        memset(st, 0, sizeof(st));
        head = 0;
        st[1] = 1;
        put(6, 17, bits); put(18, 29, width); put(30, 39, rf); put(40, 49, rp);
        for (int i = 50; i < 80; i++) st[i] = 1;
        for (int i = 0; i < 160; i++) update();
    }
    u64 word64() {
        u64 v = 0;
        for (int i = 0; i < 64; i++) {
            int nb = update();
            while (!nb) { update(); nb = update(); }
            v = (v << 1) | (u64)update();
        }
        return v;
    }
};

void poseidon_init() {
    Grain g;
    g.init(64, W, RF, RP);
    for (int i = 0; i < (RF + RP) * W; i++) {
        u64 v;
        do v = g.word64(); while (v >= LF_P);
        g_ark[i] = v;
    }
    u64 xs[W], ys[W];
    for (int i = 0; i < W; i++) xs[i] = g.word64() % LF_P;
    for (int i = 0; i < W; i++) ys[i] = g.word64() % LF_P;
    for (int i = 0; i < W; i++)
        for (int j = 0; j < W; j++) g_mds[i * W + j] = fq_inv(fq_add(xs[i], ys[j]));
}
inline u64 sbox(u64 x) {
    u64 x2 = fq_mul(x, x), x3 = fq_mul(x2, x), x4 = fq_mul(x2, x2);
    return fq_mul(x4, x3);
}
typedef unsigned __int128 u128;
// sum_j a[j]*b[j] mod p for n <= 2^32 terms: low and high 64-bit halves of the products are summed separately
inline u64 dot_fq(const u64 *a, const u64 *b, int n) {
    u128 lo = 0, hi = 0;
    for (int j = 0; j < n; j++) {
        u128 pr = (u128)a[j] * b[j];
        lo += (u64)pr;
        hi += (u64)(pr >> 64);
    }
    u64 l = fq_canon(fq_reduce128_loose((u64)lo, (u64)(lo >> 64)));
    u64 h = fq_canon(fq_reduce128_loose((u64)hi, (u64)(hi >> 64)));
    return fq_add(l, fq_mul(h, LF_EPS));  // 2^64 = 2^32 - 1 (mod p)
}

// out = M x for a row-major n x n matrix (ld = row stride): RB rows at a time with independent 192-bit accumulators
// (add / adc / adc per product), one reduction per row; 2^128 = -2^32 (mod p).
#ifndef LF_MATVEC_RB
#define LF_MATVEC_RB 2   /* measured on EPYC 9575F: 0/1/2/3/4 -> 5.2/4.05/4.14/4.10/7.5 us per permutation */
#endif
template <int RB>
inline void matvec_fq_t(const u64 *M, int ld, int n, const u64 *x, u64 *out) {
    int i = 0;
    for (; i + RB <= n; i += RB) {
        u64 lo[RB], mid[RB], hi[RB];
        for (int r = 0; r < RB; r++) lo[r] = mid[r] = hi[r] = 0;
        const u64 *row = M + (size_t)i * ld;
        for (int j = 0; j < n; j++) {
            u64 xv = x[j];
#pragma unroll
            for (int r = 0; r < RB; r++) {
                u128 pr = (u128)xv * row[(size_t)r * ld + j];
                u128 t = (u128)lo[r] + (u64)pr;
                lo[r] = (u64)t;
                t = (u128)mid[r] + (u64)(pr >> 64) + (u64)(t >> 64);
                mid[r] = (u64)t;
                hi[r] += (u64)(t >> 64);
            }
        }
        for (int r = 0; r < RB; r++) out[i + r] = fq_sub(fq_canon(fq_reduce128_loose(lo[r], mid[r])), hi[r] << 32);
    }
    for (; i < n; i++) out[i] = dot_fq(x, M + (size_t)i * ld, n);
}
inline void matvec_fq(const u64 *M, int ld, int n, const u64 *x, u64 *out) {
#if LF_MATVEC_RB == 0
    for (int i = 0; i < n; i++) out[i] = dot_fq(x, M + (size_t)i * ld, n);
#else
    matvec_fq_t<LF_MATVEC_RB>(M, ld, n, x, out);
#endif
}

// Partial rounds through the sparse factorisation M*diag(1,E) = diag(1,E') * [[e00, row],[col, I]] (Poseidon paper,
// appendix on optimised partial rounds): identical output, 47 instead of 576 multiplications per partial round.
struct PartialOpt {
    u64 cst[RP][W];       // round constants pulled through the deferred block-diagonal factor
    u64 e00[RP];
    u64 row[RP][W - 1];
    u64 col[RP][W - 1];
    u64 post[W - 1][W - 1];  // deferred factor applied once after the last partial round
};
PartialOpt g_opt;

bool mat_inv(const u64 *in, u64 *out, int n) {  // Gauss-Jordan over F_p
    std::vector<u64> M((size_t)n * 2 * n, 0);
    for (int r = 0; r < n; r++) {
        for (int c = 0; c < n; c++) M[(size_t)r * 2 * n + c] = in[r * n + c];
        M[(size_t)r * 2 * n + n + r] = 1;
    }
    for (int col = 0; col < n; col++) {
        int piv = -1;
        for (int r = col; r < n; r++)
            if (M[(size_t)r * 2 * n + col]) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != col)
            for (int c = 0; c < 2 * n; c++) std::swap(M[(size_t)piv * 2 * n + c], M[(size_t)col * 2 * n + c]);
        u64 inv = fq_inv(M[(size_t)col * 2 * n + col]);
        for (int c = 0; c < 2 * n; c++) M[(size_t)col * 2 * n + c] = fq_mul(M[(size_t)col * 2 * n + c], inv);
        for (int r = 0; r < n; r++) {
            u64 f = M[(size_t)r * 2 * n + col];
            if (r == col || !f) continue;
            for (int c = 0; c < 2 * n; c++) M[(size_t)r * 2 * n + c] = fq_sub(M[(size_t)r * 2 * n + c], fq_mul(f, M[(size_t)col * 2 * n + c]));
        }
    }
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) out[r * n + c] = M[(size_t)r * 2 * n + n + c];
    return true;
}

void partial_opt_init() {
    const int n = W - 1;
    std::vector<u64> Eprev((size_t)n * n, 0), EprevInv((size_t)n * n, 0), eff((size_t)W * W), Eh((size_t)n * n), Ei((size_t)n * n);
    for (int i = 0; i < n; i++) Eprev[(size_t)i * n + i] = EprevInv[(size_t)i * n + i] = 1;
    for (int r = 0; r < RP; r++) {
        const u64 *c = g_ark + (size_t)(RF / 2 + r) * W;
        // constants: c' = diag(1, Eprev^-1) c
        g_opt.cst[r][0] = c[0];
        for (int i = 0; i < n; i++) g_opt.cst[r][1 + i] = dot_fq(&EprevInv[(size_t)i * n], c + 1, n);
        // eff = M * diag(1, Eprev)
        for (int i = 0; i < W; i++) {
            eff[(size_t)i * W] = g_mds[i * W];
            for (int j = 0; j < n; j++) {
                u64 acc = 0;
                for (int k = 0; k < n; k++) acc = fq_add(acc, fq_mul(g_mds[i * W + 1 + k], Eprev[(size_t)k * n + j]));
                eff[(size_t)i * W + 1 + j] = acc;
            }
        }
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Eh[(size_t)i * n + j] = eff[(size_t)(1 + i) * W + 1 + j];
        if (!mat_inv(Eh.data(), Ei.data(), n)) abort();
        g_opt.e00[r] = eff[0];
        for (int j = 0; j < n; j++) g_opt.row[r][j] = eff[1 + j];
        for (int i = 0; i < n; i++) {
            u64 acc = 0;
            for (int k = 0; k < n; k++) acc = fq_add(acc, fq_mul(Ei[(size_t)i * n + k], eff[(size_t)(1 + k) * W]));
            g_opt.col[r][i] = acc;
        }
        Eprev = Eh;
        EprevInv = Ei;
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) g_opt.post[i][j] = Eprev[(size_t)i * n + j];
}

// AVX-512 IFMA lanes (lf_poseidon_simd.cc) when the CPU has them; LF_POSEIDON_SCALAR=1 keeps the scalar path
bool g_simd = false;
void init_all() {
    poseidon_init();
    partial_opt_init();
    if (psimd::supported() && !getenv("LF_POSEIDON_SCALAR")) {
        psimd::build(g_ark, g_mds, &g_opt.cst[0][0], g_opt.e00, &g_opt.row[0][0], &g_opt.col[0][0], &g_opt.post[0][0]);
        g_simd = true;
    }
}

inline void full_round(u64 st[W], const u64 *ark) {
    u64 nw[W];
    for (int i = 0; i < W; i++) st[i] = sbox(fq_add(st[i], ark[i]));
    matvec_fq(g_mds, W, W, st, nw);
    memcpy(st, nw, sizeof(nw));
}
}  // namespace

void Transcript::params(const u64 **ark, const u64 **mds) {
    std::call_once(g_once, init_all);
    *ark = g_ark;
    *mds = g_mds;
}

// plain definition (arkworks PoseidonSponge::permute): used by the self-test
void Transcript::permute_plain(u64 st[24]) {
    std::call_once(g_once, init_all);
    u64 nw[W];
    for (int r = 0; r < RF + RP; r++) {
        const u64 *ark = g_ark + r * W;
        bool full = r < RF / 2 || r >= RF / 2 + RP;
        for (int i = 0; i < W; i++) st[i] = fq_add(st[i], ark[i]);
        if (full) for (int i = 0; i < W; i++) st[i] = sbox(st[i]);
        else st[0] = sbox(st[0]);
        for (int i = 0; i < W; i++) nw[i] = dot_fq(st, g_mds + i * W, W);
        memcpy(st, nw, sizeof(nw));
    }
}

void Transcript::permute(u64 st[24]) {
    std::call_once(g_once, init_all);
    if (g_simd) { psimd::permute(st); return; }
    permute_scalar(st);
}
void Transcript::permute_scalar(u64 st[24]) {
    std::call_once(g_once, init_all);
    for (int r = 0; r < RF / 2; r++) full_round(st, g_ark + r * W);
    for (int r = 0; r < RP; r++) {
        for (int i = 0; i < W; i++) st[i] = fq_add(st[i], g_opt.cst[r][i]);
        st[0] = sbox(st[0]);
        u64 x0 = st[0];
        // y0 = e00*x0 + row.x[1..] ; y_i = col_i*x0 + x_i
        u128 lo = (u128)g_opt.e00[r] * x0, hi = 0;
        hi = (u64)(lo >> 64);
        lo = (u64)lo;
        for (int j = 0; j < W - 1; j++) {
            u128 pr = (u128)g_opt.row[r][j] * st[1 + j];
            lo += (u64)pr;
            hi += (u64)(pr >> 64);
        }
        u64 l = fq_canon(fq_reduce128_loose((u64)lo, (u64)(lo >> 64)));
        u64 h = fq_canon(fq_reduce128_loose((u64)hi, (u64)(hi >> 64)));
        for (int i = 0; i < W - 1; i++) st[1 + i] = fq_add(st[1 + i], fq_mul(g_opt.col[r][i], x0));
        st[0] = fq_add(l, fq_mul(h, LF_EPS));
    }
    {   // deferred block-diagonal factor
        u64 nw[W - 1];
        matvec_fq(&g_opt.post[0][0], W - 1, W - 1, st + 1, nw);
        memcpy(st + 1, nw, sizeof(nw));
    }
    for (int r = RF / 2 + RP; r < RF + RP; r++) full_round(st, g_ark + r * W);
}

Transcript::Transcript() : squeezing_(false), idx_(0) {
    std::call_once(g_once, init_all);
    memset(st_, 0, sizeof(st_));
}

void Transcript::absorb_fq(const u64 *x, size_t n) {
    if (!n) return;
    int idx;
    if (!squeezing_) {
        idx = idx_;
        if (idx == RATE) { permute(st_); idx = 0; }
    } else {
        permute(st_);
        idx = 0;
    }
    for (;;) {
        if ((size_t)idx + n <= (size_t)RATE) {
            for (size_t i = 0; i < n; i++) st_[CAP + idx + i] = fq_add(st_[CAP + idx + i], x[i]);
            squeezing_ = false;
            idx_ = idx + (int)n;
            return;
        }
        size_t take = RATE - idx;
        for (size_t i = 0; i < take; i++) st_[CAP + idx + i] = fq_add(st_[CAP + idx + i], x[i]);
        permute(st_);
        x += take; n -= take; idx = 0;
    }
}

void Transcript::squeeze(u64 *out, size_t n) {
    int idx;
    if (!squeezing_) { permute(st_); idx = 0; }
    else {
        idx = idx_;
        if (idx == RATE) { permute(st_); idx = 0; }
    }
    for (;;) {
        if ((size_t)idx + n <= (size_t)RATE) {
            memcpy(out, st_ + CAP + idx, n * sizeof(u64));
            squeezing_ = true;
            idx_ = idx + (int)n;
            return;
        }
        size_t take = RATE - idx;
        memcpy(out, st_ + CAP + idx, take * sizeof(u64));
        if (n != (size_t)RATE) permute(st_);
        out += take; n -= take; idx = 0;
    }
}

static void basis3(const u64 *M, const u64 *v, u64 *o) {   // o = M v over F_p, 3x3
    for (int i = 0; i < 3; i++) o[i] = fq_add(fq_add(fq_mul(M[3 * i], v[0]), fq_mul(M[3 * i + 1], v[1])), fq_mul(M[3 * i + 2], v[2]));
}
void Transcript::absorb_ring(const u64 *e, size_t count) {
    if (!bT_) {
        for (size_t i = 0; i < count; i++) absorb_fq(e + 24 * i, 24);
        return;
    }
    for (size_t i = 0; i < count; i++) {   // internal -> external basis, slot by slot
        u64 x[24];
        for (int sl = 0; sl < 8; sl++) basis3(bT_, e + 24 * i + 3 * sl, x + 3 * sl);
        absorb_fq(x, 24);
    }
}
void Transcript::absorb_label(const char *s) {
    unsigned __int128 v = 0;
    for (; *s; s++) v = ((v << 8) | (unsigned char)*s) % LF_P;
    absorb_u64_as_ring((u64)v);
}
void Transcript::absorb_fq3_as_ring(Fq3 c) {
    u64 e[24];
    HostRing::from_fq3(c, e);
    absorb_ring(e, 1);
}
void Transcript::absorb_u64_as_ring(u64 v) {
    u64 e[24];
    HostRing::from_u64(v, e);
    absorb_ring(e, 1);
}
Fq3 Transcript::get_challenge() {
    u64 c[3];
    squeeze(c, 3);
    absorb_fq(c, 3);          // the squeezed words are the EXTERNAL coordinates and go back as they are
    if (bTi_) { u64 o[3]; basis3(bTi_, c, o); return fq3_make(o[0], o[1], o[2]); }
    return fq3_make(c[0], c[1], c[2]);
}
void Transcript::get_short_challenge(u64 out[24]) {
    // squeeze_bytes(18): 3 field elements, 7 low little-endian bytes each; then 24 six-bit fields - 32
    u64 e[3];
    squeeze(e, 3);
    unsigned char bs[21];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 7; j++) bs[7 * i + j] = (unsigned char)(e[i] >> (8 * j));
    for (int g = 0; g < 6; g++) {
        u32 w = (u32)bs[3 * g] | ((u32)bs[3 * g + 1] << 8) | ((u32)bs[3 * g + 2] << 16);
        for (int j = 0; j < 4; j++) out[4 * g + j] = fq_from_i64((int64_t)((w >> (6 * j)) & 63) - 32);
    }
}

}  // namespace lf

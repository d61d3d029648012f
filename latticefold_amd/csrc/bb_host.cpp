// bb_host.cpp -- BabyBearRingNTT: ring tables, small host ring ops, Poseidon, transcript (see bb_host.h).
#include "bb_host.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <utility>

#include "lf_host.h"   // lf::Transcript::params: the Grain-generated 64-bit Poseidon table shared by both rings
#if defined(__AVX2__) && !defined(__HIP_DEVICE_COMPILE__)
#include "bb_poseidon_simd.h"
#define BB_POSEIDON_SIMD 1
#endif

namespace lfbb {

namespace simd512 {   // bb_poseidon_avx512.cc (host-only translation unit, entered after a cpuid check)
bool supported();
void build(const u64 *ark, const u64 *mds, const u64 *cst, const u64 *e00, const u64 *row, const u64 *col, const u64 *post);
void permute(u64 st[24]);
}  // namespace simd512

u64 hpow(u64 a, u64 e) {
    u64 r = 1;
    a %= BB_P;
    while (e) {
        if (e & 1) r = hmul(r, a);
        a = hmul(a, a);
        e >>= 1;
    }
    return r;
}

static H9 h9_zero() { H9 r; memset(&r, 0, sizeof(r)); return r; }
static H9 h9_one() { H9 r = h9_zero(); r.c[0] = 1; return r; }
// (inputs canonical: a product is < p^2 < 2^62, so four of them fit a 64-bit word -- the 81 products of an F_{p^9} product take 27 reductions instead of 81;
// this is the host's ring arithmetic on proof-sized data: the folded instance, the challenge powers, the message completion of the split rounds)
static H9 h9_mul_nu(const H9 &a, const H9 &b, u64 nu) {
    u64 lo[TAU], hi[TAU];
    for (int k = 0; k < TAU; k++) {
        u64 acc = 0, tot = 0;
        int cnt = 0;
        for (int i = 0; i <= k; i++) {
            acc += a.c[i] * b.c[k - i];
            if (++cnt == 4) { tot += acc % BB_P; acc = 0; cnt = 0; }
        }
        lo[k] = (tot + acc % BB_P) % BB_P;
        acc = 0; tot = 0; cnt = 0;
        for (int i = k + 1; i < TAU; i++) {
            acc += a.c[i] * b.c[k + TAU - i];
            if (++cnt == 4) { tot += acc % BB_P; acc = 0; cnt = 0; }
        }
        hi[k] = (tot + acc % BB_P) % BB_P;
    }
    H9 r;
    for (int k = 0; k < TAU; k++) r.c[k] = hadd(lo[k], hmul(nu, hi[k]));
    return r;
}

void bb_default_ring(u64 *nonres, u64 *y) {
    // F_{p^9} = F_p[Y]/(Y^9 - 2): 2 is a non-cube mod p (3 | p-1, 9 does not), so the binomial is irreducible, and multiplying by
    // the non-residue is a doubling.  zeta = first g^((p-1)/24), g = 2,3,.., of exact order 24; slot e (ascending over (Z/24)^*)
    // maps X -> c Y^g with g in {1,2} the class for which zeta^e / 2^g is a cube, and c its 9th root inside the cube subgroup
    // (x -> x^9 is a bijection there: the subgroup has order (p-1)/3 = 5 * 2^27, coprime to 9).
    static const int E[8] = {1, 5, 7, 11, 13, 17, 19, 23};
    u64 zeta = 0;
    for (u64 g = 2;; g++) {
        u64 z = hpow(g, (BB_P - 1) / 24);
        if (hpow(z, 12) != 1 && hpow(z, 8) != 1) { zeta = z; break; }
    }
    *nonres = 2;
    const u64 sub = (BB_P - 1) / 3;
    u64 e9 = 0;   // 9^-1 mod (p-1)/3
    for (u64 k = 1; k < 9; k++)
        if ((k * sub + 1) % 9 == 0) { e9 = (k * sub + 1) / 9; break; }
    memset(y, 0, 8 * TAU * sizeof(u64));
    for (int k = 0; k < 8; k++) {
        u64 ze = hpow(zeta, (u64)E[k]);
        for (int g = 1; g <= 2; g++) {
            u64 w = hmul(ze, hinv(hpow(2, (u64)g)));
            if (hpow(w, sub) != 1) continue;   // not a cube
            y[TAU * k + g] = hpow(w, e9);
            break;
        }
    }
}

int bb_build_tables(u64 nonres, const u64 *y, BbTables &T) {
    memset(&T, 0, sizeof(T));
    T.nu = nonres % BB_P;
    u64 zeta[8];
    for (int k = 0; k < 8; k++) {
        for (int c = 0; c < TAU; c++) T.y[k].c[c] = y[TAU * k + c] % BB_P;
        H9 p = h9_one();
        for (int c = 0; c < D; c++) {
            T.ypow[k][c] = p;
            p = h9_mul_nu(p, T.y[k], T.nu);
        }
        const H9 &y9 = T.ypow[k][TAU];
        for (int c = 1; c < TAU; c++)
            if (y9.c[c]) return -1;
        zeta[k] = y9.c[0];
        u64 z4 = hpow(zeta[k], 4), z8 = hmul(z4, z4);
        if (hadd(hsub(z8, z4), 1) != 0) return -1;   // root of Phi_24
        for (int j = 0; j < k; j++)
            if (zeta[j] == zeta[k]) return -1;
    }
    {   // dense inverse (Gauss-Jordan over F_p)
        std::vector<u64> M((size_t)D * 2 * D, 0);
        auto at = [&](int r, int c) -> u64 & { return M[(size_t)r * 2 * D + c]; };
        for (int k = 0; k < 8; k++)
            for (int c = 0; c < D; c++)
                for (int q = 0; q < TAU; q++) at(TAU * k + q, c) = T.ypow[k][c].c[q];
        for (int r = 0; r < D; r++) at(r, D + r) = 1;
        for (int col = 0; col < D; col++) {
            int piv = -1;
            for (int r = col; r < D; r++)
                if (at(r, col)) { piv = r; break; }
            if (piv < 0) return -1;
            if (piv != col)
                for (int c = 0; c < 2 * D; c++) std::swap(at(piv, c), at(col, c));
            u64 inv = hinv(at(col, col));
            for (int c = 0; c < 2 * D; c++) at(col, c) = hmul(at(col, c), inv);
            for (int r = 0; r < D; r++) {
                u64 f = at(r, col);
                if (r == col || !f) continue;
                for (int c = 0; c < 2 * D; c++) at(r, c) = hsub(at(r, c), hmul(f, at(col, c)));
            }
        }
        for (int r = 0; r < D; r++)
            for (int c = 0; c < D; c++) T.icrt[r][c] = at(r, D + c);
    }
    u64 w = zeta[0];
    T.w1 = w; T.w2 = hpow(w, 2); T.w4 = hpow(w, 4); T.w5 = hpow(w, 5); T.w7 = hpow(w, 7); T.w10 = hpow(w, 10); T.w11 = hpow(w, 11);
    static const int ENAT[8] = {1, 13, 7, 19, 5, 17, 11, 23};   // butterfly output order (exponent of omega)
    for (int p = 0; p < 8; p++) {
        u64 root = hpow(w, (u64)ENAT[p]);
        int slot = -1;
        for (int k = 0; k < 8; k++)
            if (zeta[k] == root) slot = k;
        if (slot < 0) return -1;
        T.slot_of_pos[p] = slot;
        for (int r = 0; r < TAU; r++) {   // y^r must be a monomial c * Y^m
            const H9 &yr = T.ypow[slot][r];
            int m = -1;
            for (int c = 0; c < TAU; c++)
                if (yr.c[c]) { if (m >= 0) return -1; m = c; }
            if (m < 0) return -1;
            T.pos[r][p] = m;
            T.tw[r][p] = yr.c[m];
        }
        for (int r = 0; r < TAU; r++)      // the 9 positions must be a permutation
            for (int r2 = 0; r2 < r; r2++)
                if (T.pos[r][p] == T.pos[r2][p]) return -1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
H9 BbHostRing::mul9(const H9 &a, const H9 &b) const { return h9_mul_nu(a, b, T.nu); }
void BbHostRing::crt(const u64 *a, u64 *out) const {
    u64 r[D];
    for (int k = 0; k < 8; k++)
        for (int q = 0; q < TAU; q++) {
            u64 acc = 0;
            for (int c = 0; c < D; c++)
                if (a[c]) acc = hadd(acc, hmul(T.ypow[k][c].c[q], a[c] % BB_P));
            r[TAU * k + q] = acc;
        }
    memcpy(out, r, sizeof(r));
}
void BbHostRing::icrt(const u64 *x, u64 *out) const {
    u64 r[D];
    for (int i = 0; i < D; i++) {
        u64 acc = 0;
        for (int j = 0; j < D; j++)
            if (x[j]) acc = hadd(acc, hmul(T.icrt[i][j], x[j]));
        r[i] = acc;
    }
    memcpy(out, r, sizeof(r));
}
static H9 ldh(const u64 *e, int k) { H9 r; memcpy(r.c, e + TAU * k, sizeof(r.c)); return r; }
static void sth(u64 *e, int k, const H9 &v) { memcpy(e + TAU * k, v.c, sizeof(v.c)); }
void BbHostRing::mul_ntt(const u64 *a, const u64 *b, u64 *out) const {
    u64 r[D];
    for (int k = 0; k < 8; k++) sth(r, k, mul9(ldh(a, k), ldh(b, k)));
    memcpy(out, r, sizeof(r));
}
void BbHostRing::mul_h9(const u64 *a, const H9 &s, u64 *out) const {
    u64 r[D];
    for (int k = 0; k < 8; k++) sth(r, k, mul9(ldh(a, k), s));
    memcpy(out, r, sizeof(r));
}
void BbHostRing::add(const u64 *a, const u64 *b, u64 *out) { for (int i = 0; i < D; i++) out[i] = hadd(a[i], b[i]); }
void BbHostRing::sub(const u64 *a, const u64 *b, u64 *out) { for (int i = 0; i < D; i++) out[i] = hsub(a[i], b[i]); }
void BbHostRing::from_u64(u64 v, u64 *out) {
    memset(out, 0, D * sizeof(u64));
    for (int k = 0; k < 8; k++) out[TAU * k] = v % BB_P;
}
void BbHostRing::from_h9(const H9 &s, u64 *out) { for (int k = 0; k < 8; k++) sth(out, k, s); }

// stark_rings::balanced_decomposition as recollected (convention is DATA-level "unpinned"): centred lift,
// truncating remainder, |rem| <= b/2 kept, otherwise rem -+ b with carry +-1, zero padded.
void bb_balanced_digits(u64 v, u64 base, unsigned digits, int64_t *out, int mode) {
    int64_t b = (int64_t)base, half = b / 2;
    int64_t cur = v <= (BB_P - 1) / 2 ? (int64_t)v : (int64_t)v - (int64_t)BB_P;
    for (unsigned k = 0; k < digits; k++) {
        int64_t rem = cur % b, q = cur / b;
        if (mode == 1 && base > 2) {   // floor rule: digits in [-base/2, base/2)
            if (rem < 0) rem += b;
            if (rem >= half) rem -= b;
            q = (cur - rem) / b;
        } else {
            int64_t ar = rem < 0 ? -rem : rem;
            if (ar > half) {
                if (rem < 0) { rem += b; q -= 1; }
                else { rem -= b; q += 1; }
            }
        }
        out[k] = rem;
        cur = q;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Poseidon over BabyBear: the reference table (rings/poseidon/babybear.rs:7-1425) holds the SAME 64-bit literals as the
// Goldilocks table, embedded with Fq::from(i128) -- i.e. the Grain-generated Goldilocks constants reduced mod p_BB.
namespace {
constexpr int W = 24, RATE = 20, CAP = 4, RF = 8, RP = 22;
u64 g_ark[(RF + RP) * W];
u64 g_mds[W * W];
std::once_flag g_once;
struct PartialOpt {
    u64 cst[RP][W];
    u64 e00[RP];
    u64 row[RP][W - 1];
    u64 col[RP][W - 1];
    u64 post[W - 1][W - 1];
};
PartialOpt g_opt;

inline u64 sbox(u64 x) {
    u64 x2 = hmul(x, x), x3 = hmul(x2, x), x4 = hmul(x2, x2);
    return hmul(x4, x3);
}
// sum of n <= 32 products of 31-bit values: four at a time fit a u64
inline u64 dot(const u64 *a, const u64 *b, int n) {
    u64 acc = 0;
    int j = 0;
    for (; j + 4 <= n; j += 4) acc = (acc + (a[j] * b[j] + a[j + 1] * b[j + 1] + a[j + 2] * b[j + 2] + a[j + 3] * b[j + 3]) % BB_P);
    for (; j < n; j++) acc += a[j] * b[j] % BB_P;
    return acc % BB_P;
}
// out = M x for an n x n matrix stored TRANSPOSED (MT[j][i] = M[i][j], row stride 24) on 31-bit words: the 64-bit products are
// split into 32-bit halves that accumulate without overflow (n <= 24 terms), so the inner loop is a plain vpmuludq/vpaddq
// stream (auto-vectorised: AVX2 under -march=x86-64-v3); one reduction per output.
constexpr u64 R32 = (1ull << 32) % BB_P;
template <int N>
inline void matvec_t(const u32 (*MT)[24], const u64 *x, u64 *out) {
    u64 lo[24] = {0}, hi[24] = {0};
    for (int j = 0; j < N; j++) {
        const u64 xj = (u32)x[j];
        const u32 *row = MT[j];
#pragma clang loop vectorize(enable) interleave(enable)
        for (int i = 0; i < 24; i++) {
            u64 pr = xj * (u64)row[i];
            lo[i] += pr & 0xffffffffull;
            hi[i] += pr >> 32;
        }
    }
    for (int i = 0; i < N; i++) out[i] = ((hi[i] % BB_P) * R32 + lo[i]) % BB_P;
}
#ifdef BB_POSEIDON_SIMD
simd::Tables g_simd;
#endif
int g_path = 0;           // 0 scalar, 1 AVX2 (bb_poseidon_simd.h), 2 AVX-512 IFMA (bb_poseidon_avx512.cc)
u32 g_mdsT[24][24];       // MDS transposed
u32 g_postT[24][24];      // deferred factor of the sparse partial rounds, transposed (23 x 23 used)

bool mat_inv(const u64 *in, u64 *out, int n) {
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
        u64 inv = hinv(M[(size_t)col * 2 * n + col]);
        for (int c = 0; c < 2 * n; c++) M[(size_t)col * 2 * n + c] = hmul(M[(size_t)col * 2 * n + c], inv);
        for (int r = 0; r < n; r++) {
            u64 f = M[(size_t)r * 2 * n + col];
            if (r == col || !f) continue;
            for (int c = 0; c < 2 * n; c++) M[(size_t)r * 2 * n + c] = hsub(M[(size_t)r * 2 * n + c], hmul(f, M[(size_t)col * 2 * n + c]));
        }
    }
    for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) out[r * n + c] = M[(size_t)r * 2 * n + n + c];
    return true;
}
// sparse factorisation of the partial rounds (Poseidon paper, optimised partial rounds): M*diag(1,E) = diag(1,E')*[[e00,row],[col,I]]
void init_all() {
    const u64 *ga, *gm;
    lf::Transcript::params(&ga, &gm);
    for (int i = 0; i < (RF + RP) * W; i++) g_ark[i] = ga[i] % BB_P;
    for (int i = 0; i < W * W; i++) g_mds[i] = gm[i] % BB_P;
    const int n = W - 1;
    std::vector<u64> Eprev((size_t)n * n, 0), EprevInv((size_t)n * n, 0), eff((size_t)W * W), Eh((size_t)n * n), Ei((size_t)n * n);
    for (int i = 0; i < n; i++) Eprev[(size_t)i * n + i] = EprevInv[(size_t)i * n + i] = 1;
    for (int r = 0; r < RP; r++) {
        const u64 *c = g_ark + (size_t)(RF / 2 + r) * W;
        g_opt.cst[r][0] = c[0];
        for (int i = 0; i < n; i++) g_opt.cst[r][1 + i] = dot(&EprevInv[(size_t)i * n], c + 1, n);
        for (int i = 0; i < W; i++) {
            eff[(size_t)i * W] = g_mds[i * W];
            for (int j = 0; j < n; j++) {
                u64 acc = 0;
                for (int k = 0; k < n; k++) acc = hadd(acc, hmul(g_mds[i * W + 1 + k], Eprev[(size_t)k * n + j]));
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
            for (int k = 0; k < n; k++) acc = hadd(acc, hmul(Ei[(size_t)i * n + k], eff[(size_t)(1 + k) * W]));
            g_opt.col[r][i] = acc;
        }
        Eprev = Eh;
        EprevInv = Ei;
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) g_opt.post[i][j] = Eprev[(size_t)i * n + j];
    memset(g_mdsT, 0, sizeof(g_mdsT));
    memset(g_postT, 0, sizeof(g_postT));
    for (int i = 0; i < W; i++)
        for (int j = 0; j < W; j++) g_mdsT[j][i] = (u32)g_mds[i * W + j];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) g_postT[j][i] = (u32)g_opt.post[i][j];
    if (getenv("LF_POSEIDON_SCALAR")) return;
#ifdef BB_POSEIDON_SIMD
    simd::build_tables(g_simd, g_ark, g_mds, g_opt.cst, g_opt.e00, g_opt.row, g_opt.col, g_opt.post);
    g_path = 1;
#endif
    if (simd512::supported() && !getenv("LF_POSEIDON_AVX2")) {
        simd512::build(g_ark, g_mds, &g_opt.cst[0][0], g_opt.e00, &g_opt.row[0][0], &g_opt.col[0][0], &g_opt.post[0][0]);
        g_path = 2;
    }
}
inline void full_round(u64 st[W], const u64 *ark) {
    u64 nw[W];
    for (int i = 0; i < W; i++) st[i] = sbox(hadd(st[i], ark[i]));
    matvec_t<W>(g_mdsT, st, nw);
    memcpy(st, nw, sizeof(nw));
}
}  // namespace

void BbTranscript::params(const u64 **ark, const u64 **mds) {
    std::call_once(g_once, init_all);
    *ark = g_ark;
    *mds = g_mds;
}
void BbTranscript::permute_plain(u64 st[24]) {
    std::call_once(g_once, init_all);
    u64 nw[W];
    for (int r = 0; r < RF + RP; r++) {
        const u64 *ark = g_ark + r * W;
        bool full = r < RF / 2 || r >= RF / 2 + RP;
        for (int i = 0; i < W; i++) st[i] = hadd(st[i], ark[i]);
        if (full) for (int i = 0; i < W; i++) st[i] = sbox(st[i]);
        else st[0] = sbox(st[0]);
        for (int i = 0; i < W; i++) nw[i] = dot(st, g_mds + i * W, W);
        memcpy(st, nw, sizeof(nw));
    }
}
void BbTranscript::permute(u64 st[24]) {
    std::call_once(g_once, init_all);
    if (g_path == 2) { simd512::permute(st); return; }
#ifdef BB_POSEIDON_SIMD
    if (g_path == 1) { simd::permute(g_simd, st); return; }
#endif
    permute_scalar(st);
}
void BbTranscript::permute_scalar(u64 st[24]) {
    std::call_once(g_once, init_all);
    for (int r = 0; r < RF / 2; r++) full_round(st, g_ark + r * W);
    for (int r = 0; r < RP; r++) {
        for (int i = 0; i < W; i++) st[i] = hadd(st[i], g_opt.cst[r][i]);
        st[0] = sbox(st[0]);
        u64 x0 = st[0];
        u64 y0 = (hmul(g_opt.e00[r], x0) + dot(g_opt.row[r], st + 1, W - 1)) % BB_P;
        for (int i = 0; i < W - 1; i++) st[1 + i] = (st[1 + i] + g_opt.col[r][i] * x0) % BB_P;
        st[0] = y0;
    }
    {
        u64 nw[W];
        matvec_t<W - 1>(g_postT, st + 1, nw);
        memcpy(st + 1, nw, (W - 1) * sizeof(u64));
    }
    for (int r = RF / 2 + RP; r < RF + RP; r++) full_round(st, g_ark + r * W);
}

BbTranscript::BbTranscript() : squeezing_(false), idx_(0) {
    std::call_once(g_once, init_all);
    memset(st_, 0, sizeof(st_));
}
void BbTranscript::absorb_fq(const u64 *x, size_t n) {
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
            for (size_t i = 0; i < n; i++) st_[CAP + idx + i] = hadd(st_[CAP + idx + i], x[i] % BB_P);
            squeezing_ = false;
            idx_ = idx + (int)n;
            return;
        }
        size_t take = RATE - idx;
        for (size_t i = 0; i < take; i++) st_[CAP + idx + i] = hadd(st_[CAP + idx + i], x[i] % BB_P);
        permute(st_);
        x += take; n -= take; idx = 0;
    }
}
void BbTranscript::squeeze(u64 *out, size_t n) {
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
static void basis9(const u64 *M, const u64 *v, u64 *o) {   // o = M v over F_p, 9x9 (words < 2^31)
    for (int i = 0; i < TAU; i++) {
        u64 acc = 0;
        for (int j = 0; j < TAU; j++) acc += (M[TAU * i + j] * v[j]) % BB_P;
        o[i] = acc % BB_P;
    }
}
void BbTranscript::absorb_ring(const u64 *e, size_t count) {
    if (!bT_) {
        for (size_t i = 0; i < count; i++) absorb_fq(e + (size_t)D * i, D);
        return;
    }
    for (size_t i = 0; i < count; i++) {   // internal -> external basis, slot by slot
        u64 x[D];
        for (int sl = 0; sl < 8; sl++) basis9(bT_, e + (size_t)D * i + TAU * sl, x + TAU * sl);
        absorb_fq(x, D);
    }
}
void BbTranscript::absorb_label(const char *s) {
    u64 v = 0;
    for (; *s; s++) v = ((v << 8) | (unsigned char)*s) % BB_P;
    absorb_u64_as_ring(v);
}
void BbTranscript::absorb_h9_as_ring(const H9 &c) {
    u64 e[D];
    BbHostRing::from_h9(c, e);
    absorb_ring(e, 1);
}
void BbTranscript::absorb_u64_as_ring(u64 v) {
    u64 e[D];
    BbHostRing::from_u64(v, e);
    absorb_ring(e, 1);
}
H9 BbTranscript::get_challenge() {
    H9 c;
    squeeze(c.c, TAU);
    absorb_fq(c.c, TAU);      // the squeezed words are the EXTERNAL coordinates and go back as they are
    if (bTi_) { H9 o; basis9(bTi_, c.c, o.c); return o; }
    return c;
}
void BbTranscript::get_short_challenge(u64 out[D]) {
    // squeeze_bytes(18) of the arkworks-0.4 PoseidonSponge: usable bytes per element = (31 - 1) / 8 = 3 -> 6 elements,
    // 3 low little-endian bytes each; then BabyBearChallengeSet (rings/babybear.rs:36-68): 24 six-bit fields - 32
    u64 e[6];
    squeeze(e, 6);
    unsigned char bs[18];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 3; j++) bs[3 * i + j] = (unsigned char)(e[i] >> (8 * j));
    memset(out, 0, D * sizeof(u64));
    for (int g = 0; g < 6; g++) {
        u32 w = (u32)bs[3 * g] | ((u32)bs[3 * g + 1] << 8) | ((u32)bs[3 * g + 2] << 16);
        for (int j = 0; j < 4; j++) out[4 * g + j] = hfrom_i64((int64_t)((w >> (6 * j)) & 63) - 32);
    }
}

}  // namespace lfbb

// lfp_protocol.cpp -- the transcript-driven part of the LatticeFold+ slice (include/lfplus.h) on the Frog ring:
//   PoseidonTranscript<RqPoly>          crates/latticefold-plus/src/transcript.rs:20-78 (host; parameters rings/poseidon/frog.rs)
//   In::set_check / Out::verify         src/setchk.rs:65-262 / 266-340   (prover on the GPU: lfp_rgchk.hip; verifier on the host)
//   Rg::range_check / Dcom::verify      src/rgchk.rs:81-186 / 193-258
// The Fiat-Shamir transcript stays on the host (one width-24 Poseidon permutation per 20 absorbed words; a sumcheck round moves four words
// up and one down); every table and every evaluation lives on the device.  No CPU fallback: the provers return LFPLUS_E_NO_DEVICE / _HIP.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include "lf_host.h"
#include "lfp_ctx.h"
#include "lfp_poseidon_simd.h"
static constexpr uint32_t LFP_HOST_SUM_BLOCKS = 256;   // block partials the host adds per round; rounds with more workgroups add theirs on the device (launch_reduce)

// LFPLUS_TIMELINE=1: wall-clock marks of the protocol stages on stderr (the stream is drained at every mark, so the stages do not overlap)
struct LfpTl {
    bool on = getenv("LFPLUS_TIMELINE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(lfplus_ctx *c, const char *what) {
        if (!on) return;
        if (c && c->st) (void)hipStreamSynchronize(c->st);
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[lfplus] %-34s +%8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count());
        t0 = t;
    }
};
static LfpTl g_tl;
#define LFP_MARK(c, w) g_tl.mark(c, w)
namespace {
typedef unsigned __int128 u128;
constexpr u64 P = lfp::P;
constexpr int D = 16, W = 24, RATE = 20, CAP = 4, RF = 8, RP = 22;
inline u64 fadd(u64 a, u64 b) { u128 s = (u128)a + b; return (u64)(s >= P ? s - P : s); }
inline u64 fsub(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
inline u64 fmul(u64 a, u64 b) { return (u64)(((u128)a * b) % P); }
u64 fpow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = fmul(r, a); a = fmul(a, a); e >>= 1; } return r; }
inline u64 to_mont(u64 a) { return (u64)((((u128)a) << 64) % P); }
const u64 RINV = fpow(to_mont(1), P - 2);   // 2^-64 mod p
inline u64 from_mont(u64 a) { return fmul(a, RINV); }

// The reference's Frog table (rings/poseidon/frog.rs:7-1425) is the Grain-LFSR table of the 64-bit Goldilocks prime embedded with
// Fq::from(i128): the generator of lf_host.cpp, reduced mod p_frog (pinned by the reference's checksums in tests/golden/kats.json)
struct Params {
    u64 ark[(RF + RP) * W], mds[W * W];
    Params() {
        const u64 *a, *m;
        lf::Transcript::params(&a, &m);
        for (int i = 0; i < (RF + RP) * W; i++) ark[i] = a[i] % P;
        for (int i = 0; i < W * W; i++) mds[i] = m[i] % P;
    }
};
const Params &params() { static const Params p; return p; }
inline u64 pow7(u64 x) { u64 x2 = fmul(x, x), x4 = fmul(x2, x2); return fmul(fmul(x4, x2), x); }
// the definition (ark-crypto-primitives PoseidonSponge::permute): the self-test reference of permute()
void permute_plain(u64 *st) {
    const Params &pp = params();
    u64 nw[W];
    for (int r = 0; r < RF + RP; r++) {
        for (int i = 0; i < W; i++) st[i] = fadd(st[i], pp.ark[r * W + i]);
        if (r < RF / 2 || r >= RF / 2 + RP) for (int i = 0; i < W; i++) st[i] = pow7(st[i]);
        else st[0] = pow7(st[0]);
        for (int i = 0; i < W; i++) {
            u64 acc = 0;
            for (int j = 0; j < W; j++) acc = fadd(acc, fmul(st[j], pp.mds[i * W + j]));
            nw[i] = acc;
        }
        memcpy(st, nw, sizeof(nw));
    }
}

// ---- the permutation as it runs: Montgomery words, lazy row sums, sparse partial rounds ---------------------------------------------------
// (a prove at n = 2^15 makes ~500 permutations; the plain form costs 80 us each on the host and was most of the wall time.)
// Partial rounds through the factorisation M diag(1, E) = diag(1, E') [[e00, row], [col, I]] (Poseidon paper, appendix on optimised partial rounds; the
// same construction as lf_host.cpp uses for the Goldilocks table): identical output, 47 instead of 576 multiplications per partial round.
struct FastPerm {
    u64 pinv, r2;                                  // p^-1 mod 2^64, 2^128 mod p
    u64 ark[(RF + RP) * W], mds[W * W];            // Montgomery
    u64 cst[RP][W], e00[RP], row[RP][W - 1], col[RP][W - 1], post[(W - 1) * (W - 1)];
    inline u64 mm(u64 a, u64 b) const {            // a b 2^-64 mod p
        const u128 t = (u128)a * b;
        const u64 m = (u64)t * pinv, th = (u64)(t >> 64), mh = (u64)(((u128)m * P) >> 64);
        return th >= mh ? th - mh : th + (P - mh);
    }
    // sum_j a[j] b[j] 2^-64 mod p, one reduction per sum
    inline u64 dot(const u64 *a, const u64 *b, int n) const {
        u128 lo = 0, hi = 0;
        for (int j = 0; j < n; j++) { const u128 pr = (u128)a[j] * b[j]; lo += (u64)pr; hi += (u64)(pr >> 64); }
        // value = lo + 2^64 hi: REDC(x) = x 2^-64 is linear, so reduce the two words separately
        return fadd(redc(lo), small(hi));
    }
    static inline u64 small(u128 h) {              // h < 2^70 mod p without a 128-bit division: 2^64 = 2^64 - p (mod p)
        u128 t = (u128)(u64)(h >> 64) * (u64)(0 - P) + (u64)h;
        while (t >= P) t -= P;                     // t < 2^67: at most a few subtractions
        return (u64)t;
    }
    inline u64 redc(u128 t) const {                // t 2^-64 mod p for any t < 2^128
        const u64 m = (u64)t * pinv, th = (u64)(t >> 64), mh = (u64)(((u128)m * P) >> 64);
        u64 r = th >= mh ? th - mh : th + (P - mh);   // th < 2^64 may exceed p
        return r >= P ? r - P : r;
    }
    static bool mat_inv(const u64 *in, u64 *out, int n) {   // Gauss-Jordan over F_p, canonical words
        std::vector<u64> M((size_t)n * 2 * n, 0);
        for (int r = 0; r < n; r++) {
            for (int c = 0; c < n; c++) M[(size_t)r * 2 * n + c] = in[r * n + c];
            M[(size_t)r * 2 * n + n + r] = 1;
        }
        for (int cc = 0; cc < n; cc++) {
            int piv = -1;
            for (int r = cc; r < n; r++) if (M[(size_t)r * 2 * n + cc]) { piv = r; break; }
            if (piv < 0) return false;
            if (piv != cc) for (int c = 0; c < 2 * n; c++) std::swap(M[(size_t)piv * 2 * n + c], M[(size_t)cc * 2 * n + c]);
            const u64 inv = fpow(M[(size_t)cc * 2 * n + cc], P - 2);
            for (int c = 0; c < 2 * n; c++) M[(size_t)cc * 2 * n + c] = fmul(M[(size_t)cc * 2 * n + c], inv);
            for (int r = 0; r < n; r++) {
                const u64 f = M[(size_t)r * 2 * n + cc];
                if (r == cc || !f) continue;
                for (int c = 0; c < 2 * n; c++) M[(size_t)r * 2 * n + c] = fsub(M[(size_t)r * 2 * n + c], fmul(f, M[(size_t)cc * 2 * n + c]));
            }
        }
        for (int r = 0; r < n; r++) for (int c = 0; c < n; c++) out[r * n + c] = M[(size_t)r * 2 * n + n + c];
        return true;
    }
    bool ok = false;
    FastPerm() {
        const Params &pp = params();
        u64 x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - P * x;   // Newton: p^-1 mod 2^64
        pinv = x;
        r2 = to_mont(to_mont(1));
        const int n = W - 1;
        std::vector<u64> Eprev((size_t)n * n, 0), EprevInv((size_t)n * n, 0), eff((size_t)W * W), Eh((size_t)n * n), Ei((size_t)n * n);
        for (int i = 0; i < n; i++) Eprev[(size_t)i * n + i] = EprevInv[(size_t)i * n + i] = 1;
        for (int r = 0; r < RP; r++) {
            const u64 *c = pp.ark + (size_t)(RF / 2 + r) * W;
            cst[r][0] = to_mont(c[0]);
            for (int i = 0; i < n; i++) {               // constants pulled through the deferred factor: c' = diag(1, Eprev^-1) c
                u64 acc = 0;
                for (int k = 0; k < n; k++) acc = fadd(acc, fmul(EprevInv[(size_t)i * n + k], c[1 + k]));
                cst[r][1 + i] = to_mont(acc);
            }
            for (int i = 0; i < W; i++) {               // eff = M diag(1, Eprev)
                eff[(size_t)i * W] = pp.mds[i * W];
                for (int j = 0; j < n; j++) {
                    u64 acc = 0;
                    for (int k = 0; k < n; k++) acc = fadd(acc, fmul(pp.mds[i * W + 1 + k], Eprev[(size_t)k * n + j]));
                    eff[(size_t)i * W + 1 + j] = acc;
                }
            }
            for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) Eh[(size_t)i * n + j] = eff[(size_t)(1 + i) * W + 1 + j];
            if (!mat_inv(Eh.data(), Ei.data(), n)) return;   // (never for this table; permute() then keeps the plain form)
            e00[r] = to_mont(eff[0]);
            for (int j = 0; j < n; j++) row[r][j] = to_mont(eff[1 + j]);
            for (int i = 0; i < n; i++) {
                u64 acc = 0;
                for (int k = 0; k < n; k++) acc = fadd(acc, fmul(Ei[(size_t)i * n + k], eff[(size_t)(1 + k) * W]));
                col[r][i] = to_mont(acc);
            }
            Eprev = Eh;
            EprevInv = Ei;
        }
        for (int i = 0; i < n * n; i++) post[i] = to_mont(Eprev[i]);
        for (int i = 0; i < (RF + RP) * W; i++) ark[i] = to_mont(pp.ark[i]);
        for (int i = 0; i < W * W; i++) mds[i] = to_mont(pp.mds[i]);
        ok = true;
        // AVX-512 IFMA lanes (lfp_poseidon_simd.cc) when the CPU has them; LFPLUS_POSEIDON_SCALAR=1 keeps this scalar form
        if (lfp_psimd::supported() && !getenv("LFPLUS_POSEIDON_SCALAR")) {
            std::vector<u64> cstP((size_t)RP * W), e00P(RP), rowP((size_t)RP * n), colP((size_t)RP * n), postP((size_t)n * n);
            for (int r = 0; r < RP; r++) {
                e00P[r] = from_mont(e00[r]);
                for (int i = 0; i < W; i++) cstP[(size_t)r * W + i] = from_mont(cst[r][i]);
                for (int i = 0; i < n; i++) { rowP[(size_t)r * n + i] = from_mont(row[r][i]); colP[(size_t)r * n + i] = from_mont(col[r][i]); }
            }
            for (int i = 0; i < n * n; i++) postP[i] = from_mont(post[i]);
            lfp_psimd::build(P, pp.ark, pp.mds, cstP.data(), e00P.data(), rowP.data(), colP.data(), postP.data());
            simd = true;
        }
    }
    bool simd = false;
    inline u64 sbox(u64 x) const { const u64 x2 = mm(x, x), x4 = mm(x2, x2); return mm(mm(x4, x2), x); }
    inline void full_round(u64 *st, const u64 *a) const {
        u64 nw[W];
        for (int i = 0; i < W; i++) st[i] = sbox(fadd(st[i], a[i]));
        for (int i = 0; i < W; i++) nw[i] = dot(st, mds + i * W, W);
        memcpy(st, nw, sizeof(nw));
    }
    void run(u64 *st) const {
        for (int i = 0; i < W; i++) st[i] = mm(st[i] % P, r2);
        for (int r = 0; r < RF / 2; r++) full_round(st, ark + r * W);
        for (int r = 0; r < RP; r++) {
            for (int i = 0; i < W; i++) st[i] = fadd(st[i], cst[r][i]);
            const u64 x0 = sbox(st[0]);
            // y0 = e00 x0 + row . x[1..];  y_i = col_i x0 + x_i
            u128 lo = (u128)e00[r] * x0, hi = (u64)(lo >> 64);
            lo = (u64)lo;
            for (int j = 0; j < W - 1; j++) { const u128 pr = (u128)row[r][j] * st[1 + j]; lo += (u64)pr; hi += (u64)(pr >> 64); }
            for (int i = 0; i < W - 1; i++) st[1 + i] = fadd(st[1 + i], mm(col[r][i], x0));
            st[0] = fadd(redc(lo), small(hi));
        }
        {
            u64 nw[W - 1];
            for (int i = 0; i < W - 1; i++) nw[i] = dot(st + 1, post + i * (W - 1), W - 1);
            memcpy(st + 1, nw, sizeof(nw));
        }
        for (int r = RF / 2 + RP; r < RF + RP; r++) full_round(st, ark + r * W);
        for (int i = 0; i < W; i++) st[i] = mm(st[i], 1);
    }
};
const FastPerm &fastperm() { static const FastPerm f; return f; }
void permute(u64 *st) {
    const FastPerm &f = fastperm();
    if (f.simd) lfp_psimd::permute(st);
    else if (f.ok) f.run(st);
    else permute_plain(st);
}
}  // namespace

// ark-crypto-primitives 0.4.0 PoseidonSponge (duplex): state[0..4) capacity, [4..24) rate
struct lfplus_transcript {
    u64 st[W] = {0};
    bool squeezing = false;
    int idx = 0;
    void absorb_fq(const u64 *x, size_t n) {
        if (!n) return;
        int i0;
        if (!squeezing) { i0 = idx; if (i0 == RATE) { permute(st); i0 = 0; } }
        else { permute(st); i0 = 0; }
        for (;;) {
            if ((size_t)i0 + n <= RATE) {
                for (size_t i = 0; i < n; i++) st[CAP + i0 + i] = fadd(st[CAP + i0 + i], x[i] % P);
                squeezing = false;
                idx = i0 + (int)n;
                return;
            }
            const size_t take = RATE - i0;
            for (size_t i = 0; i < take; i++) st[CAP + i0 + i] = fadd(st[CAP + i0 + i], x[i] % P);
            permute(st);
            x += take; n -= take; i0 = 0;
        }
    }
    void squeeze_fq(u64 *out, size_t n) {
        int i0;
        if (!squeezing) { permute(st); i0 = 0; }
        else { i0 = idx; if (i0 == RATE) { permute(st); i0 = 0; } }
        for (;;) {
            if ((size_t)i0 + n <= RATE) {
                memcpy(out, st + CAP + i0, n * sizeof(u64));
                squeezing = true;
                idx = i0 + (int)n;
                return;
            }
            const size_t take = RATE - i0;
            memcpy(out, st + CAP + i0, take * sizeof(u64));
            if (n != RATE) permute(st);
            out += take; n -= take; i0 = 0;
        }
    }
    void absorb_ring(const u64 *e, size_t count) { for (size_t i = 0; i < count; i++) absorb_fq(e + i * D, D); }   // Transcript::absorb: the 16 coefficients
    void absorb_const(u64 c) { u64 e[D] = {0}; e[0] = c % P; absorb_ring(e, 1); }                                   // absorb(&R::from(c))
    u64 challenge() { u64 c; squeeze_fq(&c, 1); absorb_fq(&c, 1); return c; }                                       // transcript.rs:44-53
    void squeeze_bytes(size_t n, uint8_t *out) {                                                                    // 7 low LE bytes per element
        std::vector<u64> e((n + 6) / 7);
        squeeze_fq(e.data(), e.size());
        for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(e[i / 7] >> (8 * (i % 7)));
    }
};

extern "C" {
lfplus_transcript *lfplus_transcript_new(void) { return new lfplus_transcript; }
lfplus_transcript *lfplus_transcript_clone(const lfplus_transcript *t) { return t ? new lfplus_transcript(*t) : nullptr; }
void lfplus_transcript_free(lfplus_transcript *t) { delete t; }
int lfplus_transcript_absorb(lfplus_transcript *t, const uint64_t *ring, size_t count) {
    if (!t || (!ring && count)) return LFPLUS_E_ARG;
    t->absorb_ring(ring, count);
    return LFPLUS_OK;
}
int lfplus_transcript_challenge(lfplus_transcript *t, uint64_t *out) {
    if (!t || !out) return LFPLUS_E_ARG;
    *out = t->challenge();
    return LFPLUS_OK;
}
int lfplus_transcript_squeeze_bytes(lfplus_transcript *t, size_t n, uint8_t *out) {
    if (!t || (!out && n)) return LFPLUS_E_ARG;
    t->squeeze_bytes(n, out);
    return LFPLUS_OK;
}
// utils::short_challenge(128, ..) (utils.rs:87-101): 16 bytes, coefficient = byte % 256 - 128
int lfplus_short_challenge(lfplus_transcript *t, uint64_t *out16) {
    if (!t || !out16) return LFPLUS_E_ARG;
    uint8_t bs[D];
    t->squeeze_bytes(D, bs);
    for (int i = 0; i < D; i++) { const int v = (int)bs[i] - 128; out16[i] = v >= 0 ? (u64)v : P - (u64)(-v); }
    return LFPLUS_OK;
}
// one permutation of a 24-word state (canonical words): plain = 1 runs the textbook definition, 2 the scalar sparse form (FastPerm), 0 the form the transcript uses
// (the AVX-512 IFMA lanes of lfp_poseidon_simd.cc when the CPU has them, else FastPerm)
int lfplus_poseidon_permute(uint64_t *state24, int plain) {
    if (!state24) return LFPLUS_E_ARG;
    for (int i = 0; i < W; i++) state24[i] %= P;
    if (plain == 1) permute_plain(state24);
    else if (plain == 2 && fastperm().ok) fastperm().run(state24);
    else permute(state24);
    return LFPLUS_OK;
}
// 1 when the transcript's permutation runs on the AVX-512 IFMA lanes
int lfplus_poseidon_simd(void) { return fastperm().simd ? 1 : 0; }
int lfplus_poseidon_params(uint64_t *ark720, uint64_t *mds576) {
    if (!ark720 || !mds576) return LFPLUS_E_ARG;
    memcpy(ark720, params().ark, sizeof(params().ark));
    memcpy(mds576, params().mds, sizeof(params().mds));
    return LFPLUS_OK;
}
}

// ---- device side of the set check -----------------------------------------------------------------------------------------------------
namespace {
// scratch of one call: taken from the scratch pool of the context the entry point named (PoolScope), returned to it when the call ends
thread_local lfplus_ctx *t_pool_ctx = nullptr;
struct PoolScope {
    lfplus_ctx *prev;
    explicit PoolScope(lfplus_ctx *c) : prev(t_pool_ctx) { t_pool_ctx = c; }
    ~PoolScope() { t_pool_ctx = prev; }
};
struct DevBuf {
    void *p = nullptr;
    lfplus_ctx *owner = nullptr;
    ~DevBuf() {
        if (!p) return;
        if (owner) owner->pool.put(p); else (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        owner = t_pool_ctx;
        if (owner) { p = owner->pool.get(bytes); return p ? 0 : -1; }
        return lfp_dev_malloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : -1;
    }
    template <class T> T *as() const { return (T *)p; }
};
struct SetRef { const int8_t *dig; u32 ncols; };   // device pointer to the exponent digits [n][ncols]
// n x n CSR matrices (ring coefficients) -> device, both orientations; the transposition runs on the host, the Montgomery conversion on the device
int upload_matrix(lfplus_ctx *c, size_t n, const u32 *rowptr, const u32 *col, const u64 *val, LfpMatrix &m) {
    if (!rowptr || !col || !val || rowptr[0] != 0) return fail(c, LFPLUS_E_ARG, "matrix: null / malformed CSR");
    for (size_t r = 0; r < n; r++) if (rowptr[r + 1] < rowptr[r]) return fail(c, LFPLUS_E_ARG, "matrix: rowptr not monotone");
    const size_t nnz = rowptr[n];
    std::vector<u32> cp(n + 1, 0), ri(nnz);
    for (size_t k = 0; k < nnz; k++) {
        if (col[k] >= n) return fail(c, LFPLUS_E_ARG, "matrix: column index out of range");
        cp[col[k] + 1]++;
    }
    if (!canonical(val, nnz * D)) return fail(c, LFPLUS_E_ARG, "matrix: non-canonical word");
    for (size_t i = 0; i < n; i++) cp[i + 1] += cp[i];
    std::vector<u32> fill(cp.begin(), cp.end() - 1);
    std::vector<u64> vv(nnz * D);
    for (size_t r = 0; r < n; r++)
        for (u32 k = rowptr[r]; k < rowptr[r + 1]; k++) {
            const u32 dst = fill[col[k]]++;
            ri[dst] = (u32)r;
            memcpy(&vv[(size_t)dst * D], val + (size_t)k * D, D * 8);
        }
    m.nnz = nnz;
    m.const_coef = true;
    for (size_t k = 0; k < nnz && m.const_coef; k++)
        for (int w = 1; w < D; w++) if (val[k * D + w]) { m.const_coef = false; break; }
    const size_t vb = (nnz ? nnz : 1) * D * 8, ib = (nnz ? nnz : 1) * 4;
    if (lfp_dev_malloc(&m.rowptr, (n + 1) * 4) != hipSuccess || lfp_dev_malloc(&m.col, ib) != hipSuccess || lfp_dev_malloc(&m.valM, vb) != hipSuccess ||
        lfp_dev_malloc(&m.colptr, (n + 1) * 4) != hipSuccess || lfp_dev_malloc(&m.rowidx, ib) != hipSuccess || lfp_dev_malloc(&m.valT, vb) != hipSuccess)
        return fail(c, LFPLUS_E_HIP, "hipMalloc (matrix)");
    HIPCHK(c, hipMemcpyAsync(m.rowptr, rowptr, (n + 1) * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(m.col, col, nnz * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(m.valT, val, nnz * D * 8, hipMemcpyHostToDevice, c->st));     // staged through valT, converted below
    lfp::launch_to_mont(m.valT, nnz * D, m.valM, c->st);
    HIPCHK(c, hipMemcpyAsync(m.colptr, cp.data(), (n + 1) * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(m.rowidx, ri.data(), nnz * 4, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipMemcpyAsync(m.valT, vv.data(), nnz * D * 8, hipMemcpyHostToDevice, c->st));
    if (m.const_coef) {
        if (lfp_dev_malloc(&m.valMc, (nnz ? nnz : 1) * 8) != hipSuccess || lfp_dev_malloc(&m.valTc, (nnz ? nnz : 1) * 8) != hipSuccess) return fail(c, LFPLUS_E_HIP, "hipMalloc (matrix)");
        if (nnz) {
            HIPCHK(c, hipMemcpy2DAsync(m.valMc, 8, m.valM, D * 8, 8, nnz, hipMemcpyDeviceToDevice, c->st));
            HIPCHK(c, hipMemcpy2DAsync(m.valTc, 8, m.valT, D * 8, 8, nnz, hipMemcpyDeviceToDevice, c->st));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->st));   // the host staging vectors die here
    return LFPLUS_OK;
}
// the matrices of one call: the caller's CSR arrays uploaded for the duration of the call, or (rowptr == NULL) the set lfplus_set_matrices left in the context
struct MatHold {
    std::vector<LfpMatrix> own;
    const LfpMatrix *m = nullptr;
    u32 count = 0;
    ~MatHold() { for (LfpMatrix &x : own) x.release(); }
    int get(lfplus_ctx *c, size_t n, u32 nM, const u32 *const *rowptr, const u32 *const *col, const u64 *const *val) {
        count = nM;
        if (!nM) return LFPLUS_OK;
        if (!rowptr) {
            if (c->mats.size() != nM || c->mats_n != n) return fail(c, LFPLUS_E_ARG, "no resident matrices of this shape (lfplus_set_matrices)");
            m = c->mats.data();
            return LFPLUS_OK;
        }
        if (!col || !val) return fail(c, LFPLUS_E_ARG, "matrix: null argument");
        own.resize(nM);
        for (u32 q = 0; q < nM; q++) {
            int rc = upload_matrix(c, n, rowptr[q], col[q], val[q], own[q]);
            if (rc) return rc;
        }
        m = own.data();
        return LFPLUS_OK;
    }
    const LfpMatrix &operator[](u32 q) const { return m[q]; }
    u32 size() const { return count; }
};

struct ScOut {          // device-side leftovers the range check reuses
    DevBuf eqr;         // eq(r, .) Montgomery, n words (whole on every rank: M^T eq reads arbitrary rows)
    std::vector<std::unique_ptr<DevBuf>> w;   // w_q = M_q^T eq(r): the rank's nloc ring elements each
    std::function<void(lfplus_transcript *)> absorb;      // the absorb of the evaluations, for a caller that deferred it (set_check_dev's defer_absorb)
    bool wscalar = false;                     // every M_q has constant coefficients: w_q holds nloc SCALARS (Montgomery) -- eq is scalar, so w_q is constant too, and
                                              // every sum weighted by it takes the scalar-weight form (no negacyclic rotations, an eighth of the weight traffic)
    DevBuf part, small;
};
// eq(c, .) over the rank's rows [row0, row0 + nloc): the local index carries the low nv_loc variables, the rank the high ones, so the slice is the
// nv_loc-variable table scaled by eq(c_hi, rank) (pt.one)
lfp::EqPt eq_point(const u64 *cch, u32 nvars) {
    lfp::EqPt pt;
    for (u32 j = 0; j < nvars; j++) { pt.c[j] = to_mont(cch[j]); pt.nc[j] = to_mont(fsub(1, cch[j])); }
    pt.one = to_mont(1);
    return pt;
}
void eq_build_local(lfplus_ctx *c, const u64 *cch, u32 nvars, u64 *eq_loc) {
    lfp::EqPt pt = eq_point(cch, nvars);
    u32 nv_loc = nvars;
    if (c->sharded()) {
        nv_loc = 0;
        while (((u64)1 << nv_loc) < c->nloc) nv_loc++;
        u64 sc = 1;
        for (u32 j = nv_loc; j < nvars; j++) sc = fmul(sc, ((u64)c->rank >> (j - nv_loc)) & 1 ? cch[j] : fsub(1, cch[j]));
        pt.one = to_mont(sc);
    }
    lfp::launch_eq_build(pt, nv_loc, eq_loc, c->st);
}
// The last log2(world) rounds of a sharded sumcheck: every rank holds ONE entry of each of `ntab` tables of `w` words (device, row stride ld words); gather them and
// lay them out as whole tables of `world` entries (entry index = rank: the high index bits) in `out` (device, ntab x world x w words, row stride world)
int gather_tables(lfplus_ctx *c, const u64 *tab, size_t ld, u32 ntab, u32 w, u64 *out) {
    std::vector<u64> mine((size_t)ntab * w), all, re((size_t)ntab * c->world * w);
    HIPCHK(c, hipMemcpy2DAsync(mine.data(), (size_t)w * 8, tab, ld * w * 8, (size_t)w * 8, ntab, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    int rc = lfp_allgather(c, mine.data(), mine.size(), all);
    if (rc) return rc;
    for (u32 t = 0; t < ntab; t++)
        for (int g = 0; g < c->world; g++) memcpy(&re[((size_t)t * c->world + g) * w], &all[((size_t)g * ntab + t) * w], (size_t)w * 8);
    HIPCHK(c, hipMemcpyAsync(out, re.data(), re.size() * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));   // re is a local buffer
    return LFPLUS_OK;
}

// In::set_check on device-resident monomial sets (matrix sets first, then vector sets, as setchk.rs:66-82 orders them)
int set_check_dev(lfplus_ctx *c, lfplus_transcript *tr, u32 nvars, const std::vector<SetRef> &mats, const std::vector<SetRef> &vecs,
                  const MatHold &M, u64 *r_out, u64 *msgs, u64 *e_out, u64 *b_out, ScOut &so, bool defer_absorb = false) {
    // Sharded (c->world > 1): the SetRefs point at the rank's rows; tables, rounds and evaluations run over those nl rows, the partial round messages and
    // partial evaluations are summed over the ranks, and the last log2(world) rounds run replicated on gathered tables.
    const size_t n = (size_t)1 << nvars, nl = c->sharded() ? (size_t)c->nloc : n;
    if (c->sharded() && c->n != n) return fail(c, LFPLUS_E_ARG, "set_check: a sharded context checks sets of its own width n");
    const u32 nmat = (u32)mats.size(), nvec = (u32)vecs.size(), nM = (u32)M.size();
    if (nmat < 1) return fail(c, LFPLUS_E_ARG, "set_check: at least one matrix set (setchk.rs:63)");
    const u32 ncols = mats[0].ncols;
    for (auto &m : mats) if (m.ncols != ncols) return fail(c, LFPLUS_E_ARG, "set_check: matrix sets of different widths");
    const u32 ntab = nmat * (2 * ncols + 1) + 3 * nvec;
    DevBuf tabs[2], coefd, tgath;
    if (tabs[0].alloc((size_t)ntab * nl * 8) || tabs[1].alloc((size_t)ntab * nl * 8) || coefd.alloc((size_t)(nmat + nvec) * ncols * 8) ||
        so.eqr.alloc(n * 8) || so.part.alloc((size_t)lfp::eval_chunks(n) * ncols * 16 * 8 * 4) ||      // (x 4: launch_whist16 takes four weight tables per pass)
        so.small.alloc((size_t)((1 + nM) * nmat * ncols + nvec + 8) * D * 8) || (c->sharded() && tgath.alloc((size_t)ntab * c->world * 8)))
        return fail(c, LFPLUS_E_HIP, "hipMalloc (set check tables)");
    std::vector<u64> alpha(nmat + nvec), cch(nvars);
    // Round 0 ahead of time.  The round polynomial is sum_i rc^i S_i(X), S_i = the sum over the pairs of eq_i (sum_j alpha_i^j (m_ij^2 - m'_ij)): S_i needs set i's tables
    // and alpha_i only, not the batching challenge rc drawn after the last set -- so set i's share of round 0 is enqueued right behind its tables, under the ~0.2 ms
    // of Poseidon the host spends on the next set's nvars + 2 challenges, and the message is combined on the host once rc is known (the same field elements).
    // (Without rc -- a single matrix set -- the reference's closure has its own shape, see coef below: the round runs in the loop as before.)
    const bool early = nmat > 1 && nl >= 2 && !(c->sharded() && nl < 2) && !getenv("LFPLUS_SC_NO_EARLY");
    DevBuf coefA;
    std::vector<u64> coefAh((size_t)(nmat + nvec) * ncols, 0);
    u64 *hpart = c->pin(std::max((size_t)lfp::sc_round_max_blocks() * 4, (size_t)(nmat + nvec) * 4));   // the kernels write their block partials into mapped host memory
    if (!hpart) return fail(c, LFPLUS_E_HIP, "hipHostMalloc (round partials)");
    if (early && coefA.alloc(coefAh.size() * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (set check coefficients)");
    // Rounds 0 and 1 straight from the exponent digits (lfp_rgchk.hip: k_sc_round0_dig, k_sc_fix_round_dig): the beta^e / beta^2e tables of n entries are
    // never written -- the first tables that exist are the n / 2-entry ones of the fused "fix at r_0 + round 1" pass.  LFPLUS_SC_TABLES=1: the table form.
    const bool from_dig = early && !c->sharded() && nl >= 8 && (ncols == 16 || ncols == 1) && !getenv("LFPLUS_SC_TABLES");
    std::vector<lfp::PwTab> pws(from_dig ? nmat + nvec : 0);
    for (u32 i = 0; i < nmat + nvec; i++) {
        const SetRef &sr = i < nmat ? mats[i] : vecs[i - nmat];
        const u32 cols = i < nmat ? ncols : 1, t0 = i < nmat ? i * (2 * ncols + 1) : nmat * (2 * ncols + 1) + 3 * (i - nmat);
        for (u32 j = 0; j < nvars; j++) cch[j] = tr->challenge();
        const u64 beta = tr->challenge();
        lfp::PwTab pw;
        u64 bp = 1;
        for (int t = 0; t < 16; t++) { pw.p[t] = to_mont(bp); pw.q[t] = to_mont(fmul(bp, bp)); bp = fmul(bp, beta); }
        if (from_dig) pws[i] = pw;
        else lfp::launch_sc_tables(sr.dig, nl, cols, pw, tabs[0].as<u64>() + (size_t)t0 * nl, nl, c->st);
        eq_build_local(c, cch.data(), nvars, tabs[0].as<u64>() + (size_t)(t0 + 2 * cols) * nl);
        alpha[i] = tr->challenge();
        if (early) {
            u64 ap = i < nmat ? 1 : alpha[i];
            for (u32 j = 0; j < cols; j++) { coefAh[(size_t)i * ncols + j] = to_mont(ap); ap = fmul(ap, alpha[i]); }
            HIPCHK(c, hipMemcpyAsync(coefA.as<u64>() + (size_t)i * ncols, &coefAh[(size_t)i * ncols], cols * 8, hipMemcpyHostToDevice, c->st));
            const lfp::ScDesc di = i < nmat ? lfp::ScDesc{1, ncols, 0, 1} : lfp::ScDesc{0, ncols, 1, 1};      // the one set, its tables at the origin
            const u32 nbi = from_dig ? lfp::launch_sc_round0_dig(sr.dig, nl, cols, pw, tabs[0].as<u64>() + (size_t)(t0 + 2 * cols) * nl, coefA.as<u64>() + (size_t)i * ncols,
                                                                  so.part.as<u64>(), c->st)
                                     : lfp::launch_sc_round(tabs[0].as<u64>() + (size_t)t0 * nl, nl, nl / 2, di, coefA.as<u64>() + (size_t)i * ncols, so.part.as<u64>(), c->st);
            if (!nbi) return fail(c, LFPLUS_E_ARG, "set_check: set width not handled by the digit rounds");
            lfp::launch_sum_parts(so.part.as<u64>(), nbi, 4, 4, c->hpin_dev + (size_t)i * 4, c->st);
        }
    }
    const bool have_rc = nmat > 1;
    const u64 rc = have_rc ? tr->challenge() : 1;
    // coef[i][j] = rc^i alpha_i^j (vector set: rc^i alpha_i); without rc the closure returns after the first matrix set (setchk.rs:172-176)
    std::vector<u64> coef((size_t)(nmat + nvec) * ncols, 0);
    u64 rcp = 1;
    for (u32 i = 0; i < nmat + nvec; i++) {
        u64 ap = i < nmat ? 1 : alpha[i];
        for (u32 j = 0; j < (i < nmat ? ncols : 1); j++) { coef[(size_t)i * ncols + j] = to_mont(fmul(rcp, ap)); ap = fmul(ap, alpha[i]); }
        rcp = fmul(rcp, rc);
    }
    HIPCHK(c, hipMemcpyAsync(coefd.p, coef.data(), coef.size() * 8, hipMemcpyHostToDevice, c->st));
    lfp::ScDesc d = {nmat, ncols, nvec, have_rc ? nmat + nvec : 1};
    LFP_MARK(c, "set check: tables");
    // MLSumcheck::prove_as_subprotocol (latticefold utils/sumcheck.rs:53-80), degree 3
    tr->absorb_const(nvars);
    tr->absorb_const(3);
    int cur = 0;
    size_t len = nl, ld = nl;
    bool dist = c->sharded();
    const u64 *tcur = tabs[0].as<u64>();
    for (u32 rnd = 0; rnd < nvars; rnd++) {
        if (dist && len == 1) {      // one entry per table and rank left: gather, finish replicated
            int rcg = gather_tables(c, tcur, ld, ntab, 1, tgath.as<u64>());
            if (rcg) return rcg;
            tcur = tgath.as<u64>();
            ld = len = (size_t)c->world;
            dist = false;
        }
        const size_t half = len / 2;
        auto tA = std::chrono::steady_clock::now();
        const bool pre = early && rnd == 0;           // the sets' shares S_i are in hpart[i][4] already (Montgomery): the message is sum_i rc^i S_i
        const bool pre1 = from_dig && rnd == 1;       // ... and round 1's shares (rc^i inside their coefficients) came with the fused fix: the message is their sum
        const u32 nb = pre || pre1 ? 0 : lfp::launch_sc_round(tcur, ld, half, d, coefd.as<u64>(), c->hpin_dev, c->st);
        HIPCHK(c, hipStreamSynchronize(c->st));
        auto tB = std::chrono::steady_clock::now();
        u64 *m = msgs + (size_t)rnd * 4 * D;
        memset(m, 0, 4 * D * 8);
        u64 sums[4];
        for (int x = 0; x < 4; x++) {
            u64 s = 0;
            if (pre) {
                u64 rp = 1;
                for (u32 i = 0; i < nmat + nvec; i++) { s = fadd(s, fmul(rp, hpart[(size_t)i * 4 + x])); rp = fmul(rp, rc); }
            } else if (pre1) {
                for (u32 i = 0; i < nmat + nvec; i++) s = fadd(s, hpart[(size_t)i * 4 + x]);
            } else
                for (u32 b = 0; b < nb; b++) s = fadd(s, hpart[(size_t)b * 4 + x]);
            sums[x] = s;
        }
        if (dist) { int rcx = lfp_xsum(c, sums, 4); if (rcx) return rcx; }
        for (int x = 0; x < 4; x++) m[x * D] = from_mont(sums[x]);
        tr->absorb_ring(m, 4);
        const u64 r = tr->challenge();
        tr->absorb_const(r);
        r_out[rnd] = r;
        auto tC = std::chrono::steady_clock::now();
        if (from_dig && rnd == 0) {   // fix at r_0 from the digits, fused with round 1 (set by set: each has its own digits and power table)
            for (u32 i = 0; i < nmat + nvec; i++) {
                const SetRef &sr = i < nmat ? mats[i] : vecs[i - nmat];
                const u32 cols = i < nmat ? ncols : 1, t0 = i < nmat ? i * (2 * ncols + 1) : nmat * (2 * ncols + 1) + 3 * (i - nmat);
                const u32 nbi = lfp::launch_sc_fix_round_dig(sr.dig, nl, cols, pws[i], tabs[0].as<u64>() + (size_t)(t0 + 2 * cols) * nl, to_mont(r),
                                                             tabs[1].as<u64>() + (size_t)t0 * nl, nl, coefd.as<u64>() + (size_t)i * ncols, so.part.as<u64>(), c->st);
                lfp::launch_sum_parts(so.part.as<u64>(), nbi, 4, 4, c->hpin_dev + (size_t)i * 4, c->st);
            }
            cur = 1;
            tcur = tabs[1].as<u64>();
        } else
        if (rnd + 1 < nvars) {   // fix_variables of every table into the other buffer (rows keep their stride)
            if (tcur == tgath.as<u64>() && c->sharded()) {      // the gathered tables have stride world: fix them into tabs[0] with the same stride
                lfp::launch_sc_fix(tcur, tabs[0].as<u64>(), ld, ntab, half, to_mont(r), c->st);
                cur = 0;
            } else {
                lfp::launch_sc_fix(tcur, tabs[cur ^ 1].as<u64>(), ld, ntab, half, to_mont(r), c->st);
                cur ^= 1;
            }
            tcur = tabs[cur].as<u64>();
        }
        if (g_tl.on) fprintf(stderr, "[lfplus]   sc round %2u: gpu+sync %6.1f us, host %6.1f us, fix launch %5.1f us (nb %u)\n", rnd, std::chrono::duration<double, std::micro>(tB - tA).count(),
                             std::chrono::duration<double, std::micro>(tC - tB).count(), std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tC).count(), nb);
        len = half;
    }
    LFP_MARK(c, "set check: sumcheck rounds");
    // Step 3 (setchk.rs:206-249): the sets at r, M_q * sets at r, the vector sets at r
    const size_t row0 = c->sharded() ? (size_t)c->row0 : 0;
    lfp::launch_eq_build(eq_point(r_out, nvars), nvars, so.eqr.as<u64>(), c->st);
    const u64 *eql = so.eqr.as<u64>() + row0;      // eq(r, .) over the rank's rows
    so.wscalar = nM > 0 && !getenv("LFPLUS_RING_WEIGHTS");
    for (u32 q = 0; q < nM; q++) so.wscalar = so.wscalar && M[q].const_coef;
    const u32 wst = so.wscalar ? 1 : 16;
    for (u32 q = 0; q < nM; q++) {                 // w_q = M_q^T eq(r) over the rank's COLUMNS (its rows of the vectors M_q multiplies); eq whole
        std::unique_ptr<DevBuf> w(new DevBuf);
        if (w->alloc(nl * wst * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (M^T eq)");
        if (so.wscalar) lfp::launch_spmvT_eq_const(M[q].colptr + row0, M[q].rowidx, M[q].valTc, so.eqr.as<u64>(), nl, w->as<u64>(), c->st);
        else lfp::launch_spmvT_eq(M[q].colptr + row0, M[q].rowidx, M[q].valT, so.eqr.as<u64>(), nl, w->as<u64>(), c->st);
        so.w.push_back(std::move(w));
    }
    LFP_MARK(c, "set check: eq(r), M^T eq(r)");
    u64 *ed = so.small.as<u64>(), *bd = ed + (size_t)(1 + nM) * nmat * ncols * D;
    // every weight table scalar (eq(r) always is; the w_q when the M_q have constant coefficients) and 16 columns: the exponent-histogram pass, four tables at a time
    const bool hist = ncols == 16 && (nM == 0 || so.wscalar);
    for (u32 i = 0; i < nmat; i++) {
        if (hist) {
            std::vector<const u64 *> wt(1 + nM);
            wt[0] = eql;
            for (u32 q = 0; q < nM; q++) wt[1 + q] = so.w[q]->as<u64>();
            for (u32 q0 = 0; q0 < 1 + nM; q0 += 4)
                lfp::launch_whist16(mats[i].dig, nl, wt.data() + q0, std::min(4u, 1 + nM - q0), so.part.as<u64>(), ed + ((size_t)q0 * nmat + i) * ncols * D, (size_t)nmat * ncols * D, c->st);
            continue;
        }
        lfp::launch_wmono(mats[i].dig, nl, ncols, eql, 1, so.part.as<u64>(), ed + (size_t)i * ncols * D, c->st);
        for (u32 q = 0; q < nM; q++)
            lfp::launch_wmono(mats[i].dig, nl, ncols, so.w[q]->as<u64>(), wst, so.part.as<u64>(), ed + ((size_t)(1 + q) * nmat + i) * ncols * D, c->st);
    }
    for (u32 i = 0; i < nvec; i++) lfp::launch_wmono(vecs[i].dig, nl, 1, eql, 1, so.part.as<u64>(), bd + (size_t)i * D, c->st);
    HIPCHK(c, hipMemcpyAsync(e_out, ed, (size_t)(1 + nM) * nmat * ncols * D * 8, hipMemcpyDeviceToHost, c->st));
    if (nvec) HIPCHK(c, hipMemcpyAsync(b_out, bd, (size_t)nvec * D * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    LFP_MARK(c, "set check: evaluation passes");
    {   // partial evaluations over the rank's rows -> sums over the ranks
        int rcx = lfp_xsum(c, e_out, (size_t)(1 + nM) * nmat * ncols * D);
        if (!rcx && nvec) rcx = lfp_xsum(c, b_out, (size_t)nvec * D);
        if (rcx) return rcx;
    }
    so.absorb = [=](lfplus_transcript *t) {                    // absorb_evaluations (setchk.rs:342-353): 3 ms of Poseidon at the range check's shape (768 ring elements)
        t->absorb_ring(e_out, (size_t)(1 + nM) * nmat * ncols);
        t->absorb_ring(b_out, nvec);
    };
    if (!defer_absorb) {       // (the range check enqueues its own evaluation passes first and absorbs while they run)
        so.absorb(tr);
        LFP_MARK(c, "set check: evaluations + absorb");
    }
    return LFPLUS_OK;
}
}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------------
// In::set_check (setchk.rs:65-262) on monomial sets handed over as exponent digits (host arrays; int8 in (-8, 8), LFPLUS_ABSENT = a zero
// entry): nmat matrices of n x ncols, nvec vectors of n entries, n = 2^nvars.
extern "C" int lfplus_set_check(lfplus_ctx *c, lfplus_transcript *tr, uint32_t nvars, const int8_t *mat_digits, uint32_t nmat, uint32_t ncols,
                                const int8_t *vec_digits, uint32_t nvec, uint32_t nM, const uint32_t *const *rowptr, const uint32_t *const *col,
                                const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out) {
    if (!c || !tr || !mat_digits || !nmat || !ncols || ncols > 64 || nvars < 1 || nvars > 28 || (nvec && !vec_digits) || !r_out || !msgs || !e_out || (nvec && !b_out))
        return fail(c, LFPLUS_E_ARG, "lfplus_set_check: bad arguments");
    const size_t n = (size_t)1 << nvars;
    for (size_t i = 0; i < (size_t)nmat * n * ncols + (size_t)nvec * n; i++) {
        const int8_t d = i < (size_t)nmat * n * ncols ? mat_digits[i] : vec_digits[i - (size_t)nmat * n * ncols];
        if (d != lfp::LFP_ABSENT && (d <= -8 || d >= 8)) return fail(c, LFPLUS_E_EXP_DOMAIN, "lfplus_set_check: digit outside (-8, 8)");
    }
    HIPCHK(c, hipSetDevice(c->device));
    PoolScope pool_scope(c);
    DevBuf dm, dv;
    if (dm.alloc((size_t)nmat * n * ncols) || dv.alloc((size_t)nvec * n)) return fail(c, LFPLUS_E_HIP, "hipMalloc (sets)");
    HIPCHK(c, hipMemcpyAsync(dm.p, mat_digits, (size_t)nmat * n * ncols, hipMemcpyHostToDevice, c->st));
    if (nvec) HIPCHK(c, hipMemcpyAsync(dv.p, vec_digits, (size_t)nvec * n, hipMemcpyHostToDevice, c->st));
    std::vector<SetRef> mats, vecs;
    const size_t r0 = c->sharded() ? (size_t)c->row0 : 0;     // a sharded context works on its rows of the sets (uploaded whole: they are int8)
    for (u32 i = 0; i < nmat; i++) mats.push_back({dm.as<int8_t>() + ((size_t)i * n + r0) * ncols, ncols});
    for (u32 i = 0; i < nvec; i++) vecs.push_back({dv.as<int8_t>() + (size_t)i * n + r0, 1});
    MatHold M;
    int rc = M.get(c, n, nM, rowptr, col, val);
    if (rc) return rc;
    ScOut so;
    return set_check_dev(c, tr, nvars, mats, vecs, M, r_out, msgs, e_out, b_out, so);
}

// Rg::range_check (rgchk.rs:81-186) on L resident instances: ctxs[l] holds the witness f_l and the results of lfplus_rg_from_f (D_f, tau,
// m_tau) -- all on the same device, same n = 2^nvars and k.  Runs on ctxs[0]'s stream.
namespace {
int range_check_core(lfplus_ctx *const *ctxs, uint32_t L, lfplus_transcript *tr, const MatHold &M, uint64_t *r_out, uint64_t *msgs,
                     uint64_t *e_out, uint64_t *b_out, uint64_t *v_out, uint64_t *a_out, uint64_t *bb_out, uint64_t *c_out, ScOut &so,
                     const std::function<int()> &before_absorb = nullptr) {      // before_absorb: device work of the caller that needs no challenge -- enqueued ahead of the 3 ms of
                                                                                  // host Poseidon that absorb the set check's evaluations
    lfplus_ctx *c = ctxs[0];
    const u32 nM = (u32)M.size();
    const u64 n = c->n;
    const u32 k = c->k;
    if (!n || (n & (n - 1))) return fail(c, LFPLUS_E_ARG, "lfplus_range_check: n must be a power of two");
    u32 nvars = 0;
    while (((u64)1 << nvars) < n) nvars++;
    for (u32 l = 0; l < L; l++)
        if (!ctxs[l] || !ctxs[l]->have || ctxs[l]->n != n || ctxs[l]->nf != n || ctxs[l]->k != k || ctxs[l]->device != c->device)
            return fail(c, LFPLUS_E_ARG, "lfplus_range_check: every instance needs lfplus_rg_from_f results of the same shape on the same device");
    HIPCHK(c, hipSetDevice(c->device));
    for (u32 l = 1; l < L; l++) HIPCHK(c, hipStreamSynchronize(ctxs[l]->st));   // their from_f results are read from ctxs[0]'s stream
    std::vector<SetRef> mats, vecs;   // rgchk.rs:87-96: all instances' M_f matrices, then all instances' m_tau
    const size_t nl = (size_t)c->nloc, row0 = (size_t)c->row0;      // the rank's rows (all of them unsharded); D_f is stored [k][nl][16]
    for (u32 l = 0; l < L; l++)
        if (ctxs[l]->sh != c->sh || ctxs[l]->nloc != c->nloc) return fail(c, LFPLUS_E_ARG, "lfplus_range_check: instances of different shardings");
    for (u32 l = 0; l < L; l++)
        for (u32 ki = 0; ki < k; ki++) mats.push_back({ctxs[l]->Df + (size_t)ki * nl * 16, 16});
    for (u32 l = 0; l < L; l++) vecs.push_back({ctxs[l]->mtau + row0, 1});
    int rc = set_check_dev(c, tr, nvars, mats, vecs, M, r_out, msgs, e_out, b_out, so, true);
    if (rc) return rc;
    // evaluations at r (rgchk.rs:107-170): v / c[0] = f at r, a[0] = tau at r, b[0] = the set check's b; per matrix M_q: ct(M_q tau), M_q m_tau, M_q f
    DevBuf ev;
    const size_t per = (size_t)(1 + nM) * (1 + 2 * D) + D;   // words per instance: a | bb | c (v = c[0])
    if (ev.alloc(L * per * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (evaluations)");
    const u64 *eql = so.eqr.as<u64>() + row0;
    HIPCHK(c, hipMemsetAsync(ev.p, 0, L * per * 8, c->st));     // (bb[0] is not computed here: keep the summed words defined)
    for (u32 l = 0; l < L; l++) {
        u64 *ad = ev.as<u64>() + l * per, *bd = ad + (1 + nM), *cd = bd + (size_t)(1 + nM) * D;
        const u64 *fl = ctxs[l]->f + row0 * D, *taul = ctxs[l]->tau + row0;
        lfp::launch_wring(fl, nl, eql, 1, so.part.as<u64>(), cd, c->st);
        lfp::launch_wdot(eql, 1, 1, taul, nl, so.part.as<u64>(), ad, c->st);
        for (u32 q = 0; q < nM; q++) {
            const u32 wst = so.wscalar ? 1 : 16;       // scalar weights are in Montgomery form, ring weights canonical
            lfp::launch_wdot(so.w[q]->as<u64>(), wst, so.wscalar ? 1 : 0, taul, nl, so.part.as<u64>(), ad + 1 + q, c->st);
            lfp::launch_wmono(ctxs[l]->mtau + row0, nl, 1, so.w[q]->as<u64>(), wst, so.part.as<u64>(), bd + (size_t)(1 + q) * D, c->st);
            lfp::launch_wring(fl, nl, so.w[q]->as<u64>(), wst, so.part.as<u64>(), cd + (size_t)(1 + q) * D, c->st);
        }
    }
    std::vector<u64> h(L * per);
    HIPCHK(c, hipMemcpyAsync(h.data(), ev.p, h.size() * 8, hipMemcpyDeviceToHost, c->st));
    if (before_absorb) { const int rcb = before_absorb(); if (rcb) return rcb; }
    so.absorb(tr);             // the set check's evaluations: host Poseidon while the passes above run
    LFP_MARK(c, "set check: evaluations absorbed (range check passes enqueued)");
    HIPCHK(c, hipStreamSynchronize(c->st));
    if ((rc = lfp_xsum(c, h.data(), h.size()))) return rc;
    for (u32 l = 0; l < L; l++) {
        const u64 *ah = h.data() + l * per, *bh = ah + (1 + nM), *ch = bh + (size_t)(1 + nM) * D;
        memcpy(a_out + (size_t)l * (1 + nM), ah, (1 + nM) * 8);
        memcpy(bb_out + (size_t)l * (1 + nM) * D, bh, (size_t)(1 + nM) * D * 8);
        memcpy(bb_out + (size_t)l * (1 + nM) * D, b_out + (size_t)l * D, D * 8);   // b[0] = out_rel.b[l]
        memcpy(c_out + (size_t)l * (1 + nM) * D, ch, (size_t)(1 + nM) * D * 8);
        memcpy(v_out + (size_t)l * D, ch, D * 8);                                    // "v is equal to c[0]" (rgchk.rs:123)
    }
    LFP_MARK(c, "range check: evaluations");
    for (u32 l = 0; l < L; l++) {   // absorb_evaluations (rgchk.rs:333-338): a as constants, then c
        for (u32 i = 0; i < 1 + nM; i++) tr->absorb_const(a_out[(size_t)l * (1 + nM) + i]);
        tr->absorb_ring(c_out + (size_t)l * (1 + nM) * D, 1 + nM);
    }
    return LFPLUS_OK;
}
}  // namespace
extern "C" int lfplus_range_check(lfplus_ctx *const *ctxs, uint32_t L, lfplus_transcript *tr, uint32_t nM, const uint32_t *const *rowptr,
                                  const uint32_t *const *col, const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out,
                                  uint64_t *v_out, uint64_t *a_out, uint64_t *bb_out, uint64_t *c_out) {
    if (!ctxs || !L || !ctxs[0]) return LFPLUS_E_ARG;
    lfplus_ctx *c = ctxs[0];
    if (!tr || !r_out || !msgs || !e_out || !b_out || !v_out || !a_out || !bb_out || !c_out)
        return fail(c, LFPLUS_E_ARG, "lfplus_range_check: bad arguments");
    if (!c->n || (c->n & (c->n - 1))) return fail(c, LFPLUS_E_ARG, "lfplus_range_check: n must be a power of two");
    HIPCHK(c, hipSetDevice(c->device));
    PoolScope pool_scope(c);
    MatHold M;
    int rc = M.get(c, c->n, nM, rowptr, col, val);
    if (rc) return rc;
    ScOut so;
    return range_check_core(ctxs, L, tr, M, r_out, msgs, e_out, b_out, v_out, a_out, bb_out, c_out, so);
}

// ---- verifiers (host only: no GPU, no context) ------------------------------------------------------------------------------------------
namespace {
u64 ev_poly(const u64 *r, u64 x) { u64 acc = 0, pw = 1; for (int i = 0; i < D; i++) { acc = fadd(acc, fmul(r[i] % P, pw)); pw = fmul(pw, x); } return acc; }   // setchk.rs:47-59
u64 eq_eval(const u64 *x, const u64 *y, u32 nv) {
    u64 r = 1;
    for (u32 i = 0; i < nv; i++) { const u64 xy = fmul(x[i], y[i]); r = fmul(r, fadd(fsub(fsub(fadd(xy, xy), x[i]), y[i]), 1)); }
    return r;
}
// ct(psi b) with psi = sum_{0<i<8} i (X^i - X^(16-i)): the element with ct(psi exp(a)) = a for -8 < a < 8 (LatticeFold+ Lemma 2.2)
u64 ct_psi(const u64 *b) {
    u64 acc = 0;
    for (int i = 1; i < D / 2; i++) { acc = fadd(acc, fmul((u64)i, b[i] % P)); acc = fsub(acc, fmul((u64)i, b[D - i] % P)); }
    return acc;
}
// verify_as_subprotocol (latticefold utils/sumcheck.rs:84-104) for constant-polynomial messages
int sumcheck_verify(lfplus_transcript *tr, u32 nv, u32 deg, const u64 *msgs, u64 *point, u64 *expected) {
    tr->absorb_const(nv);
    tr->absorb_const(deg);
    u64 cur = 0;
    for (u32 rnd = 0; rnd < nv; rnd++) {
        const u64 *m = msgs + (size_t)rnd * (deg + 1) * D;
        tr->absorb_ring(m, deg + 1);
        const u64 r = tr->challenge();
        tr->absorb_const(r);
        point[rnd] = r;
        u64 y[8];
        for (u32 x = 0; x <= deg; x++) {
            for (int cidx = 1; cidx < D; cidx++) if (m[x * D + cidx]) return 2;
            y[x] = m[x * D] % P;
        }
        if (fadd(y[0], y[1]) != cur) return 1;
        u64 res = 0;   // interpolate_uni_poly (sumcheck/verifier.rs:141-257)
        for (u32 i = 0; i <= deg; i++) {
            u64 num = 1, den = 1;
            for (u32 j = 0; j <= deg; j++) if (j != i) { num = fmul(num, fsub(r, j)); den = fmul(den, fsub(i, j)); }
            res = fadd(res, fmul(y[i], fmul(num, fpow(den, P - 2))));
        }
        cur = res;
    }
    *expected = cur;
    return 0;
}
}  // namespace

// Out::verify (setchk.rs:266-340).  LFPLUS_OK = accepted, LFPLUS_E_REJECT with *stage = 1 / 2 (sumcheck), 3 (final evaluation)
extern "C" int lfplus_set_check_verify(lfplus_transcript *tr, uint32_t nvars, uint32_t nmat, uint32_t ncols, uint32_t nvec, uint32_t nM, const uint64_t *msgs,
                                       const uint64_t *e, const uint64_t *b, uint64_t *r_out, int *stage) {
    if (!tr || !nmat || !msgs || !e || (nvec && !b) || !r_out || nvars < 1 || nvars > 32) return LFPLUS_E_ARG;
    if (nmat > 4096 || !ncols || ncols > 64 || nvec > 4096 || nM > 64) return LFPLUS_E_ARG;   // the parameter envelope of the provers (array lengths are the caller's contract: lfplus.h)
    const u32 ncl = nmat + nvec;
    std::vector<u64> cs((size_t)ncl * nvars), beta(ncl), alpha(ncl);
    for (u32 i = 0; i < ncl; i++) {
        for (u32 j = 0; j < nvars; j++) cs[(size_t)i * nvars + j] = tr->challenge();
        beta[i] = tr->challenge();
        alpha[i] = tr->challenge();
    }
    const u64 rc = nmat > 1 ? tr->challenge() : 1;
    u64 v;
    int st = sumcheck_verify(tr, nvars, 3, msgs, r_out, &v);
    if (!st) {
        tr->absorb_ring(e, (size_t)(1 + nM) * nmat * ncols);
        tr->absorb_ring(b, nvec);
        u64 ver = 0, rcp = 1;
        for (u32 i = 0; i < nmat; i++) {
            const u64 eq = eq_eval(&cs[(size_t)i * nvars], r_out, nvars), b2 = fmul(beta[i], beta[i]);
            u64 sum = 0, ap = 1;
            for (u32 j = 0; j < ncols; j++) {
                const u64 *ej = e + ((size_t)i * ncols + j) * D;
                const u64 e1 = ev_poly(ej, beta[i]), e2 = ev_poly(ej, b2);
                sum = fadd(sum, fmul(fsub(fmul(e1, e1), e2), ap));
                ap = fmul(ap, alpha[i]);
            }
            ver = fadd(ver, fmul(fmul(eq, sum), rcp));
            rcp = fmul(rcp, rc);
        }
        for (u32 i = 0; i < nvec; i++) {
            const u32 kk = nmat + i;
            const u64 eq = eq_eval(&cs[(size_t)kk * nvars], r_out, nvars), b2 = fmul(beta[kk], beta[kk]);
            const u64 e1 = ev_poly(b + (size_t)i * D, beta[kk]), e2 = ev_poly(b + (size_t)i * D, b2);
            ver = fadd(ver, fmul(fmul(fmul(eq, alpha[kk]), fsub(fmul(e1, e1), e2)), rcp));
            rcp = fmul(rcp, rc);
        }
        if (ver != v) st = 3;
    }
    if (stage) *stage = st;
    return st ? LFPLUS_E_REJECT : LFPLUS_OK;
}
// Dcom::verify (rgchk.rs:193-258): stages 1..3 set check, 4 ct(psi b) != a, 5 ct(psi sum_i d'^i u_i) != v / c
extern "C" int lfplus_range_check_verify(lfplus_transcript *tr, uint32_t nvars, uint32_t L, uint32_t k, uint32_t nM, const uint64_t *msgs, const uint64_t *e,
                                         const uint64_t *b, const uint64_t *v, const uint64_t *a, const uint64_t *bb, const uint64_t *cc, uint64_t *r_out, int *stage) {
    if (!tr || !L || !k || !msgs || !e || !b || !v || !a || !bb || !cc || !r_out) return LFPLUS_E_ARG;
    if (L > 256 || k > 16 || nM > 64) return LFPLUS_E_ARG;
    int st = 0;
    int rc = lfplus_set_check_verify(tr, nvars, L * k, D, L, nM, msgs, e, b, r_out, &st);
    if (rc == LFPLUS_E_ARG) return rc;
    if (!st) {
        for (u32 l = 0; l < L; l++) {
            for (u32 i = 0; i < 1 + nM; i++) tr->absorb_const(a[(size_t)l * (1 + nM) + i]);
            tr->absorb_ring(cc + (size_t)l * (1 + nM) * D, 1 + nM);
        }
        for (u32 l = 0; l < L && !st; l++) {
            for (u32 i = 0; i < 1 + nM && !st; i++)
                if (ct_psi(bb + ((size_t)l * (1 + nM) + i) * D) != a[(size_t)l * (1 + nM) + i] % P) st = 4;
            for (u32 ni = 0; ni < 1 + nM && !st; ni++)
                for (u32 t = 0; t < (u32)D && !st; t++) {
                    u64 uc[D] = {0}, pw = 1;
                    for (u32 i = 0; i < k; i++) {
                        const u64 *u = e + ((((size_t)ni * L * k) + (size_t)k * l + i) * D + t) * D;
                        for (int x = 0; x < D; x++) uc[x] = fadd(uc[x], fmul(u[x] % P, pw));
                        pw = fmul(pw, D / 2);
                    }
                    const u64 want = ni == 0 ? v[(size_t)l * D + t] : cc[((size_t)l * (1 + nM) + ni) * D + t];
                    if (ct_psi(uc) != want % P) st = 5;
                }
        }
    }
    if (stage) *stage = st;
    return st ? LFPLUS_E_REJECT : LFPLUS_OK;
}

// ---- Cm::prove / CmProof::verify (cm.rs:56-347 / 349-580) ----------------------------------------------------------------------------------
namespace {
// negacyclic product mod X^16 + 1 of canonical elements, accumulated: acc += a * b
void rmul_acc(u64 *acc, const u64 *a, const u64 *b) {
    for (int i = 0; i < D; i++) {
        if (!a[i]) continue;
        for (int j = 0; j < D; j++) {
            const u64 pr = fmul(a[i], b[j]);
            if (i + j < D) acc[i + j] = fadd(acc[i + j], pr);
            else acc[i + j - D] = fsub(acc[i + j - D], pr);
        }
    }
}
void rscale_acc(u64 *acc, const u64 *e, u64 s) { for (int i = 0; i < D; i++) acc[i] = fadd(acc[i], fmul(e[i] % P, s)); }
// tensor (utils.rs:68-83)
std::vector<u64> tensor(const u64 *c, u32 nv) {
    std::vector<u64> t(1, 1);
    for (u32 j = 0; j < nv; j++) {
        std::vector<u64> nx(t.size() * 2);
        for (size_t i = 0; i < t.size(); i++) { nx[2 * i] = fmul(t[i], fsub(1, c[j])); nx[2 * i + 1] = fmul(t[i], c[j]); }   // tensor_product(result, [1 - c_j, c_j]): the new factor is the fast index (KAT utils.rs:118-131)
        t.swap(nx);
    }
    return t;
}
struct CmChallenges {
    u64 s[3][D];
    std::vector<u64> sp;       // k*16 short ring elements
    u64 cz[2][32];
    u32 logk;
};
// the challenges both sides draw after the range check (cm.rs:66-80 / 366-381); comh is absorbed between s' and c
void cm_challenges(lfplus_transcript *tr, u32 k, u32 kappa, const u64 *comh, u32 L, CmChallenges &ch) {
    auto sc = [&](u64 *o) {
        uint8_t bs[D];
        tr->squeeze_bytes(D, bs);
        for (int i = 0; i < D; i++) { const int v = (int)bs[i] - 128; o[i] = v >= 0 ? (u64)v : P - (u64)(-v); }
    };
    for (int i = 0; i < 3; i++) sc(ch.s[i]);
    ch.sp.resize((size_t)k * D * D);
    for (u32 i = 0; i < k * D; i++) sc(&ch.sp[(size_t)i * D]);
    (void)comh; (void)L; (void)kappa;
}
void cm_c_challenges(lfplus_transcript *tr, u32 kappa, CmChallenges &ch) {
    ch.logk = 0;
    while (((u32)1 << ch.logk) < kappa) ch.logk++;
    for (int z = 0; z < 2; z++) for (u32 j = 0; j < ch.logk; j++) ch.cz[z][j] = tr->challenge();
}
// calculate_t_z (cm.rs:593-603): tensor(c) (x) s' (x) (1, d', .., d'^(l-1)) (x) (1, X, .., X^15), zero padded to n ring elements; the reference
// panics when it does not fit ("t0 too large!")
bool calc_t(const u64 *cz, u32 logk, const std::vector<u64> &sp, u32 kd, u32 ell, size_t n, std::vector<u64> &out) {
    const size_t tl = (size_t)1 << logk;
    if (tl * kd * ell * D > n) return false;
    const std::vector<u64> tc = tensor(cz, logk);
    out.assign(tl * kd * ell * D * D, 0);   // the non-zero prefix only: the other n - tl kd l d entries of t(z) are zero
    for (size_t a = 0; a < tl; a++)
        for (u32 b = 0; b < kd; b++) {
            u64 pw = 1;
            for (u32 i = 0; i < ell; i++) {
                const u64 sc = fmul(tc[a], pw);
                u64 e[D];                                   // tensor_c[a] d'^i s'[b]; its 16 rotations X^m e follow without products
                for (int t = 0; t < D; t++) e[t] = fmul(sp[(size_t)b * D + t], sc);
                for (int m = 0; m < D; m++) {
                    u64 *o = &out[(((a * kd + b) * ell + i) * D + m) * D];
                    for (int t = 0; t < D; t++) {
                        if (t + m < D) o[t + m] = e[t]; else o[t + m - D] = fsub(0, e[t]);
                    }
                }
                pw = fmul(pw, D / 2);
            }
        }
    return true;
}
// CmProof::x (cm.rs:545-580): cm_g = comh + s0 C_Mf + s1 cm_mtau + s2 cm_f, vo = e[4q + 3] + s0 e[4q] + s1 e[4q + 1] + s2 e[4q + 2] at ro_a and ro_b
void cm_x(const CmChallenges &ch, u32 L, u32 kappa, u32 nM, const u64 *const *fcoms, const u64 *comh, const u64 *ea, const u64 *eb, u64 *cm_g, u64 *vo) {
    const u32 per = 4 + 4 * nM;
    for (u32 l = 0; l < L; l++) {
        for (u32 i = 0; i < kappa; i++) {
            u64 *o = cm_g + ((size_t)l * kappa + i) * D;
            memcpy(o, comh + ((size_t)l * kappa + i) * D, D * 8);
            rmul_acc(o, ch.s[0], fcoms[l] + ((size_t)1 * kappa + i) * D);
            rmul_acc(o, ch.s[1], fcoms[l] + ((size_t)2 * kappa + i) * D);
            rmul_acc(o, ch.s[2], fcoms[l] + ((size_t)0 * kappa + i) * D);
        }
        for (u32 q = 0; q < 1 + nM; q++)
            for (int pass = 0; pass < 2; pass++) {
                const u64 *e4 = (pass ? eb : ea) + ((size_t)l * per + 4 * q) * D;
                u64 *o = vo + (((size_t)l * (1 + nM) + q) * 2 + pass) * D;
                memcpy(o, e4 + 3 * D, D * 8);
                rmul_acc(o, ch.s[0], e4); rmul_acc(o, ch.s[1], e4 + D); rmul_acc(o, ch.s[2], e4 + 2 * D);
            }
    }
}
}  // namespace

// Cm::prove on L resident instances (as lfplus_range_check: ctxs[l] holds f_l, the commitment matrix and the from_f results; `ell` is
// DecompParameters::l).  Outputs (host): the range check's r .. c exactly as lfplus_range_check writes them; comh L x kappa ring elements; the two
// sumcheck proofs pa / pb (nvars x 3 ring elements); their evaluations ea / eb (L x (4 + 4 nM) ring elements, the reference's table order); the
// folded instance x = (cm_g L x kappa, ro = ro_a | ro_b 2 x nvars words, vo L x (1 + nM) x 2).  The folded witness g_l stays on the device in
// ctxs[l] (lfplus_cm_read_g); g_out, when not null, receives L x n ring elements.
extern "C" int lfplus_cm_prove(lfplus_ctx *const *ctxs, uint32_t L, lfplus_transcript *tr, uint32_t ell, uint32_t nM, const uint32_t *const *rowptr,
                               const uint32_t *const *col, const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out,
                               uint64_t *v_out, uint64_t *a_out, uint64_t *bb_out, uint64_t *c_out, uint64_t *comh, uint64_t *pa, uint64_t *pb, uint64_t *ea,
                               uint64_t *eb, uint64_t *cm_g, uint64_t *ro, uint64_t *vo, uint64_t *g_out) {
    if (!ctxs || !L || !ctxs[0]) return LFPLUS_E_ARG;
    lfplus_ctx *c = ctxs[0];
    if (!tr || !ell || !r_out || !msgs || !e_out || !b_out || !v_out || !a_out || !bb_out || !c_out || !comh || !pa || !pb || !ea || !eb || !cm_g || !ro || !vo)
        return fail(c, LFPLUS_E_ARG, "lfplus_cm_prove: bad arguments");
    const size_t n = c->n;
    if (!n || (n & (n - 1))) return fail(c, LFPLUS_E_ARG, "lfplus_cm_prove: n must be a power of two");
    u32 nvars = 0;
    while (((size_t)1 << nvars) < n) nvars++;
    const u32 k = c->k, kappa = c->kappa;
    for (u32 l = 0; l < L; l++)
        if (!ctxs[l] || ctxs[l]->kappa != kappa) return fail(c, LFPLUS_E_ARG, "lfplus_cm_prove: instances of different shapes");
    HIPCHK(c, hipSetDevice(c->device));
    PoolScope pool_scope(c);
    MatHold M;
    int rc = M.get(c, n, nM, rowptr, col, val);
    if (rc) return rc;
    ScOut so;
    const size_t nl = (size_t)c->nloc, row0 = (size_t)c->row0;       // the rank's rows (= n, 0 unsharded)
    const bool shd = c->sharded();
    // tables of the sumcheckers.  Scalars (Montgomery): eq(r, .) | tau_l.  Ring (canonical): per instance m_tau, f, h, then per matrix M tau, M m_tau, M f, M h; then t0, t1
    // Sharded: every table holds the rank's nl rows; the M_q x rows read x at arbitrary columns, so their inputs are whole vectors -- tau, m_tau and f are whole
    // on every rank already, h is all-gathered (n ring elements per instance: the one large exchange of Cm::prove)
    const u32 per = 4 + 4 * nM, nring = L * (per - 1), nS = 1 + L, nR = nring + 2;
    // The sumcheckers run over the BATCHED tables (eq, V | U, Z: lfp_rgchk.hip, k_cm_combine): kS = kR = 2 tables per round instead of nS and nR.
    // LFPLUS_CM_FULL=1 keeps the rounds over all the instance tables (the reference's own shape; the parity tests run both)
    const bool cm_full = getenv("LFPLUS_CM_FULL") != nullptr;      // (read per call: the tests flip it)
    const bool batched = !cm_full;
    const u32 kS = batched ? 2 : nS, kR = batched ? 2 : nR;
    DevBuf S0, R0, Sw[2], Rw[2], rcpd, part, tauring, mtring, hwhole, Sg, Rg, S2, R2, eqro, evpart;
    const u32 nb0 = lfp::cm_round_blocks(nl / 2);
    u64 *S = nullptr, *R = nullptr;
    // Everything of the tables that needs no challenge of Cm::prove -- eq(r, .), tau, m_tau, f and their products with the M_q: three quarters of the table work --
    // is enqueued from inside the range check, ahead of the host's absorb of the set check's evaluations (before_absorb); h, M_q h, t0 and t1 follow below
    // Compact instance tables (lfp_rgchk.hip, k_cm_combine_c): m_tau stays the exponent bytes from_f left, M_q tau a column of scalars -- valid when every M_q has
    // constant coefficients; the batched, unsharded form only (LFPLUS_CM_DENSE=1: every table as ring elements)
    bool compact = batched && !shd && L <= 8 && L * nM <= 64 && nring <= 64 && !getenv("LFPLUS_CM_DENSE");
    for (u32 q = 0; q < nM; q++) compact = compact && M[q].const_coef;
    DevBuf mtsb;
    lfp::CmCompact cc = {};
    lfp::CmTabList dense_list = {};
    u32 ndense = 0;
    auto tables_early = [&]() -> int {
        if (S0.alloc((size_t)nS * nl * 8) || R0.alloc((size_t)nR * nl * D * 8) || Sw[0].alloc((size_t)kS * (nl / 2) * 8) || Sw[1].alloc((size_t)kS * (nl / 4 + 1) * 8) ||
            Rw[0].alloc((size_t)kR * (nl / 2) * D * 8) || Rw[1].alloc((size_t)kR * (nl / 4 + 1) * D * 8) || rcpd.alloc((size_t)(L * per + 2) * 8) ||
            part.alloc((size_t)nb0 * 48 * 8) || (nM && tauring.alloc(n * D * 8)) || (nM && shd && (mtring.alloc(n * D * 8) || hwhole.alloc(n * D * 8))) ||
            (shd && (Sg.alloc((size_t)kS * c->world * 8) || Rg.alloc((size_t)kR * c->world * D * 8))) ||
            (batched && (S2.alloc((size_t)2 * nl * 8) || R2.alloc((size_t)2 * nl * D * 8) || eqro.alloc(nl * 8) ||
                         evpart.alloc(std::max((size_t)lfp::cm_eval_chunks(nl) * nring * D, (size_t)lfp::eval_chunks(nl) * 4) * 8))))
            return fail(c, LFPLUS_E_HIP, "hipMalloc (Cm tables)");
        S = S0.as<u64>(); R = R0.as<u64>();
        HIPCHK(c, hipMemcpyAsync(S, so.eqr.as<u64>() + row0, nl * 8, hipMemcpyDeviceToDevice, c->st));
        if (compact) {
            if (nM && mtsb.alloc((size_t)L * nM * nl * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (Cm scalar tables)");
            cc.mts = mtsb.as<u64>(); cc.ldm = nl;
            for (u32 l = 0; l < L; l++) cc.mtau[l] = ctxs[l]->mtau;
            for (u32 l = 0; l < L; l++)
                for (u32 j = 1; j < per; j++)
                    if (j != 1 && !(j >= 4 && ((j - 4) & 3) == 0)) dense_list.idx[ndense++] = (uint16_t)(l * (per - 1) + j - 1);
        }
        for (u32 l = 0; l < L; l++) {
            u64 *base = R + (size_t)l * (per - 1) * nl * D;
            if (compact) {      // f as a ring table; M_q tau as scalars, M_q m_tau from the exponent bytes, M_q f
                lfp::launch_to_mont(ctxs[l]->tau, nl, S + (size_t)(1 + l) * nl, c->st);
                HIPCHK(c, hipMemcpyAsync(base + nl * D, ctxs[l]->f, nl * D * 8, hipMemcpyDeviceToDevice, c->st));
                for (u32 q = 0; q < nM; q++) {
                    const LfpMatrix &m = M[q];
                    u64 *mq = base + (size_t)(3 + 4 * q) * nl * D;
                    lfp::launch_spmv_scalar_const(m.rowptr, m.col, m.spmv_vals(), ctxs[l]->tau, nl, mtsb.as<u64>() + ((size_t)l * nM + q) * nl, c->st);
                    lfp::launch_spmv_mono_const(m.rowptr, m.col, m.spmv_vals(), ctxs[l]->mtau, nl, mq + (size_t)1 * nl * D, c->st);
                    lfp::launch_spmv_ring(m.rowptr, m.col, m.spmv_vals(), ctxs[l]->f, nl, mq + (size_t)2 * nl * D, c->st, 1);
                }
                continue;
            }
            lfp::launch_to_mont(ctxs[l]->tau + row0, nl, S + (size_t)(1 + l) * nl, c->st);
            lfp::launch_cm_materialize(ctxs[l]->mtau + row0, nullptr, nl, base, c->st);
            HIPCHK(c, hipMemcpyAsync(base + nl * D, ctxs[l]->f + row0 * D, nl * D * 8, hipMemcpyDeviceToDevice, c->st));
            if (!nM) continue;
            lfp::launch_cm_materialize(nullptr, ctxs[l]->tau, n, tauring.as<u64>(), c->st);
            const u64 *xin[2] = {base, base + nl * D};      // m_tau, f as whole vectors
            if (shd) {
                lfp::launch_cm_materialize(ctxs[l]->mtau, nullptr, n, mtring.as<u64>(), c->st);
                xin[0] = mtring.as<u64>(); xin[1] = ctxs[l]->f;
            }
            for (u32 q = 0; q < nM; q++) {
                const LfpMatrix &m = M[q];
                u64 *mq = base + (size_t)(3 + 4 * q) * nl * D;
                lfp::launch_spmv_ring(m.rowptr + row0, m.col, m.spmv_vals(), tauring.as<u64>(), nl, mq, c->st, m.const_coef);
                for (int j = 0; j < 2; j++) lfp::launch_spmv_ring(m.rowptr + row0, m.col, m.spmv_vals(), xin[j], nl, mq + (size_t)(1 + j) * nl * D, c->st, m.const_coef);
            }
        }
        return LFPLUS_OK;
    };
    rc = range_check_core(ctxs, L, tr, M, r_out, msgs, e_out, b_out, v_out, a_out, bb_out, c_out, so, tables_early);
    if (rc) return rc;
    CmChallenges ch;
    cm_challenges(tr, k, kappa, nullptr, L, ch);
    LFP_MARK(c, "cm: challenges");
    // h_l = sum_ki M_f[ki] s'_ki (cm.rs:82-103) on the device; comh_l = sum_ki comM_f[ki] s'_ki (:105-126) on the host (k kappa 256 products)
    std::vector<int32_t> spi((size_t)k * D * D);
    for (size_t i = 0; i < spi.size(); i++) spi[i] = ch.sp[i] > P / 2 ? -(int32_t)(P - ch.sp[i]) : (int32_t)ch.sp[i];
    DevBuf spd;
    std::vector<std::unique_ptr<DevBuf>> h(L);
    if (spd.alloc(spi.size() * 4)) return fail(c, LFPLUS_E_HIP, "hipMalloc (s')");
    HIPCHK(c, hipMemcpyAsync(spd.p, spi.data(), spi.size() * 4, hipMemcpyHostToDevice, c->st));
    std::vector<std::vector<u64>> fcoms(L, std::vector<u64>((size_t)3 * kappa * D));
    std::vector<u64> comMf((size_t)k * kappa * D * D);
    for (u32 l = 0; l < L; l++) {
        h[l].reset(new DevBuf);
        if (h[l]->alloc(nl * D * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (h)");
        lfp::launch_cm_h(ctxs[l]->Df, nl, k, spd.as<int32_t>(), h[l]->as<u64>(), c->st);
        HIPCHK(c, hipMemcpyAsync(comMf.data(), ctxs[l]->comMf, comMf.size() * 8, hipMemcpyDeviceToHost, c->st));
        // fcoms[l] = cm_f | C_Mf | cm_mtau; the context keeps cm_f behind comM_f and C_Mf | cm_mtau in coms
        HIPCHK(c, hipMemcpyAsync(fcoms[l].data(), ctxs[l]->comMf + comMf.size(), (size_t)kappa * D * 8, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipMemcpyAsync(fcoms[l].data() + (size_t)kappa * D, ctxs[l]->coms, (size_t)2 * kappa * D * 8, hipMemcpyDeviceToHost, c->st));
        HIPCHK(c, hipStreamSynchronize(c->st));
        for (u32 i = 0; i < kappa; i++) {
            u64 *o = comh + ((size_t)l * kappa + i) * D;
            memset(o, 0, D * 8);
            for (u32 ki = 0; ki < k; ki++)
                for (int j = 0; j < D; j++) rmul_acc(o, &comMf[((((size_t)ki * kappa + i) * D) + j) * D], &ch.sp[((size_t)ki * D + j) * D]);
        }
    }
    tr->absorb_ring(comh, (size_t)L * kappa);
    cm_c_challenges(tr, kappa, ch);
    LFP_MARK(c, "cm: h, com_h, c challenges");
    // the challenge-dependent rest of the tables: h_l and the M_q h_l
    for (u32 l = 0; l < L; l++) {
        u64 *base = R + (size_t)l * (per - 1) * nl * D;
        HIPCHK(c, hipMemcpyAsync(base + 2 * nl * D, h[l]->p, nl * D * 8, hipMemcpyDeviceToDevice, c->st));
        const u64 *hin = base + 2 * nl * D;
        if (nM && shd) {
            int rcg = lfp_allgather_dev(c, h[l]->as<u64>(), hwhole.as<u64>(), nl * D);
            if (rcg) return rcg;
            hin = hwhole.as<u64>();
        }
        for (u32 q = 0; q < nM; q++) {
            const LfpMatrix &m = M[q];
            lfp::launch_spmv_ring(m.rowptr + row0, m.col, m.spmv_vals(), hin, nl, base + (size_t)(3 + 4 * q + 3) * nl * D, c->st, m.const_coef);
        }
    }
    HIPCHK(c, hipMemsetAsync(R + (size_t)nring * nl * D, 0, (size_t)2 * nl * D * 8, c->st));   // t0 | t1: zero beyond the prefix the host computed
    // t(z) on the host (0.8 ms at k = 4) while the device builds the instance tables enqueued above
    std::vector<u64> t0, t1;
    if (!calc_t(ch.cz[0], ch.logk, ch.sp, k * D, ell, n, t0) || !calc_t(ch.cz[1], ch.logk, ch.sp, k * D, ell, n, t1))
        return fail(c, LFPLUS_E_ARG, "lfplus_cm_prove: t0 too large (kappa' * k d * l * d > n; the reference panics, cm.rs:601)");
    LFP_MARK(c, "cm: table launches, t(z) on the host");
    for (int z = 0; z < 2; z++) {        // the rank's rows of the non-zero prefix
        const std::vector<u64> &tz = z ? t1 : t0;
        const size_t pre = tz.size() / D;
        if (pre > row0) {
            const size_t cnt = std::min(pre - row0, nl);
            HIPCHK(c, hipMemcpyAsync(R + (size_t)(nring + z) * nl * D, tz.data() + row0 * D, cnt * D * 8, hipMemcpyHostToDevice, c->st));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    LFP_MARK(c, "cm: tables");
    // the two sumcheckers (cm.rs:201-347): same tables, different batching challenge rc; degree 2, ring-valued messages
    const lfp::CmDesc desc = {L, nM};
    std::vector<u64> rcps((size_t)L * per + 2), evh((size_t)nR * D + nS);
    u64 *hpart = c->pin((size_t)nb0 * 48);   // block partials land in mapped host memory
    if (!hpart) return fail(c, LFPLUS_E_HIP, "hipHostMalloc (round partials)");
    for (int pass = 0; pass < 2; pass++) {
        const u64 rcv = tr->challenge();
        u64 pw = 1;
        for (size_t i = 0; i < rcps.size(); i++) { rcps[i] = to_mont(pw); pw = fmul(pw, rcv); }
        HIPCHK(c, hipMemcpyAsync(rcpd.p, rcps.data(), rcps.size() * 8, hipMemcpyHostToDevice, c->st));
        u64 *proof = pass ? pb : pa, *evs = pass ? eb : ea, *rop = ro + (size_t)pass * nvars;
        tr->absorb_const(nvars);
        tr->absorb_const(2);
        const u64 *Sc = S, *Rc = R;
        if (batched) {
            if (compact) lfp::launch_cm_combine_c(S, nl, R, nl, nl, desc, rcpd.as<u64>(), cc, S2.as<u64>(), R2.as<u64>(), nl, c->st);
            else
            lfp::launch_cm_combine(S, nl, R, nl, nl, desc, rcpd.as<u64>(), S2.as<u64>(), R2.as<u64>(), nl, c->st);
            Sc = S2.as<u64>(); Rc = R2.as<u64>();
        }
        size_t ld = nl, len = nl;
        int w = 0;
        bool dist = shd;
        // unsharded: fix_variables is deferred into the next round's kernel (k_cm_round_fused: one read of the previous tables, one write of the fixed ones, one
        // launch per round); `pending` = the challenge whose fix has not been applied to Sc / Rc yet.  LFPLUS_CM_UNFUSED=1: separate k_cm_fix passes.
        const bool unfused = getenv("LFPLUS_CM_UNFUSED") != nullptr;   // (read per call, like the other switches)
        const bool fuse = !shd && !unfused;
        bool pending = false;
        u64 rpend = 0;
        for (u32 rnd = 0; rnd < nvars; rnd++) {
            if (dist && len == 1) {      // one entry per table and rank left: gather (entry index = rank), finish replicated
                int rcg = gather_tables(c, Sc, ld, kS, 1, Sg.as<u64>());
                if (!rcg) rcg = gather_tables(c, Rc, ld, kR, D, Rg.as<u64>());
                if (rcg) return rcg;
                Sc = Sg.as<u64>(); Rc = Rg.as<u64>();
                ld = len = (size_t)c->world;
                dist = false;
            }
            const size_t half = pending ? len / 4 : len / 2;       // pairs this round evaluates
            const u32 nb = lfp::cm_round_blocks(half);
            auto tA = std::chrono::steady_clock::now();
            // up to 256 blocks write their partial sums straight into mapped host memory (the host adds them); the large rounds need more workgroups than that to
            // reach the HBM rate (6.3 GB of tables in round 0 at 2^20 rows: 5.4 ms with 256 blocks, 1.8 ms with 4096) and add theirs on the device
            const bool dev_sum = nb > LFP_HOST_SUM_BLOCKS;
            u64 *pdst = dev_sum ? part.as<u64>() : c->hpin_dev;
            if (pending) {
                const size_t ldo = w == 0 ? nl / 2 : nl / 4 + 1;
                if (batched) lfp::launch_cm2_round_fused(Sc, ld, Rc, ld, half, to_mont(rpend), Sw[w].as<u64>(), Rw[w].as<u64>(), ldo, pdst, c->st);
                else lfp::launch_cm_round_fused(Sc, ld, Rc, ld, half, desc, rcpd.as<u64>(), to_mont(rpend), Sw[w].as<u64>(), Rw[w].as<u64>(), ldo, pdst, c->st);
                Sc = Sw[w].as<u64>(); Rc = Rw[w].as<u64>(); ld = ldo; w ^= 1;
                len /= 2;
                pending = false;
            } else if (batched)
                lfp::launch_cm2_round(Sc, ld, Rc, ld, half, pdst, c->st);
            else
                lfp::launch_cm_round(Sc, ld, Rc, ld, half, desc, rcpd.as<u64>(), pdst, c->st);
            if (dev_sum) lfp::launch_reduce(part.as<u64>(), nb, 48, c->hpin_dev, 0, c->kappa, 0, 2, 0, nullptr, c->st);
            HIPCHK(c, hipStreamSynchronize(c->st));
            auto tB = std::chrono::steady_clock::now();
            u64 *m = proof + (size_t)rnd * 3 * D;
            const u32 nbh = dev_sum ? 1u : nb;
            for (int x = 0; x < 3 * D; x++) {
                u64 s = 0;
                for (u32 b = 0; b < nbh; b++) s = fadd(s, hpart[(size_t)b * 48 + x]);
                m[x] = s;              // canonical: every product of the kernel pairs one Montgomery operand with one canonical operand
            }
            if (dist) { int rcx = lfp_xsum(c, m, 3 * D); if (rcx) return rcx; }
            tr->absorb_ring(m, 3);
            const u64 r = tr->challenge();
            tr->absorb_const(r);
            rop[rnd] = r;
            if (fuse && rnd + 1 < nvars) { pending = true; rpend = r; }     // (the last challenge is applied below: the evaluations need the fully fixed tables)
            else if (batched && rnd + 1 == nvars) len = half;               // (batched: the evaluations come from the original tables, nothing reads a last fix)
            else {
                const size_t ldo = w == 0 ? nl / 2 : nl / 4 + 1;
                lfp::launch_cm_fix(Sc, ld, Sw[w].as<u64>(), ldo, 1, kS, half, to_mont(r), c->st);
                lfp::launch_cm_fix(Rc, ld, Rw[w].as<u64>(), ldo, D, kR, half, to_mont(r), c->st);
                Sc = Sw[w].as<u64>(); Rc = Rw[w].as<u64>(); ld = ldo; w ^= 1;
                len = half;
            }
            if (g_tl.on) fprintf(stderr, "[lfplus]   cm round %2u: gpu+sync %6.1f us, host + fix launches %6.1f us (nb %u)\n", rnd, std::chrono::duration<double, std::micro>(tB - tA).count(),
                                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tB).count(), nb);
        }
        // evals (cm.rs:313-331): every instance table at ro = the fully fixed tables
        DevBuf evd;
        if (evd.alloc(evh.size() * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (evals)");
        if (batched) {     // T(ro) = sum_b eq(ro, b) T(b) over the rank's rows of the ORIGINAL instance tables (ring) and of the tau_l (scalars, Montgomery)
            eq_build_local(c, rop, nvars, eqro.as<u64>());
            if (compact) lfp::launch_cm_evals_c(R, nl, nl, eqro.as<u64>(), nring, dense_list, ndense, cc, L, nM, per, evpart.as<u64>(), evd.as<u64>(), c->st);
            else
            lfp::launch_cm_evals(R, nl, nl, eqro.as<u64>(), nring, evpart.as<u64>(), evd.as<u64>(), c->st);
            for (u32 l = 0; l < L; l++)
                lfp::launch_wdot(eqro.as<u64>(), 1, 1, S + (size_t)(1 + l) * nl, nl, evpart.as<u64>(), evd.as<u64>() + (size_t)nR * D + 1 + l, c->st);
            HIPCHK(c, hipMemcpyAsync(evh.data(), evd.p, evh.size() * 8, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipStreamSynchronize(c->st));
            int rcx = lfp_xsum(c, evh.data(), (size_t)nring * D);
            if (!rcx) rcx = lfp_xsum(c, evh.data() + (size_t)nR * D + 1, L);
            if (rcx) return rcx;
        } else {
            HIPCHK(c, hipMemcpy2DAsync(evd.p, D * 8, Rc, ld * D * 8, D * 8, nR, hipMemcpyDeviceToDevice, c->st));
            HIPCHK(c, hipMemcpy2DAsync(evd.as<u64>() + (size_t)nR * D, 8, Sc, ld * 8, 8, nS, hipMemcpyDeviceToDevice, c->st));
            HIPCHK(c, hipMemcpyAsync(evh.data(), evd.p, evh.size() * 8, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipStreamSynchronize(c->st));
        }
        for (u32 l = 0; l < L; l++) {
            u64 *el = evs + (size_t)l * per * D;
            memset(el, 0, D * 8);
            el[0] = from_mont(evh[(size_t)nR * D + 1 + l]);
            memcpy(el + D, &evh[(size_t)l * (per - 1) * D], (size_t)(per - 1) * D * 8);
        }
        tr->absorb_ring(evs, (size_t)L * per);
        LFP_MARK(c, "cm: sumchecker");
    }
    // g_l = s0 tau + s1 m_tau + s2 f + h (cm.rs:164-181), kept on the device
    lfp::CmShort cs;
    for (int i = 0; i < 3; i++) for (int t = 0; t < D; t++) cs.v[i][t] = ch.s[i][t] > P / 2 ? -(int32_t)(P - ch.s[i][t]) : (int32_t)ch.s[i][t];
    for (u32 l = 0; l < L; l++) {
        lfplus_ctx *cl = ctxs[l];
        if (cl->g_n != nl) {          // the rank's rows of g
            if (cl->g) { cl->own_free(cl->g); cl->g = nullptr; cl->g_n = 0; }
            HIPCHK(c, cl->own_alloc(&cl->g, nl * D * 8));
            cl->g_n = nl;
        }
        lfp::launch_cm_g(cl->tau + row0, cl->mtau + row0, cl->f + row0 * D, h[l]->as<u64>(), nl, cs, cl->g, c->st);
        cl->g_valid = true;
        if (g_out && !shd) HIPCHK(c, hipMemcpyAsync(g_out + (size_t)l * n * D, cl->g, n * D * 8, hipMemcpyDeviceToHost, c->st));
        if (g_out && shd) {
            DevBuf gw;
            if (gw.alloc(n * D * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (g)");
            int rcg = lfp_allgather_dev(c, cl->g, gw.as<u64>(), nl * D);
            if (rcg) return rcg;
            HIPCHK(c, hipMemcpyAsync(g_out + (size_t)l * n * D, gw.p, n * D * 8, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipStreamSynchronize(c->st));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    std::vector<const u64 *> fc(L);
    for (u32 l = 0; l < L; l++) fc[l] = fcoms[l].data();
    cm_x(ch, L, kappa, nM, fc.data(), comh, ea, eb, cm_g, vo);
    LFP_MARK(c, "cm: g, x");
    return LFPLUS_OK;
}
extern "C" int lfplus_cm_read_g(lfplus_ctx *c, uint64_t *g_out) {
    if (!c || !g_out) return LFPLUS_E_ARG;
    if (!c->g || !c->g_n || !c->g_valid)
        return fail(c, LFPLUS_E_ARG, "lfplus_cm_read_g: no folded witness (call lfplus_cm_prove; after lfplus_mlin ctxs[0] holds the SUM of the instances' g as its resident witness)");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->sharded()) {
        HIPCHK(c, hipMemcpy(g_out, c->g, c->g_n * D * 8, hipMemcpyDeviceToHost));
        return LFPLUS_OK;
    }
    PoolScope pool_scope(c);      // sharded: the ranks hold their rows of g -- collective: every rank must make this call
    DevBuf gw;
    if (gw.alloc(c->n * D * 8)) return fail(c, LFPLUS_E_HIP, "hipMalloc (g)");
    int rc = lfp_allgather_dev(c, c->g, gw.as<u64>(), c->nloc * D);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(g_out, gw.p, c->n * D * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return LFPLUS_OK;
}

// CmProof::verify (cm.rs:349-543), host only.  fcoms[l] = cm_f | C_Mf | cm_mtau of instance l (kappa ring elements each).  LFPLUS_OK = accepted and
// cm_g / ro / vo hold the folded instance ComX (cm.rs:545-580) recomputed from the proof; LFPLUS_E_REJECT with *stage = 1..5 (range check), 6 (a
// sumcheck round), 7 (t0 too large), 8 (final evaluation of a sumchecker)
extern "C" int lfplus_cm_verify(lfplus_transcript *tr, uint32_t nvars, uint32_t L, uint32_t k, uint32_t ell, uint32_t kappa, uint32_t nM, const uint64_t *const *fcoms,
                                const uint64_t *msgs, const uint64_t *e, const uint64_t *b, const uint64_t *v, const uint64_t *a, const uint64_t *bb,
                                const uint64_t *cc, const uint64_t *comh, const uint64_t *pa, const uint64_t *pb, const uint64_t *ea, const uint64_t *eb,
                                uint64_t *cm_g, uint64_t *ro, uint64_t *vo, int *stage) {
    if (!tr || !L || !k || !ell || !kappa || !fcoms || !comh || !pa || !pb || !ea || !eb || !cm_g || !ro || !vo || nvars < 1 || nvars > 32) return LFPLUS_E_ARG;
    if (L > 256 || k > 16 || ell > 64 || kappa > 64 || nM > 64) return LFPLUS_E_ARG;
    for (u32 l = 0; l < L; l++) if (!fcoms[l]) return LFPLUS_E_ARG;
    const size_t n = (size_t)1 << nvars;
    std::vector<u64> r(nvars);
    int st = 0;
    int rc = lfplus_range_check_verify(tr, nvars, L, k, nM, msgs, e, b, v, a, bb, cc, r.data(), &st);
    if (rc == LFPLUS_E_ARG) return rc;
    const u32 per = 4 + 4 * nM, z_idx = L * per;
    CmChallenges ch;
    if (!st) {
        cm_challenges(tr, k, kappa, nullptr, L, ch);
        tr->absorb_ring(comh, (size_t)L * kappa);
        cm_c_challenges(tr, kappa, ch);
    }
    std::vector<u64> t0, t1;
    if (!st && (!calc_t(ch.cz[0], ch.logk, ch.sp, k * D, ell, n, t0) || !calc_t(ch.cz[1], ch.logk, ch.sp, k * D, ell, n, t1))) st = 7;
    if (!st) {
        // u[l][q] = sum over the instance's k 16 set-check evaluations of e * s' (cm.rs:383-401); tensor(c_z) . comh_l
        std::vector<u64> u((size_t)L * (1 + nM) * D, 0), tcch((size_t)2 * L * D, 0);
        for (u32 l = 0; l < L; l++)
            for (u32 q = 0; q < 1 + nM; q++)
                for (u32 j = 0; j < k * D; j++) rmul_acc(&u[((size_t)l * (1 + nM) + q) * D], e + (((size_t)q * L * k + (size_t)l * k) * D + j) * D, &ch.sp[(size_t)j * D]);
        for (int z = 0; z < 2; z++) {
            const std::vector<u64> tc = tensor(ch.cz[z], ch.logk);
            for (u32 l = 0; l < L; l++)
                for (u32 i = 0; i < kappa && i < tc.size(); i++) rscale_acc(&tcch[((size_t)z * L + l) * D], comh + ((size_t)l * kappa + i) * D, tc[i]);
        }
        for (int pass = 0; pass < 2 && !st; pass++) {
            const u64 rcv = tr->challenge();
            std::vector<u64> rcps(z_idx + 2);
            rcps[0] = 1;
            for (u32 i = 1; i < z_idx + 2; i++) rcps[i] = fmul(rcps[i - 1], rcv);
            u64 cur[D] = {0};
            for (u32 l = 0; l < L; l++) {
                for (u32 q = 0; q < 1 + nM; q++) {
                    const u32 idx = l * per + 4 * q;
                    const size_t o = (size_t)l * (1 + nM) + q;
                    cur[0] = fadd(cur[0], fmul(a[o] % P, rcps[idx]));
                    rscale_acc(cur, bb + o * D, rcps[idx + 1]);
                    rscale_acc(cur, cc + o * D, rcps[idx + 2]);
                    rscale_acc(cur, &u[o * D], rcps[idx + 3]);
                }
                rscale_acc(cur, &tcch[((size_t)0 * L + l) * D], rcps[z_idx]);
                rscale_acc(cur, &tcch[((size_t)1 * L + l) * D], rcps[z_idx + 1]);
            }
            const u64 *proof = pass ? pb : pa, *evs = pass ? eb : ea;
            u64 *rop = ro + (size_t)pass * nvars;
            // verify_as_subprotocol with ring-valued messages: coefficient-wise, degree 2
            tr->absorb_const(nvars);
            tr->absorb_const(2);
            for (u32 rnd = 0; rnd < nvars && !st; rnd++) {
                const u64 *m = proof + (size_t)rnd * 3 * D;
                tr->absorb_ring(m, 3);
                const u64 x = tr->challenge();
                tr->absorb_const(x);
                rop[rnd] = x;
                // Lagrange weights on the nodes 0, 1, 2
                const u64 inv2 = fpow(2, P - 2), x1 = fsub(x, 1), x2 = fsub(x, 2);
                const u64 w0 = fmul(fmul(x1, x2), inv2), w1 = fsub(0, fmul(x, x2)), w2 = fmul(fmul(x, x1), inv2);
                for (int ci = 0; ci < D; ci++) {
                    const u64 y0 = m[ci] % P, y1 = m[D + ci] % P, y2 = m[2 * D + ci] % P;
                    if (fadd(y0, y1) != cur[ci]) { st = 6; break; }
                    cur[ci] = fadd(fadd(fmul(y0, w0), fmul(y1, w1)), fmul(y2, w2));
                }
            }
            if (st) break;
            // t0, t1 at ro: fold the tables along the point, variable 0 first
            u64 tz[2][D];
            for (int z = 0; z < 2; z++) {
                // only the first tl k d l d entries are non-zero: fold that prefix (an odd length pairs its last entry with a zero)
                size_t nz = std::min(n, ((size_t)1 << ch.logk) * k * D * ell * D);
                std::vector<u64> cur_t((z ? t1 : t0).begin(), (z ? t1 : t0).begin() + nz * D);
                for (u32 j = 0; j < nvars; j++) {
                    const size_t half = (nz + 1) / 2;
                    for (size_t i = 0; i < half; i++)
                        for (int ci = 0; ci < D; ci++) {
                            const u64 lo = cur_t[(2 * i) * D + ci], hi = 2 * i + 1 < nz ? cur_t[(2 * i + 1) * D + ci] : 0;
                            cur_t[i * D + ci] = fadd(lo, fmul(rop[j], fsub(hi, lo)));
                        }
                    nz = half;
                }
                memcpy(tz[z], cur_t.data(), D * 8);
            }
            tr->absorb_ring(evs, (size_t)L * per);
            const u64 eq = eq_eval(r.data(), rop, nvars);
            u64 want[D] = {0};
            for (u32 l = 0; l < L; l++) {
                const u64 *el = evs + (size_t)l * per * D;
                u64 inner[D] = {0}, t[D];
                for (u32 j = 0; j < per; j++) rscale_acc(inner, el + (size_t)j * D, rcps[l * per + j]);
                rscale_acc(want, inner, eq);
                memset(t, 0, sizeof(t)); rmul_acc(t, tz[0], el); rscale_acc(want, t, rcps[z_idx]);
                memset(t, 0, sizeof(t)); rmul_acc(t, tz[1], el); rscale_acc(want, t, rcps[z_idx + 1]);
            }
            if (memcmp(want, cur, sizeof(want))) st = 8;
        }
    }
    if (!st) cm_x(ch, L, kappa, nM, fcoms, comh, ea, eb, cm_g, vo);
    if (stage) *stage = st;
    return st ? LFPLUS_E_REJECT : LFPLUS_OK;
}

// ---- ComR1CS::linearize / ComR1CSProof::verify (r1cs.rs:76-162), DecompProof::verify (decomp.rs:101-123), Mlin::mlin (mlin.rs:42-107) -----------
// The constraint-system matrices (n x n, CSR, ring coefficients), uploaded once: every entry point of this file that takes (rowptr, col, val) uses them when
// rowptr is NULL (nM must then equal their number).  PlusProver keeps A, B, C of the R1CS here for linearize, the range check, Cm::prove and Decomp.
extern "C" int lfplus_set_matrices(lfplus_ctx *c, uint64_t n, uint32_t nM, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val) {
    if (!c) return LFPLUS_E_ARG;
    if (!n || n > (1ull << 32) || nM > 64 || (nM && (!rowptr || !col || !val))) return fail(c, LFPLUS_E_ARG, "lfplus_set_matrices: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    c->drop_mats();
    std::vector<LfpMatrix> fresh(nM);
    for (u32 q = 0; q < nM; q++) {
        int rc = upload_matrix(c, n, rowptr[q], col[q], val[q], fresh[q]);
        if (rc) { for (LfpMatrix &m : fresh) m.release(); return rc; }
    }
    c->mats_ref = std::make_shared<lfplus_ctx::MatsOwner>();
    c->mats_ref->m = fresh;      // the owner releases the device arrays with its last holder; c->mats is the view the kernels take
    c->mats.swap(fresh);
    c->mats_n = n;
    return LFPLUS_OK;
}

// the resident matrices of `from` (same device), not copied; reference-counted like the commitment matrix
extern "C" int lfplus_share_matrices(lfplus_ctx *c, lfplus_ctx *from) {
    if (!c || !from || c == from || c->device != from->device) return fail(c, LFPLUS_E_ARG, "lfplus_share_matrices: bad arguments");
    c->drop_mats();
    c->mats = from->mats;
    c->mats_ref = from->mats_ref;
    c->mats_n = from->mats_n;
    return LFPLUS_OK;
}

// ComR1CS::linearize on the resident witness f (lfplus_set_witness; n = 2^nvars ring elements) and the three R1CS matrices (n x n, CSR, ring
// coefficients; A, B, C in this order).  Outputs: msgs nvars x 4 ring elements (the degree-3 sumcheck), ro (nvars words), evals = v | va | vb | vc
// (f, A f, B f, C f at ro)
extern "C" int lfplus_r1cs_linearize(lfplus_ctx *c, lfplus_transcript *tr, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val,
                                     uint64_t *msgs, uint64_t *ro, uint64_t *evals) {
    if (!c) return LFPLUS_E_ARG;
    if (!tr || !msgs || !ro || !evals) return fail(c, LFPLUS_E_ARG, "lfplus_r1cs_linearize: bad arguments");
    const size_t n = c->nf;
    if (!c->f || n < 2 || (n & (n - 1))) return fail(c, LFPLUS_E_ARG, "lfplus_r1cs_linearize: needs a resident witness of 2^nvars ring elements");
    u32 nvars = 0;
    while (((size_t)1 << nvars) < n) nvars++;
    HIPCHK(c, hipSetDevice(c->device));
    PoolScope pool_scope(c);
    LFP_MARK(c, "(linearize: entry)");
    // Sharded: the witness is whole, the three tables g_q = M_q f and eq(r, .) hold the rank's nl rows; partial round messages are summed over the ranks,
    // the last log2(world) rounds run replicated on gathered tables, v = f(ro) is a partial sum over the rank's rows.
    const size_t nl = c->sharded() ? (size_t)c->nloc : n, row0 = c->sharded() ? (size_t)c->row0 : 0;
    if (c->sharded() && c->n != n) return fail(c, LFPLUS_E_ARG, "lfplus_r1cs_linearize: the witness of a sharded context has the matrix's (global) width");
    DevBuf E[2], G[2], part, small, Eg, Gg;
    const u32 nb0 = lfp::cm_round_blocks(nl / 2);
    if (E[0].alloc(nl * 8) || E[1].alloc(nl / 2 * 8) || G[0].alloc((size_t)3 * nl * D * 8) || G[1].alloc((size_t)3 * (nl / 2) * D * 8) ||
        part.alloc(std::max<size_t>((size_t)nb0 * 64, (size_t)lfp::eval_chunks(n) * D) * 8) || small.alloc(4 * D * 8) ||
        (c->sharded() && (Eg.alloc((size_t)c->world * 8) || Gg.alloc((size_t)3 * c->world * D * 8))))
        return fail(c, LFPLUS_E_HIP, "hipMalloc (linearize tables)");
    LFP_MARK(c, "(linearize: allocs)");
    MatHold M;
    int rcm = M.get(c, n, 3, rowptr, col, val);
    if (rcm) return rcm;
    LFP_MARK(c, "(before linearize)");
    for (u32 q = 0; q < 3; q++) lfp::launch_spmv_ring(M[q].rowptr + row0, M[q].col, M[q].spmv_vals(), c->f, nl, G[0].as<u64>() + (size_t)q * nl * D, c->st, M[q].const_coef);
    std::vector<u64> r(nvars);
    for (u32 j = 0; j < nvars; j++) r[j] = tr->challenge();
    eq_build_local(c, r.data(), nvars, E[0].as<u64>());
    tr->absorb_const(nvars);
    tr->absorb_const(3);
    u64 *hpart = c->pin((size_t)nb0 * 64);
    if (!hpart) return fail(c, LFPLUS_E_HIP, "hipHostMalloc (round partials)");
    // round 0 reads the full tables (stride nl), later rounds ping-pong between the first halves of the two buffers
    const u64 *Ec = E[0].as<u64>(), *Gc = G[0].as<u64>();
    size_t ld = nl, len = nl;
    int w = 1;
    bool dist = c->sharded();
    // unsharded: fix_variables is deferred into the next round's kernel (k_r1cs_round_fused), as in Cm::prove
    const bool unfused = getenv("LFPLUS_CM_UNFUSED") != nullptr;   // (read per call, like the other switches)
    const bool fuse = !c->sharded() && !unfused;
    bool pending = false;
    u64 xpend = 0;
    for (u32 rnd = 0; rnd < nvars; rnd++) {
        if (dist && len == 1) {      // one entry per table and rank left: gather (entry index = rank), finish replicated
            int rcg = gather_tables(c, Ec, ld, 1, 1, Eg.as<u64>());
            if (!rcg) rcg = gather_tables(c, Gc, ld, 3, D, Gg.as<u64>());
            if (rcg) return rcg;
            Ec = Eg.as<u64>(); Gc = Gg.as<u64>();
            ld = len = (size_t)c->world;
            dist = false;
        }
        const size_t half = pending ? len / 4 : len / 2;
        const u32 nb = lfp::cm_round_blocks(half);
        auto tA = std::chrono::steady_clock::now();
        const bool dev_sum = nb > LFP_HOST_SUM_BLOCKS;     // (as in Cm::prove: small rounds write block partials into mapped host memory, large ones add them on the device)
        u64 *pdst = dev_sum ? part.as<u64>() : c->hpin_dev;
        if (pending) {
            const size_t ldo = nl / 2;
            u64 *Eo = w ? E[1].as<u64>() : E[0].as<u64>(), *Go = w ? G[1].as<u64>() : G[0].as<u64>();
            lfp::launch_r1cs_round_fused(Ec, Gc, ld, half, to_mont(xpend), Eo, Go, ldo, pdst, c->st);
            Ec = Eo; Gc = Go; ld = ldo; w ^= 1;
            len /= 2;
            pending = false;
        } else
            lfp::launch_r1cs_round(Ec, Gc, ld, half, pdst, c->st);
        if (dev_sum) lfp::launch_reduce(part.as<u64>(), nb, 64, c->hpin_dev, 0, c->kappa, 0, 2, 0, nullptr, c->st);
        HIPCHK(c, hipStreamSynchronize(c->st));
        auto tB = std::chrono::steady_clock::now();
        u64 *m = msgs + (size_t)rnd * 4 * D;
        const u32 nbh = dev_sum ? 1u : nb;
        for (int x = 0; x < 4 * D; x++) {
            u64 s = 0;
            for (u32 b = 0; b < nbh; b++) s = fadd(s, hpart[(size_t)b * 64 + x]);
            m[x] = s;
        }
        if (dist) { int rcx = lfp_xsum(c, m, 4 * D); if (rcx) return rcx; }
        tr->absorb_ring(m, 4);
        const u64 x = tr->challenge();
        tr->absorb_const(x);
        ro[rnd] = x;
        if (fuse && rnd + 1 < nvars) { pending = true; xpend = x; }
        else {
            // in-place is not safe (entry b is written while 2b, 2b + 1 of another thread are read): alternate buffers; both hold nl / 2 entries
            const size_t ldo = nl / 2;
            u64 *Eo = w ? E[1].as<u64>() : E[0].as<u64>(), *Go = w ? G[1].as<u64>() : G[0].as<u64>();
            lfp::launch_cm_fix(Ec, ld, Eo, ldo, 1, 1, half, to_mont(x), c->st);
            lfp::launch_cm_fix(Gc, ld, Go, ldo, D, 3, half, to_mont(x), c->st);
            Ec = Eo; Gc = Go; ld = ldo; w ^= 1;
            len = half;
        }
        if (g_tl.on) fprintf(stderr, "[lfplus]   r1cs round %2u: gpu+sync %6.1f us, host + fix launches %6.1f us (nb %u)\n", rnd, std::chrono::duration<double, std::micro>(tB - tA).count(),
                             std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tB).count(), nb);
    }
    LFP_MARK(c, "linearize: tables + rounds");
    // v = f at ro (a sum over the rank's rows); va, vb, vc = the fully fixed tables (replicated)
    HIPCHK(c, hipMemcpy2DAsync(small.as<u64>() + D, D * 8, Gc, ld * D * 8, D * 8, 3, hipMemcpyDeviceToDevice, c->st));    // before E[0] / G are reused
    eq_build_local(c, ro, nvars, E[0].as<u64>());
    lfp::launch_wring(c->f + row0 * D, nl, E[0].as<u64>(), 1, part.as<u64>(), small.as<u64>(), c->st);
    HIPCHK(c, hipMemcpyAsync(evals, small.p, 4 * D * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    { int rcx = lfp_xsum(c, evals, D); if (rcx) return rcx; }
    tr->absorb_ring(evals, 4);
    LFP_MARK(c, "linearize: v = f(ro), absorb");
    return LFPLUS_OK;
}
// ComR1CSProof::verify, host only: stage 1 = a sumcheck round, 2 = e (va vb - vc) != s (the reference asserts, r1cs.rs:159)
extern "C" int lfplus_r1cs_verify(lfplus_transcript *tr, uint32_t nvars, const uint64_t *msgs, const uint64_t *evals, uint64_t *ro, int *stage) {
    if (!tr || !msgs || !evals || !ro || nvars < 1 || nvars > 32) return LFPLUS_E_ARG;
    std::vector<u64> r(nvars);
    for (u32 j = 0; j < nvars; j++) r[j] = tr->challenge();
    int st = 0;
    u64 cur[D] = {0};
    tr->absorb_const(nvars);
    tr->absorb_const(3);
    for (u32 rnd = 0; rnd < nvars && !st; rnd++) {
        const u64 *m = msgs + (size_t)rnd * 4 * D;
        tr->absorb_ring(m, 4);
        const u64 x = tr->challenge();
        tr->absorb_const(x);
        ro[rnd] = x;
        u64 wgt[4];   // Lagrange weights on the nodes 0..3
        for (u32 i = 0; i < 4; i++) {
            u64 num = 1, den = 1;
            for (u32 j = 0; j < 4; j++) if (j != i) { num = fmul(num, fsub(x, (u64)j)); den = fmul(den, fsub((u64)i, (u64)j)); }
            wgt[i] = fmul(num, fpow(den, P - 2));
        }
        for (int ci = 0; ci < D; ci++) {
            if (fadd(m[ci] % P, m[D + ci] % P) != cur[ci]) { st = 1; break; }
            u64 v = 0;
            for (int i = 0; i < 4; i++) v = fadd(v, fmul(m[i * D + ci] % P, wgt[i]));
            cur[ci] = v;
        }
    }
    if (!st) {
        tr->absorb_ring(evals, 4);
        const u64 e = eq_eval(r.data(), ro, nvars);
        u64 t[D] = {0};
        rmul_acc(t, evals + D, evals + 2 * D);
        for (int i = 0; i < D; i++) if (fmul(e, fsub(t[i], evals[3 * D + i] % P)) != cur[i]) st = 2;
    }
    if (stage) *stage = st;
    return st ? LFPLUS_E_REJECT : LFPLUS_OK;
}
// DecompProof::verify (decomp.rs:101-123), host only: C0 + B C1 = cm_f (stage 1) and v0 + B v1 = v over `count` pairs (stage 2)
extern "C" int lfplus_decomp_verify(const uint64_t *C0, const uint64_t *C1, uint32_t kappa, const uint64_t *v0, const uint64_t *v1, uint32_t count, const uint64_t *cm_f,
                                    const uint64_t *v, uint64_t B, int *stage) {
    if (!C0 || !C1 || !v0 || !v1 || !cm_f || !v) return LFPLUS_E_ARG;
    int st = 0;
    for (size_t i = 0; i < (size_t)kappa * D && !st; i++)
        if (fadd(C0[i] % P, fmul(B % P, C1[i] % P)) != cm_f[i] % P) st = 1;
    for (size_t i = 0; i < (size_t)count * 2 * D && !st; i++)
        if (fadd(v0[i] % P, fmul(B % P, v1[i] % P)) != v[i] % P) st = 2;
    if (stage) *stage = st;
    return st ? LFPLUS_E_REJECT : LFPLUS_OK;
}
// Mlin::mlin (mlin.rs:42-107): RgInstance::from_f on every resident witness, Cm::prove, then the folded LinB2: cm_g / vo summed over the instances
// (host, outputs cm_g_sum kappa ring elements, vo_sum (1 + nM) x 2) and g = sum_l g_l on the device -- it becomes ctxs[0]'s resident witness
// (Decomp::decompose reads it there: plus.rs:90-95).  The Cm outputs are lfplus_cm_prove's.
extern "C" int lfplus_mlin(lfplus_ctx *const *ctxs, uint32_t L, lfplus_transcript *tr, uint64_t b, uint32_t k, uint32_t l, uint32_t nM, const uint32_t *const *rowptr,
                           const uint32_t *const *col, const uint64_t *const *val, uint64_t *r_out, uint64_t *msgs, uint64_t *e_out, uint64_t *b_out, uint64_t *v_out,
                           uint64_t *a_out, uint64_t *bb_out, uint64_t *c_out, uint64_t *comh, uint64_t *pa, uint64_t *pb, uint64_t *ea, uint64_t *eb, uint64_t *cm_g,
                           uint64_t *ro, uint64_t *vo, uint64_t *fcoms_out, uint64_t *cm_g_sum, uint64_t *vo_sum) {
    if (!ctxs || !L || !ctxs[0]) return LFPLUS_E_ARG;
    lfplus_ctx *c = ctxs[0];
    if (!cm_g_sum || !vo_sum || !cm_g || !vo) return fail(c, LFPLUS_E_ARG, "lfplus_mlin: bad arguments");
    LFP_MARK(c, "(before mlin)");
    for (u32 i = 0; i < L; i++) {
        if (!ctxs[i]) return fail(c, LFPLUS_E_ARG, "lfplus_mlin: null instance");
        int rc = lfplus_rg_from_f(ctxs[i], b, k, l);
        if (rc) { if (i) c->err = ctxs[i]->err; return rc; }
        LFP_MARK(ctxs[i], "mlin: from_f");
    }
    int rc = lfplus_cm_prove(ctxs, L, tr, l, nM, rowptr, col, val, r_out, msgs, e_out, b_out, v_out, a_out, bb_out, c_out, comh, pa, pb, ea, eb, cm_g, ro, vo, nullptr);
    if (rc) return rc;
    const u32 kappa = c->kappa;
    const size_t n = c->n;
    if (fcoms_out)
        for (u32 i = 0; i < L; i++) {
            const size_t cw = (size_t)kappa * D;
            HIPCHK(c, hipMemcpyAsync(fcoms_out + (size_t)i * 3 * cw, ctxs[i]->comMf + (size_t)k * kappa * D * D, cw * 8, hipMemcpyDeviceToHost, c->st));
            HIPCHK(c, hipMemcpyAsync(fcoms_out + (size_t)i * 3 * cw + cw, ctxs[i]->coms, 2 * cw * 8, hipMemcpyDeviceToHost, c->st));
        }
    memset(cm_g_sum, 0, (size_t)kappa * D * 8);
    memset(vo_sum, 0, (size_t)(1 + nM) * 2 * D * 8);
    for (u32 i = 0; i < L; i++) {
        for (size_t x = 0; x < (size_t)kappa * D; x++) cm_g_sum[x] = fadd(cm_g_sum[x], cm_g[(size_t)i * kappa * D + x]);
        for (size_t x = 0; x < (size_t)(1 + nM) * 2 * D; x++) vo_sum[x] = fadd(vo_sum[x], vo[(size_t)i * (1 + nM) * 2 * D + x]);
        if (i) lfp::launch_vec_add(c->g, ctxs[i]->g, (size_t)c->nloc * D, c->st);
    }
    // the folded witness replaces ctxs[0]'s f: the RgInstance results of ctxs[0] no longer describe the resident witness, and its g buffer holds the
    // sum, not g_0 (lfplus_cm_read_g on ctxs[0] is refused from here on; the other instances keep their g_l)
    c->have = false;
    c->g_valid = false;
    if (!c->sharded()) HIPCHK(c, hipMemcpyAsync(c->f, c->g, n * D * 8, hipMemcpyDeviceToDevice, c->st));
    else {        // a witness is whole on every rank (Decomp::decompose's M_j F_i rows read arbitrary columns): all-gather the rows of g
        int rcg = lfp_allgather_dev(c, c->g, c->f, (size_t)c->nloc * D);
        if (rcg) return rcg;
    }
    HIPCHK(c, hipStreamSynchronize(c->st));
    return LFPLUS_OK;
}

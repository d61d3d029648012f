// lf_common.h -- small helpers shared by the two ring backends of the C ABI (lf_capi.cpp: Goldilocks, bb_capi.cpp: BabyBear)
#pragma once
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <atomic>

#include "../../include/lfhip.h"

#define HIPCHK(x)                                    \
    do {                                             \
        hipError_t e__ = (x);                        \
        if (e__ != hipSuccess) return LF_ERR_HIP;    \
    } while (0)
#define RET(x)                      \
    do {                            \
        int rc__ = (x);             \
        if (rc__ != LF_OK) return rc__; \
    } while (0)

// Device allocation of the main path: when the driver is out of memory, the idle scratch blocks that destroyed LatticeFold+ contexts left in the
// process-wide cache (lfp_ctx.h LfpDevCache) are released and the request retried -- one process may run both provers, and that cache is invisible to hipMalloc.
extern "C" void lfplus_scratch_trim(int device);
template <class T> static inline hipError_t lf_dev_malloc(T **p, size_t bytes) {
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return e;
    lfplus_scratch_trim(dev);
    return hipMalloc((void **)p, bytes);
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t b) {
        if (b <= bytes) return LF_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        // head room so that a slightly larger request of the next call does not reallocate -- capped: the big tables are sized by the shape
        // and would otherwise carry 12.5 % of slack each (1.6 GiB at 2^20 rows)
        const size_t slack = (b >> 3) < ((size_t)8 << 20) ? (b >> 3) : ((size_t)8 << 20);
        size_t want = b + slack + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            if (lf_dev_malloc(&p, b) != hipSuccess) return LF_ERR_HIP;
            want = b;
        }
        bytes = want;
        return LF_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// process-wide serial number of witnesses: a handle that was freed and whose address a later witness reuses is a DIFFERENT witness (caches keyed
// by a witness -- the bit-plane form a step builds -- compare pointer and serial number)
inline uint64_t lf_next_witness_id() {
    static std::atomic<uint64_t> next{1};
    return next.fetch_add(1, std::memory_order_relaxed);
}
struct lf_witness {
    lf_ctx *ctx;      // identity only (compared, never dereferenced by lf_witness_free: a witness may outlive its context)
    int32_t *planes;  // [d][N] centred integer coefficients (d = 24 Goldilocks, 72 BabyBear)
    size_t N;
    int device;          // device the planes live on
    size_t plane_bytes;  // size of the planes allocation (pool key)
    uint64_t id = lf_next_witness_id();
    // Witness::from_f (arith.rs:299-313) also builds f (NTT form) and w_ccs: a fold step materialises both behind compute_f_0; witnesses made by
    // lf_witness_from_* build them on demand (lf_witness_get_f / _get_w_ccs); device buffers from the same pool as the planes, null when not materialised
    uint64_t *f_ntt = nullptr;
    size_t f_bytes = 0;
    uint64_t *w_ccs = nullptr;
    size_t w_bytes = 0;
};

// Device buffers of witness planes are recycled through a small per-context pool: a fold step produces one folded witness and
// its caller frees one, and hipMalloc / hipFree of ~100 MB cost several hundred microseconds (hipFree synchronises the device).
int lf_planes_alloc(lf_ctx *ctx, size_t bytes, int32_t **out);
void lf_planes_release(lf_ctx *ctx, size_t bytes, int32_t *p);
int lf_ctx_device(const lf_ctx *ctx);


// Test/diagnostic switches of the fold-step driver (environment variables): read ONCE at the start of every
// lf_linearize / lf_fold_step call -- never inside the round loops.
struct Tunables {
    bool lin_u_eval = false, fold_unfused = false, fold_no_lut = false, fold_tab_r1 = false, theta_eval = false, fold_no_r4tab = false, fold_no_r5tab = false;
    bool fold_rounds_no_split = false;   // LF_FOLD_ROUNDS_NO_SPLIT: the large table rounds (k_fold_round modes 1, 6, 7) evaluate all five points themselves (four lazy products per table)
    bool fold_sv_no_split = false;    // LF_FOLD_SV_NO_SPLIT: GEMM rounds against the digits of eqB(2p), eqB(2p+1) (three column tiles) instead of the per-pair E_i[p] (two)
    bool lin_no_split = false;        // LF_LIN_NO_SPLIT: linearization rounds on the full eq table (k_lin_round evaluates every point) instead of the split form
    size_t lin_split_min = (size_t)1 << 17;   // LF_LIN_SPLIT_MIN: entries of a round's tables from which the split form is used (at 2^16 rows the host's completion costs what the kernel saves)
    size_t fold_split_min = 8192;     // LF_FOLD_SPLIT_MIN: pairs of a table round (modes 6 / 7) from which it runs in the split form
    bool force_exchange = false;     // LF_DIST_FORCE_EXCHANGE: run the sharded exchanges even with a 1-rank RCCL communicator (test hook)
    bool device_transcript = false;  // LF_DEVICE_TRANSCRIPT: Poseidon sponge of the tail rounds on the device (opt-in: slower than the host's)
    bool shard_plain_rounds = false; // LF_SHARD_PLAIN_ROUNDS: sharded folding rounds on materialised tables only (no fused fix / look-up-table rounds)
    int shard_two_lanes = -1;        // LF_SHARD_TWO_LANES: 1 = threaded two-lane schedule in a sharded step, 0 = one host thread issues every exchange, unset (-1) = two lanes
                                     // when the transport has proved that its two channels work concurrently (lf_dist_init's handshake / two host callbacks)
    size_t shard_lin_min = 16384;    // LF_SHARD_LIN_MIN: a sharded linearization sumcheck hands over to the replicated rounds once its tables have this many entries or fewer
                                     // (a round there is a ~30 us launch: an exchange costs as much as it saves); never above m / 16
    size_t shard_fold_min = 2048;    // LF_SHARD_FOLD_MIN: the same for the folding sumcheck (96 tables per entry: rounds stay worth sharding down to the persistent tail's size)
    long i8_wgs = 0;                 // LF_I8_WGS: workgroups of the int8 commit kernel (0: one per CU)
    bool lin_no_r1cs = false;        // LF_LIN_NO_R1CS: (BabyBear) the R1CS shape through the generic linearization round kernel
    bool lin_no_small = false;       // LF_LIN_NO_SMALL: (BabyBear) small linearization rounds as separate fix / round / reduce launches
    size_t dot_min = 4096;           // LF_DOT_MIN: columns from which the int8 form of the inner products is used
    bool dot_valu = false;           // LF_DOT_VALU: u_s / eta inner products on the 64-bit VALU kernel (k_dot_batch) instead of the int8 matrix cores
    bool fold_no_sv = false;         // LF_FOLD_NO_SV: rounds 1-3 of the folding sumcheck on the VALU kernels instead of the int8 matrix-core GEMMs (lf_sv_rounds.hip)
    size_t sv_min = 65536;           // LF_FOLD_SV_MIN: pairs of a round from which the GEMM form is used
    int sv_rounds = 3;               // LF_FOLD_SV_ROUNDS: last round in GEMM form (1..3)
    bool no_tail = false;            // LF_NO_TAIL: keep one launch set + stream sync per tail round instead of the persistent tail kernel
    size_t fuse_min = 16384, lut_min = (size_t)1 << 17, tab_min = 16384;
    size_t r5_min = 8192;            // LF_FOLD_R5_MIN: pairs of round 5 from which it runs on the planes (mode 7; measured: 2^16 rows slower, 2^20 faster)
    size_t tail_n = 2048;            // LF_TAIL_N: table entries from which the persistent tail kernel takes over
    static Tunables read(size_t lut_min_default) {
        Tunables t;
        t.lut_min = lut_min_default;
        const char *e;
        t.lin_u_eval = getenv("LF_LIN_U_EVAL") != nullptr;
        t.fold_unfused = getenv("LF_FOLD_UNFUSED") != nullptr;
        t.fold_no_lut = getenv("LF_FOLD_NO_LUT") != nullptr;
        t.fold_tab_r1 = getenv("LF_FOLD_TAB_R1") != nullptr;
        t.fold_no_r4tab = getenv("LF_FOLD_NO_R4TAB") != nullptr;
        t.fold_no_r5tab = getenv("LF_FOLD_NO_R5TAB") != nullptr;
        t.lin_no_split = getenv("LF_LIN_NO_SPLIT") != nullptr;
        t.fold_sv_no_split = getenv("LF_FOLD_SV_NO_SPLIT") != nullptr;
        t.fold_rounds_no_split = getenv("LF_FOLD_ROUNDS_NO_SPLIT") != nullptr;
        if (const char *e = getenv("LF_LIN_SPLIT_MIN")) t.lin_split_min = (size_t)atoll(e);
        if (const char *e = getenv("LF_FOLD_SPLIT_MIN")) t.fold_split_min = (size_t)atoll(e);
        t.theta_eval = getenv("LF_THETA_EVAL") != nullptr;
        t.no_tail = getenv("LF_NO_TAIL") != nullptr;
        t.fold_no_sv = getenv("LF_FOLD_NO_SV") != nullptr;
        t.dot_valu = getenv("LF_DOT_VALU") != nullptr;
        if ((e = getenv("LF_DOT_MIN"))) t.dot_min = (size_t)atoll(e);
        t.lin_no_small = getenv("LF_LIN_NO_SMALL") != nullptr;
        t.lin_no_r1cs = getenv("LF_LIN_NO_R1CS") != nullptr;
        if ((e = getenv("LF_FOLD_SV_MIN"))) t.sv_min = (size_t)atoll(e);
        if ((e = getenv("LF_FOLD_SV_ROUNDS"))) t.sv_rounds = atoi(e);
        if ((e = getenv("LF_I8_WGS"))) t.i8_wgs = atol(e);
        if ((e = getenv("LF_SHARD_TWO_LANES"))) t.shard_two_lanes = atoi(e) != 0;
        if ((e = getenv("LF_SHARD_LIN_MIN"))) t.shard_lin_min = (size_t)atoll(e);
        if ((e = getenv("LF_SHARD_FOLD_MIN"))) t.shard_fold_min = (size_t)atoll(e);
        t.shard_plain_rounds = getenv("LF_SHARD_PLAIN_ROUNDS") != nullptr;
        t.device_transcript = getenv("LF_DEVICE_TRANSCRIPT") != nullptr;
        t.force_exchange = getenv("LF_DIST_FORCE_EXCHANGE") != nullptr;
        if ((e = getenv("LF_FOLD_FUSE_MIN"))) t.fuse_min = (size_t)atoll(e);
        if ((e = getenv("LF_FOLD_LUT_MIN"))) t.lut_min = (size_t)atoll(e);
        if ((e = getenv("LF_FOLD_TAB_MIN"))) t.tab_min = (size_t)atoll(e);
        if ((e = getenv("LF_TAIL_N"))) t.tail_n = (size_t)atoll(e);
        if ((e = getenv("LF_FOLD_R5_MIN"))) t.r5_min = (size_t)atoll(e);
        return t;
    }
};

// CSR sanity of the CCS matrices handed to lf_ccs_load: rowptr starts at 0 and is monotone, every column index is < n, every value
// word is a canonical residue.  Run BEFORE any context state is touched.
static inline int lf_validate_csr(unsigned t, size_t m, size_t n, const uint32_t *const *rowptr, const uint32_t *const *col,
                                  const uint64_t *const *val, int words, uint64_t modulus) {
    for (unsigned j = 0; j < t; j++) {
        if (!rowptr[j] || !col[j] || !val[j]) return LF_ERR_INVALID;
        if (rowptr[j][0] != 0) return LF_ERR_INVALID;
        for (size_t r = 0; r < m; r++)
            if (rowptr[j][r + 1] < rowptr[j][r]) return LF_ERR_INVALID;
        const size_t nnz = rowptr[j][m];
        for (size_t k = 0; k < nnz; k++)
            if (col[j][k] >= n) return LF_ERR_INVALID;
        for (size_t k = 0; k < nnz * (size_t)words; k++)
            if (val[j][k] >= modulus) return LF_ERR_INVALID;
    }
    return LF_OK;
}

// External coordinate basis of F_{p^tau} (SURVEY 8c "conventions are data"): the kernels compute in the binomial basis 1, Y, .., Y^(tau-1) of
// F_p[Y]/(Y^tau - nu); the caller's library may present field elements in another F_p-basis (a tower basis, a permuted order ...):
// ext = T * int, per slot.  Everything that crosses the ABI in NTT form and everything the transcript absorbs / squeezes is converted
// with T / T^-1; T = identity (the default) costs nothing.
struct ExtBasis {
    bool on = false;
    int tau = 0;
    uint64_t p = 0;
    uint64_t T[81], Ti[81];
    static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)(((unsigned __int128)a * b) % p); }
    void apply(const uint64_t *M, uint64_t *v) const {   // v (tau words) <- M v
        uint64_t o[9];
        for (int i = 0; i < tau; i++) {
            unsigned __int128 acc = 0;
            for (int j = 0; j < tau; j++) acc += (unsigned __int128)mulmod(M[i * tau + j], v[j], p);
            o[i] = (uint64_t)(acc % p);
        }
        for (int i = 0; i < tau; i++) v[i] = o[i];
    }
    void to_ext(uint64_t *v, size_t slots) const { if (on) for (size_t s = 0; s < slots; s++) apply(T, v + s * tau); }
    void to_int(uint64_t *v, size_t slots) const { if (on) for (size_t s = 0; s < slots; s++) apply(Ti, v + s * tau); }
    // returns LF_ERR_BAD_TABLES when T is singular or does not fix 1 (the base field must sit in coordinate 0 of both bases)
    int set(const uint64_t *Tin, int tau_, uint64_t p_) {
        tau = tau_; p = p_;
        uint64_t M[9][18];
        bool ident = true;
        for (int i = 0; i < tau; i++)
            for (int j = 0; j < tau; j++) {
                uint64_t v = Tin[i * tau + j] % p;
                T[i * tau + j] = v; M[i][j] = v; M[i][tau + j] = i == j;
                if (v != (uint64_t)(i == j)) ident = false;
            }
        for (int i = 0; i < tau; i++)
            if (T[i * tau] != (uint64_t)(i == 0)) return LF_ERR_BAD_TABLES;
        auto inv = [&](uint64_t a) { uint64_t r = 1, e = p - 2; while (e) { if (e & 1) r = mulmod(r, a, p); a = mulmod(a, a, p); e >>= 1; } return r; };
        for (int c = 0; c < tau; c++) {
            int piv = -1;
            for (int r = c; r < tau; r++) if (M[r][c]) { piv = r; break; }
            if (piv < 0) return LF_ERR_BAD_TABLES;
            if (piv != c) for (int k = 0; k < 2 * tau; k++) { uint64_t t = M[piv][k]; M[piv][k] = M[c][k]; M[c][k] = t; }
            uint64_t iv = inv(M[c][c]);
            for (int k = 0; k < 2 * tau; k++) M[c][k] = mulmod(M[c][k], iv, p);
            for (int r = 0; r < tau; r++) {
                if (r == c || !M[r][c]) continue;
                uint64_t f = M[r][c];
                for (int k = 0; k < 2 * tau; k++) { uint64_t t = mulmod(f, M[c][k], p); M[r][k] = M[r][k] >= t ? M[r][k] - t : M[r][k] + (p - t); }   // (no a + p: p is 64 bits wide for Goldilocks)
            }
        }
        for (int i = 0; i < tau; i++)
            for (int j = 0; j < tau; j++) Ti[i * tau + j] = M[i][tau + j];
        on = !ident;
        return LF_OK;
    }
};

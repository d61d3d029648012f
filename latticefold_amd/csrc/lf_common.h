// lf_common.h -- small helpers shared by the two ring backends of the C ABI (lf_capi.cpp: Goldilocks, bb_capi.cpp: BabyBear)
#pragma once
#include <hip/hip_runtime.h>

#include <stdlib.h>

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

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t b) {
        if (b <= bytes) return LF_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = b + (b >> 3) + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            if (hipMalloc(&p, b) != hipSuccess) return LF_ERR_HIP;
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

struct lf_witness {
    lf_ctx *ctx;      // identity only (compared, never dereferenced by lf_witness_free: a witness may outlive its context)
    int32_t *planes;  // [d][N] centred integer coefficients (d = 24 Goldilocks, 72 BabyBear)
    size_t N;
    int device;          // device the planes live on
    size_t plane_bytes;  // size of the planes allocation (pool key)
};

// Device buffers of witness planes are recycled through a small per-context pool: a fold step produces one folded witness and
// its caller frees one, and hipMalloc / hipFree of ~100 MB cost several hundred microseconds (hipFree synchronises the device).
int lf_planes_alloc(lf_ctx *ctx, size_t bytes, int32_t **out);
void lf_planes_release(lf_ctx *ctx, size_t bytes, int32_t *p);
int lf_ctx_device(const lf_ctx *ctx);


// Test/diagnostic switches of the fold-step driver (environment variables): read ONCE at the start of every
// lf_linearize / lf_fold_step call -- never inside the round loops.
struct Tunables {
    bool lin_u_eval = false, fold_unfused = false, fold_no_lut = false, fold_tab_r1 = false, fold_no_mutab = false, theta_eval = false;
    bool force_exchange = false;     // LF_DIST_FORCE_EXCHANGE: run the sharded exchanges even with a 1-rank RCCL communicator (test hook)
    bool device_transcript = false;  // LF_DEVICE_TRANSCRIPT: Poseidon sponge of the tail rounds on the device (opt-in: slower than the host's)
    bool shard_two_lanes = false;    // LF_SHARD_TWO_LANES: threaded two-lane schedule also in a sharded step (default there: one host thread)
    bool no_tail = false;            // LF_NO_TAIL: keep one launch set + stream sync per tail round instead of the persistent tail kernel
    size_t fuse_min = 16384, lut_min = (size_t)1 << 17, tab_min = 16384;
    size_t tail_n = 2048;            // LF_TAIL_N: table entries from which the persistent tail kernel takes over
    long lin_blocks = -1;            // -1: automatic
    static Tunables read(size_t lut_min_default) {
        Tunables t;
        t.lut_min = lut_min_default;
        const char *e;
        t.lin_u_eval = getenv("LF_LIN_U_EVAL") != nullptr;
        t.fold_unfused = getenv("LF_FOLD_UNFUSED") != nullptr;
        t.fold_no_lut = getenv("LF_FOLD_NO_LUT") != nullptr;
        t.fold_tab_r1 = getenv("LF_FOLD_TAB_R1") != nullptr;
        t.fold_no_mutab = getenv("LF_FOLD_NO_MUTAB") != nullptr;
        t.theta_eval = getenv("LF_THETA_EVAL") != nullptr;
        t.no_tail = getenv("LF_NO_TAIL") != nullptr;
        t.shard_two_lanes = getenv("LF_SHARD_TWO_LANES") != nullptr;
        t.device_transcript = getenv("LF_DEVICE_TRANSCRIPT") != nullptr;
        t.force_exchange = getenv("LF_DIST_FORCE_EXCHANGE") != nullptr;
        if ((e = getenv("LF_FOLD_FUSE_MIN"))) t.fuse_min = (size_t)atoll(e);
        if ((e = getenv("LF_FOLD_LUT_MIN"))) t.lut_min = (size_t)atoll(e);
        if ((e = getenv("LF_FOLD_TAB_MIN"))) t.tab_min = (size_t)atoll(e);
        if ((e = getenv("LF_LIN_BLOCKS"))) t.lin_blocks = atol(e);
        if ((e = getenv("LF_TAIL_N"))) t.tail_n = (size_t)atoll(e);
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

// lf_common.h -- small helpers shared by the two ring backends of the C ABI (lf_capi.cpp: Goldilocks, bb_capi.cpp: BabyBear)
#pragma once
#include <hip/hip_runtime.h>

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
    lf_ctx *ctx;
    int32_t *planes;  // [d][N] centred integer coefficients (d = 24 Goldilocks, 72 BabyBear)
    size_t N;
};

// Device buffers of witness planes are recycled through a small per-context pool: a fold step produces one folded witness and
// its caller frees one, and hipMalloc / hipFree of ~100 MB cost several hundred microseconds (hipFree synchronises the device).
int lf_planes_alloc(lf_ctx *ctx, size_t bytes, int32_t **out);
void lf_planes_release(lf_ctx *ctx, size_t bytes, int32_t *p);


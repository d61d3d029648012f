// lfp_field.cuh -- F_p of the Frog ring (p = 15912092521325583641) on the device: Montgomery products with R = 2^64
#pragma once
#include "lfp_kernels.h"
namespace lfp {
// ---- F_p: Montgomery products (R = 2^64) for the final reductions and for tensor / tensor_product
constexpr u64 mont_pinv() {   // -p^{-1} mod 2^64
    u64 x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - P * x;
    return ~x + 1;
}
constexpr u64 mont_r2() {     // 2^128 mod p
    unsigned __int128 r = 1;
    for (int i = 0; i < 128; i++) { r <<= 1; if (r >= P) r -= P; }
    return (u64)r;
}
constexpr u64 PINV = mont_pinv(), R2 = mont_r2();   // forced compile-time evaluation: called in a device function the loops would run per thread
static_assert((u64)(P * (0 - PINV)) == 1, "mont_pinv");
__device__ __forceinline__ u64 mont_mul(u64 a, u64 b) {   // a b 2^-64 mod p, a, b < p
    u64 lo = a * b, hi = __umul64hi(a, b);
    u64 m = lo * PINV;
    u64 mh = __umul64hi(m, P), ml = m * P;
    u64 cy = (lo + ml) < lo;      // the low word cancels to 0 (mod 2^64); only its carry matters
    u64 u = hi + mh, o1 = u < hi;
    u64 v = u + cy, o2 = v < cy;
    if (o1 || o2 || v >= P) v -= P;
    return v;
}
__device__ __forceinline__ u64 mul_p(u64 a, u64 b) { return mont_mul(mont_mul(a, b), R2); }
__device__ __forceinline__ u64 add_p(u64 a, u64 b) {
    u64 s = a + b;
    if (s < a || s >= P) s -= P;
    return s;
}
__device__ __forceinline__ u64 sub_p(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
__device__ __forceinline__ u64 to_mont(u64 a) { return mont_mul(a, R2); }        // a 2^64 mod p
__device__ __forceinline__ u64 from_mont(u64 a) { return mont_mul(a, 1); }
}  // namespace lfp

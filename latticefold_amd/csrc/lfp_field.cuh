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
// ---- lazy sums of 64 x 64-bit products: a 160-bit integer in five 32-bit registers, ONE Montgomery reduction for the whole sum (a mont_mul is three 64-bit
// multiplications -- the product, lo * PINV, m * P -- and its corrections; a lazy term is one multiplication and a five-word add with carry)
struct Acc160 { u32 a[5]; };
__device__ __forceinline__ void acc160_zero(Acc160 &n) { n.a[0] = n.a[1] = n.a[2] = n.a[3] = n.a[4] = 0; }
// signed use (negacyclic products: the wrapped terms are subtracted): start from 16 p 2^64, a multiple of p 2^64 that 16 subtracted products (< 16 p^2) cannot
// exhaust -- the integer stays >= 0 and the bias vanishes in the reduction
__device__ __forceinline__ void acc160_bias16(Acc160 &n) { n.a[0] = n.a[1] = 0; n.a[2] = (u32)(P << 4); n.a[3] = (u32)((P << 4) >> 32); n.a[4] = (u32)(P >> 60); }
__device__ __forceinline__ void acc160_mad(Acc160 &n, u64 x, u64 y) {                 // += x y   (up to 2^32 terms)
    const u64 lo = x * y, hi = __umul64hi(x, y);
    const u32 p0 = (u32)lo, p1 = (u32)(lo >> 32), p2 = (u32)hi, p3 = (u32)(hi >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %5\n\tv_addc_co_u32 %1, vcc, %1, %6, vcc\n\tv_addc_co_u32 %2, vcc, %2, %7, vcc\n\tv_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
        "v_addc_co_u32 %4, vcc, 0, %4, vcc"
        : "+v"(n.a[0]), "+v"(n.a[1]), "+v"(n.a[2]), "+v"(n.a[3]), "+v"(n.a[4])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "vcc");
}
__device__ __forceinline__ void acc160_mad_signed(Acc160 &n, u64 x, u64 y, bool neg) { // += x y or -= x y: two's complement over the five words (xor mask, carry-in, sign word)
    const u64 lo = x * y, hi = __umul64hi(x, y);
    const u32 m = neg ? 0xffffffffu : 0u;
    const u32 p0 = (u32)lo ^ m, p1 = (u32)(lo >> 32) ^ m, p2 = (u32)hi ^ m, p3 = (u32)(hi >> 32) ^ m;
    asm("v_cmp_ne_u32 vcc, 0, %9\n\tv_addc_co_u32 %0, vcc, %0, %5, vcc\n\tv_addc_co_u32 %1, vcc, %1, %6, vcc\n\tv_addc_co_u32 %2, vcc, %2, %7, vcc\n\t"
        "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\tv_addc_co_u32 %4, vcc, %4, %9, vcc"
        : "+v"(n.a[0]), "+v"(n.a[1]), "+v"(n.a[2]), "+v"(n.a[3]), "+v"(n.a[4])
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(m)
        : "vcc");
}
// (w0 + w1 2^64 + w2 2^128) 2^-64 mod p = w0 2^-64 + w1 + w2 2^64: the value the same terms give through mont_mul and add_p, term by term
__device__ __forceinline__ u64 acc160_red(const Acc160 &n) {
    u64 w0 = ((u64)n.a[1] << 32) | n.a[0], w1 = ((u64)n.a[3] << 32) | n.a[2];
    if (w0 >= P) w0 -= P;
    if (w1 >= P) w1 -= P;
    return add_p(add_p(mont_mul(w0, 1), w1), mont_mul((u64)n.a[4], R2));
}
}  // namespace lfp

// lfp_kernels.h -- LatticeFold+ slice on the Frog ring Z_p[X]/(X^16 + 1), p = 15912092521325583641, COEFFICIENT form
// (the ring the reference runs latticefold-plus on: cyclotomic-rings/src/rings/frog.rs, crates/latticefold-plus/src/rgchk.rs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace lfp {
typedef uint64_t u64;
typedef uint32_t u32;
constexpr u64 P = 15912092521325583641ull;
constexpr int D = 16;
constexpr int TJ = 32;        // witness rows per LDS tile
constexpr int IG = 4;         // commitment-matrix rows per launch group
constexpr int KG = 4;         // digit planes per launch group
constexpr int RED_WAVES = 8;  // waves per block of the partial-sum reduction

struct Phase1Args {
    const u64 *f;       // n x 16 canonical words
    const u64 *A;       // kappa x n x 16
    u64 n;
    u32 kappa, i0, icnt;
    u32 k, k0, kcnt;
    u64 b;              // digit base
    int sh;             // log2(b) when b is a power of two, else -1
    u32 J;              // rows per block (multiple of TJ)
    int8_t *Df;         // [k][n][16], written when write_df
    u64 *pm_lo, *pm_hi; // [nblk][k*kappa*256] partial sums of the monomial products (128-bit)
    u64 *pf0, *pf1, *pf2; // [nblk][kappa*16] partial sums of A f (192-bit)
    u32 *err;
    int write_df, do_f;
};
struct Phase2Args {
    const u64 *A;
    const u64 *tau;     // n canonical words
    u64 n;
    u32 kappa, i0, icnt;
    u32 J;
    int8_t *mtau;       // n exponents (centred tau), written when i0 == 0
    u64 *pc0, *pc1, *pc2;  // [nblk][kappa*16]  A * tau       (192-bit)
    u64 *pt_lo, *pt_hi;    // [nblk][kappa*16]  A * exp(tau)  (128-bit)
    u32 *err;
};
void launch_phase1(const Phase1Args &a, u32 nblk, hipStream_t s);
void launch_phase2(const Phase2Args &a, u32 nblk, hipStream_t s);
// out[o] = (sum over blocks of the 128/192-bit partials) mod p;  w2 may be null
void launch_reduce(const u64 *w0, const u64 *w1, const u64 *w2, u32 nblk, u32 nout, u64 *out, hipStream_t s);
// tau = split(hconcat(comM_f), n, base, l) (utils.rs:12-43): comMf [k][kappa][16][16] -> tau positions [0, kappa*k*16*l*16)
void launch_split(const u64 *comMf, u32 kappa, u32 k, u64 base, u32 l, u64 *tau, hipStream_t s);
void launch_tensor_level(const u64 *cur, u64 len, u64 r, u64 *nxt, hipStream_t s);
void launch_tensor_product(const u64 *a, u64 m, const u64 *b, u64 n, u64 *out, hipStream_t s);
}  // namespace lfp

// lfp_kernels.hip -- gfx950 kernels of the LatticeFold+ double commitment RgInstance::from_f (crates/latticefold-plus/src/rgchk.rs:260-331)
// on the Frog ring in coefficient form.  Integer / HBM-bound work; no MFMA.
//
//   phase 1 (k_rg_phase1): one pass over f and A.  Per tile of TJ witness rows a block
//       * cuts the centred coefficients of f into k balanced base-b digits (D_f, written out as int8),
//       * keeps exp(D_f) only as EXPONENTS: multiplying A[i][j] by the unit monomial X^e is a negacyclic rotation, so
//         comM_f[ki][i][c] = sum_j A[i][j] * exp(D_f[ki][j][c]) is 16 x 16 shift-adds per (ki, i, j) -- a 32-entry LDS table
//         [a_0 .. a_15, p - a_0 .. p - a_15] per (i, j) turns rotation + sign into ONE masked index,
//       * accumulates cm_f = A f as lazy 64 x 64 products (AccP of lf_field.cuh: no modular reduction inside the loop).
//     Sums stay lazy (96-bit / 192-bit integers); a block writes its partial sums, k_reduce adds the blocks and reduces mod p once.
//   split (k_split): base-(d/2) gadget digits of comM_f -> tau.
//   phase 2 (k_rg_phase2): second pass over A: C_Mf = A tau (scalar multiples) and cm_mtau = A exp(tau) (rotations through a
//     16-lane shuffle).
#include "lfp_kernels.h"
#include "lf_field.cuh"

namespace lfp {
using lf::AccP;
using lf::accp_mad;

__device__ __forceinline__ void add96(u64 &lo, u32 &hi, u64 v) {
    u64 s = lo + v;
    hi += (u32)(s < v);
    lo = s;
}
struct U192 {
    u64 w0, w1, w2;
};
__device__ __forceinline__ void u192_add(U192 &a, const U192 &b) {
    u64 s0 = a.w0 + b.w0, c0 = s0 < b.w0;
    u64 s1 = a.w1 + b.w1, c1 = s1 < b.w1;
    u64 s1b = s1 + c0;
    c1 += s1b < c0;
    a.w0 = s0;
    a.w1 = s1b;
    a.w2 = a.w2 + b.w2 + c1;
}
// s00 + 2^32 s01 + 2^64 (s11 + c00) + 2^96 c01 + 2^128 c11
__device__ __forceinline__ U192 accp_to_u192(const AccP &s) {
    U192 r = {s.s00, 0, 0}, t;
    t = {s.s01 << 32, (s.s01 >> 32) + ((u64)s.c01 << 32), 0};
    u192_add(r, t);
    t = {0, s.s11, 0};
    u192_add(r, t);
    t = {0, (u64)s.c00, (u64)s.c11};
    u192_add(r, t);
    return r;
}
__device__ __forceinline__ void accp_zero(AccP &s) {
    s.s00 = s.s01 = s.s11 = 0;
    s.c00 = s.c01 = s.c11 = 0;
}
// balanced digit step (stark_rings::balanced_decomposition as restated in oracle/lfp.c: truncating remainder, |rem| <= b/2 kept)
template <bool POW2>
__device__ __forceinline__ int64_t digit_step(int64_t &cur, u64 b, int sh) {
    int64_t q, rem;
    if (POW2) {
        q = (cur + ((cur >> 63) & (int64_t)(b - 1))) >> sh;
        rem = cur - (q << sh);
    } else {
        q = cur / (int64_t)b;
        rem = cur - q * (int64_t)b;
    }
    int64_t half = (int64_t)(b >> 1), ar = rem < 0 ? -rem : rem;
    if (ar > half) {
        if (rem < 0) { rem += (int64_t)b; q -= 1; }
        else { rem -= (int64_t)b; q += 1; }
    }
    cur = q;
    return rem;
}
__device__ __forceinline__ int64_t centre(u64 v) { return v <= (P - 1) / 2 ? (int64_t)v : -(int64_t)(P - v); }

template <bool POW2>
__global__ __launch_bounds__(256) void k_rg_phase1(Phase1Args a) {
    __shared__ u64 ftab[TJ][32];
    __shared__ u64 atab[IG][TJ][32];
    __shared__ unsigned long long ex[KG][D][TJ / 8];   // byte jj of (ki, c): 8 * exponent
    const int tid = threadIdx.x, t = tid & 15, hi4 = tid >> 4;
    const u32 blk = blockIdx.x;
    const u64 jb = (u64)blk * a.J;
    const u32 nout_m = a.k * a.kappa * 256, nout_f = a.kappa * 16;

    u64 mlo[KG][IG];
    u32 mhi[KG][IG];
    AccP fa[IG];
#pragma unroll
    for (int q = 0; q < KG; q++)
#pragma unroll
        for (int i = 0; i < IG; i++) { mlo[q][i] = 0; mhi[q][i] = 0; }
#pragma unroll
    for (int i = 0; i < IG; i++) accp_zero(fa[i]);

    for (u32 tj = 0; tj < a.J; tj += TJ) {
        const u64 j0 = jb + tj;
        if (j0 >= a.n) break;
        __syncthreads();
        // ---- f tile: digits, exponent bytes, +-f table
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int idx = tid + 256 * h, jj = idx >> 4, c = idx & 15;
            u64 j = j0 + jj;
            u64 v = j < a.n ? a.f[j * D + c] : 0;
            ftab[jj][c] = v;
            ftab[jj][16 + c] = P - v;
            if (a.kcnt) {
                int64_t cur = centre(v);
                for (u32 ki = 0; ki < a.k0 + a.kcnt; ki++) {
                    int64_t dg = digit_step<POW2>(cur, a.b, a.sh);
                    if (dg <= -(D / 2) || dg >= D / 2) { atomicOr(a.err, 1u); dg = 0; }
                    if (a.write_df && j < a.n && ki >= a.k0) a.Df[((u64)ki * a.n + j) * D + c] = (int8_t)dg;
                    if (ki >= a.k0) ((unsigned char *)&ex[ki - a.k0][c][0])[jj] = (unsigned char)(((int)dg & 15) << 3);
                }
            }
        }
        // ---- A tiles of the row group
        for (u32 ii = 0; ii < a.icnt; ii++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                int idx = tid + 256 * h, jj = idx >> 4, c = idx & 15;
                u64 j = j0 + jj;
                u64 v = j < a.n ? a.A[((u64)(a.i0 + ii) * a.n + j) * D + c] : 0;
                atab[ii][jj][c] = v;
                atab[ii][jj][16 + c] = P - v;
            }
        }
        __syncthreads();
        // ---- cm_f: thread (jl = hi4, t) owns output coefficient t of rows jl, jl + 16 of the tile
        if (a.do_f) {
#pragma unroll
            for (int h = 0; h < TJ / 16; h++) {
                const int jj = hi4 + 16 * h;
#pragma unroll
                for (int s = 0; s < D; s++) {
                    u64 fv = ftab[jj][(t - s) & 31];
#pragma unroll
                    for (int i = 0; i < IG; i++)
                        if (i < (int)a.icnt) accp_mad(fa[i], atab[i][jj][s], fv);
                }
            }
        }
        // ---- comM_f: thread (c = hi4, t) owns coefficient t of column c
        const u32 t8 = (u32)t << 3;
#pragma unroll
        for (int q = 0; q < KG; q++) {
            if (q < (int)a.kcnt) {
#pragma unroll
                for (int w = 0; w < TJ / 8; w++) {
                    unsigned long long eb = ex[q][hi4][w];
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        u32 e8 = (u32)(eb >> (8 * m)) & 0xFFu;
                        u32 u8 = (t8 - e8) & 0xF8u;
                        const int jj = w * 8 + m;
#pragma unroll
                        for (int i = 0; i < IG; i++)
                            if (i < (int)a.icnt) add96(mlo[q][i], mhi[q][i], *(const u64 *)((const char *)&atab[i][jj][0] + u8));
                    }
                }
            }
        }
    }
    // ---- partial sums of the block
#pragma unroll
    for (int q = 0; q < KG; q++)
#pragma unroll
        for (int i = 0; i < IG; i++)
            if (q < (int)a.kcnt && i < (int)a.icnt) {
                u64 o = (u64)blk * nout_m + (((u64)(a.k0 + q) * a.kappa + a.i0 + i) * D + hi4) * D + t;
                a.pm_lo[o] = mlo[q][i];
                a.pm_hi[o] = mhi[q][i];
            }
    if (a.do_f) {
        U192 *red = (U192 *)&atab[0][0][0];   // 256 * 24 B
#pragma unroll
        for (int i = 0; i < IG; i++) {
            if (i < (int)a.icnt) {
                __syncthreads();
                red[tid] = accp_to_u192(fa[i]);
                __syncthreads();
                if (tid < 16) {
                    U192 s = red[tid];
                    for (int l = 1; l < 16; l++) u192_add(s, red[l * 16 + tid]);
                    u64 o = (u64)blk * nout_f + (a.i0 + i) * D + tid;
                    a.pf0[o] = s.w0;
                    a.pf1[o] = s.w1;
                    a.pf2[o] = s.w2;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_rg_phase2(Phase2Args a) {
    __shared__ U192 red[256];
    __shared__ u64 redm[256][2];
    const int tid = threadIdx.x, t = tid & 15, jl = tid >> 4;
    const u32 blk = blockIdx.x;
    const u64 jb = (u64)blk * a.J, je = jb + a.J < a.n ? jb + a.J : a.n;
    const u32 nout = a.kappa * 16;
    AccP ca[IG];
    u64 tlo[IG];
    u32 thi[IG];
#pragma unroll
    for (int i = 0; i < IG; i++) { accp_zero(ca[i]); tlo[i] = 0; thi[i] = 0; }
    for (u64 j = jb + jl; j < je; j += 16) {
        u64 tv = a.tau[j];
        int64_t tc = centre(tv);
        if (tc <= -(D / 2) || tc >= D / 2) { atomicOr(a.err, 2u); tc = 0; }
        if (a.i0 == 0 && t == 0) a.mtau[j] = (int8_t)tc;
        const int src = t - ((int)tc & 15);
#pragma unroll
        for (int i = 0; i < IG; i++) {
            if (i < (int)a.icnt) {
                u64 av = a.A[((u64)(a.i0 + i) * a.n + j) * D + t];
                accp_mad(ca[i], av, tv);
                u64 r = __shfl(av, src & 15, 16);
                add96(tlo[i], thi[i], src < 0 ? P - r : r);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < IG; i++) {
        if (i < (int)a.icnt) {
            __syncthreads();
            red[tid] = accp_to_u192(ca[i]);
            redm[tid][0] = tlo[i];
            redm[tid][1] = thi[i];
            __syncthreads();
            if (tid < 16) {
                U192 s = red[tid], m = {redm[tid][0], redm[tid][1], 0};
                for (int l = 1; l < 16; l++) {
                    u192_add(s, red[l * 16 + tid]);
                    U192 x = {redm[l * 16 + tid][0], redm[l * 16 + tid][1], 0};
                    u192_add(m, x);
                }
                u64 o = (u64)blk * nout + (a.i0 + i) * D + tid;
                a.pc0[o] = s.w0; a.pc1[o] = s.w1; a.pc2[o] = s.w2;
                a.pt_lo[o] = m.w0; a.pt_hi[o] = m.w1;
            }
        }
    }
}

// (w3 w2 w1 w0) mod p by shift-subtract (a few hundred outputs per call: not worth a Barrett constant)
__device__ u64 mod_p_256(u64 w0, u64 w1, u64 w2, u64 w3) {
    u64 w[4] = {w0, w1, w2, w3};
    u64 r = 0;
    for (int k = 3; k >= 0; k--) {
        if (k && !w[k] && !r) continue;
        for (int bit = 63; bit >= 0; bit--) {
            u64 top = r >> 63;
            r = (r << 1) | ((w[k] >> bit) & 1);
            if (top || r >= P) r -= P;
        }
    }
    return r;
}
__global__ __launch_bounds__(64 * RED_WAVES) void k_reduce(const u64 *w0, const u64 *w1, const u64 *w2, u32 nblk, u32 nout, u64 *out) {
    __shared__ u64 part[RED_WAVES][64][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 o = blockIdx.x * 64 + lane;
    u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (o < nout)
        for (u32 b = wv; b < nblk; b += RED_WAVES) {
            u64 i = (u64)b * nout + o;
            u64 x0 = w0[i], x1 = w1[i], x2 = w2 ? w2[i] : 0;
            u64 a0 = s0 + x0, c0 = a0 < x0;
            u64 a1 = s1 + x1, c1 = a1 < x1;
            u64 a1b = a1 + c0; c1 += a1b < c0;
            u64 a2 = s2 + x2, c2 = a2 < x2;
            u64 a2b = a2 + c1; c2 += a2b < c1;
            s0 = a0; s1 = a1b; s2 = a2b; s3 += c2;
        }
    part[wv][lane][0] = s0; part[wv][lane][1] = s1; part[wv][lane][2] = s2; part[wv][lane][3] = s3;
    __syncthreads();
    if (wv == 0 && o < nout) {
        for (int v = 1; v < RED_WAVES; v++) {
            u64 x0 = part[v][lane][0], x1 = part[v][lane][1], x2 = part[v][lane][2], x3 = part[v][lane][3];
            u64 a0 = s0 + x0, c0 = a0 < x0;
            u64 a1 = s1 + x1, c1 = a1 < x1;
            u64 a1b = a1 + c0; c1 += a1b < c0;
            u64 a2 = s2 + x2, c2 = a2 < x2;
            u64 a2b = a2 + c1; c2 += a2b < c1;
            s0 = a0; s1 = a1b; s2 = a2b; s3 += x3 + c2;
        }
        out[o] = mod_p_256(s0, s1, s2, s3);
    }
}

__global__ void k_split(const u64 *comMf, u32 kappa, u32 k, u64 base, int sh, u32 l, u64 *tau) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= kappa * k * 256) return;
    u32 t = g & 15, c = (g >> 4) & 15, ik = g >> 8, ki = ik % k, i = ik / k;
    int64_t cur = centre(comMf[(((u64)ki * kappa + i) * D + c) * D + t]);
    u64 pos = (((u64)i * k + ki) * D + c) * (u64)l * D + t;
    for (u32 j = 0; j < l; j++) {
        int64_t dg = sh >= 0 ? digit_step<true>(cur, base, sh) : digit_step<false>(cur, base, sh);
        tau[pos + (u64)j * D] = dg >= 0 ? (u64)dg : P - (u64)(-dg);
    }
}

// ---- F_p products for tensor / tensor_product (utils.rs:45-83): Montgomery, R = 2^64
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
__device__ __forceinline__ u64 mont_mul(u64 a, u64 b) {
    u64 lo = a * b, hi = __umul64hi(a, b);
    u64 m = lo * mont_pinv();
    u64 mh = __umul64hi(m, P), ml = m * P;
    u64 cy = (lo + ml) < lo;      // the low word cancels to 0 (mod 2^64); only its carry matters
    u64 u = hi + mh, o1 = u < hi;
    u64 v = u + cy, o2 = v < cy;
    if (o1 || o2 || v >= P) v -= P;
    return v;
}
__device__ __forceinline__ u64 mul_p(u64 a, u64 b) { return mont_mul(mont_mul(a, b), mont_r2()); }
__global__ void k_tensor_level(const u64 *cur, u64 len, u64 r, u64 *nxt) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= len) return;
    u64 v = cur[x], one_minus = r <= 1 ? 1 - r : P + 1 - r;
    nxt[2 * x] = mul_p(v, one_minus);
    nxt[2 * x + 1] = mul_p(v, r);
}
__global__ void k_tensor_product(const u64 *a, u64 m, const u64 *b, u64 n, u64 *out) {
    u64 x = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= m * n) return;
    out[x] = mul_p(a[x / n], b[x % n]);
}

static int log2_exact(u64 b) { return (b && !(b & (b - 1))) ? __builtin_ctzll(b) : -1; }
void launch_phase1(const Phase1Args &a, u32 nblk, hipStream_t s) {
    if (a.sh >= 0) hipLaunchKernelGGL(k_rg_phase1<true>, dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_rg_phase1<false>, dim3(nblk), dim3(256), 0, s, a);
}
void launch_phase2(const Phase2Args &a, u32 nblk, hipStream_t s) { hipLaunchKernelGGL(k_rg_phase2, dim3(nblk), dim3(256), 0, s, a); }
void launch_reduce(const u64 *w0, const u64 *w1, const u64 *w2, u32 nblk, u32 nout, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce, dim3((nout + 63) / 64), dim3(64 * RED_WAVES), 0, s, w0, w1, w2, nblk, nout, out);
}
void launch_split(const u64 *comMf, u32 kappa, u32 k, u64 base, u32 l, u64 *tau, hipStream_t s) {
    u32 th = kappa * k * 256;
    hipLaunchKernelGGL(k_split, dim3((th + 255) / 256), dim3(256), 0, s, comMf, kappa, k, base, log2_exact(base), l, tau);
}
void launch_tensor_level(const u64 *cur, u64 len, u64 r, u64 *nxt, hipStream_t s) {
    hipLaunchKernelGGL(k_tensor_level, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, cur, len, r, nxt);
}
void launch_tensor_product(const u64 *a, u64 m, const u64 *b, u64 n, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_tensor_product, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, s, a, m, b, n, out);
}
}  // namespace lfp

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
#include "lfp_field.cuh"

namespace lfp {
using lf::AccP;
using lf::accp_mad;

// acc (96 bits in three VGPRs) += v
__device__ __forceinline__ void add96(u32 &a0, u32 &a1, u32 &a2, u64 v) {
    u32 v0 = (u32)v, v1 = (u32)(v >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc"
        : "+v"(a0), "+v"(a1), "+v"(a2)
        : "v"(v0), "v"(v1)
        : "vcc");
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

// (r 2^64 + w) mod p for r < p: mont_mul(r, 2^128) = r 2^64
__device__ __forceinline__ u64 red_word(u64 r, u64 w) { return add_p(mont_mul(r, R2), w >= P ? w - P : w); }
__device__ __forceinline__ u64 mod_p_192(U192 x) { return red_word(red_word(x.w2 % P, x.w1), x.w0); }

// balanced digit step (stark_rings::balanced_decomposition as restated in oracle/lfp.c: truncating remainder, |rem| <= b/2 kept)
__device__ __forceinline__ int64_t digit_step(int64_t &cur, u64 b, int sh) {
    int64_t q, rem;
    if (sh >= 0) {
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

// sum over the 16 row lanes (tid >> 4) of a value mod p held by thread (jl, t): two 16-lane shuffles inside each wave, LDS across waves;
// valid in threads tid < 16
__device__ __forceinline__ u64 sum_over_row_lanes(u64 v, u64 (*sh)[16], int tid) {
    v = add_p(v, __shfl_xor(v, 16));
    v = add_p(v, __shfl_xor(v, 32));
    __syncthreads();
    if ((tid & 63) < 16) sh[tid >> 6][tid & 15] = v;
    __syncthreads();
    u64 r = 0;
    if (tid < 16) r = add_p(add_p(sh[0][tid], sh[1][tid]), add_p(sh[2][tid], sh[3][tid]));
    return r;
}
// ICNT rows of A and KCNT digit planes per launch, both compile-time: the inner loops carry no predicates, so the LDS reads of a
// whole 8-row group are in flight together.  The next tile's global loads are issued before the current tile is consumed.
template <int ICNT, int KCNT>
__global__ __launch_bounds__(256) void k_rg_phase1(Phase1Args a) {
    constexpr int KX = KCNT ? KCNT : 1;
    __shared__ u64 ftab[TJ][32];
    __shared__ u64 atab[ICNT][TJ][32];
    __shared__ unsigned long long ex[KX][D][TJ / 8];   // byte jj of (ki, c): 8 * exponent
    const int tid = threadIdx.x, t = tid & 15, hi4 = tid >> 4;
    const u32 blk = blockIdx.x;
    const u64 jb = (u64)blk * a.J;
    const u32 nout = a.k * a.kappa * 256 + a.kappa * 16;

    u32 m0[KX][ICNT], m1[KX][ICNT], m2[KX][ICNT];
    AccP fa[ICNT];
#pragma unroll
    for (int q = 0; q < KX; q++)
#pragma unroll
        for (int i = 0; i < ICNT; i++) m0[q][i] = m1[q][i] = m2[q][i] = 0;
#pragma unroll
    for (int i = 0; i < ICNT; i++) accp_zero(fa[i]);

    u64 nf[2], na[ICNT][2];
    auto fetch = [&](u64 j0) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int idx = tid + 256 * h;
            u64 j = j0 + (idx >> 4);
            bool in = j < a.n;
            nf[h] = in ? a.f[j * D + (idx & 15)] : 0;
#pragma unroll
            for (int i = 0; i < ICNT; i++) na[i][h] = in ? a.A[((u64)(a.i0 + i) * a.n + j) * D + (idx & 15)] : 0;
        }
    };
    fetch(jb);
    for (u32 tj = 0; tj < a.J; tj += TJ) {
        const u64 j0 = jb + tj;
        if (j0 >= a.n) break;
        __syncthreads();
        // ---- tile -> LDS: +-f table, digits (D_f out, exponent bytes), +-A tables
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int idx = tid + 256 * h, jj = idx >> 4, c = idx & 15;
            u64 j = j0 + jj, v = nf[h];
            ftab[jj][c] = v;
            ftab[jj][16 + c] = P - v;
#pragma unroll
            for (int i = 0; i < ICNT; i++) {
                atab[i][jj][c] = na[i][h];
                atab[i][jj][16 + c] = P - na[i][h];
            }
            if (KCNT) {
                int64_t cur = centre(v);
                for (u32 ki = 0; ki < a.k0 + KCNT; ki++) {
                    int64_t dg = digit_step(cur, a.b, a.sh);
                    if (dg <= -(D / 2) || dg >= D / 2) { atomicOr(a.err, 1u); dg = 0; }
                    if (ki >= a.k0) {
                        if (a.write_df && j < a.n) a.Df[((u64)ki * a.n + j) * D + c] = (int8_t)dg;
                        ((unsigned char *)&ex[ki - a.k0][c][0])[jj] = (unsigned char)(((int)dg & 15) << 3);
                    }
                }
            }
        }
        if (tj + TJ < a.J && j0 + TJ < a.n) fetch(j0 + TJ);
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
                    for (int i = 0; i < ICNT; i++) accp_mad(fa[i], atab[i][jj][s], fv);
                }
            }
        }
        // ---- comM_f: thread (c = hi4, t) owns coefficient t of column c
        if (KCNT) {
            const u32 t8 = (u32)t << 3;
#pragma unroll
            for (int q = 0; q < KCNT; q++) {
#pragma unroll
                for (int w = 0; w < TJ / 8; w++) {
                    unsigned long long eb = ex[q][hi4][w];
                    u64 vals[8][ICNT];
#pragma unroll
                    for (int m = 0; m < 8; m++) {
                        u32 u8 = (t8 - ((u32)(eb >> (8 * m)) & 0xFFu)) & 0xF8u;
#pragma unroll
                        for (int i = 0; i < ICNT; i++) vals[m][i] = *(const u64 *)((const char *)&atab[i][w * 8 + m][0] + u8);
                    }
#pragma unroll
                    for (int m = 0; m < 8; m++)
#pragma unroll
                        for (int i = 0; i < ICNT; i++) add96(m0[q][i], m1[q][i], m2[q][i], vals[m][i]);
                }
            }
        }
    }
    // ---- partial sums of the block, reduced mod p: [blk][ comM_f (k, kappa, 16, 16) | cm_f (kappa, 16) ]
    u64 *part = a.part + (u64)blk * nout;
    if (KCNT) {
#pragma unroll
        for (int q = 0; q < KCNT; q++)
#pragma unroll
            for (int i = 0; i < ICNT; i++)
                part[(((u64)(a.k0 + q) * a.kappa + a.i0 + i) * D + hi4) * D + t] = red_word(m2[q][i], ((u64)m1[q][i] << 32) | m0[q][i]);
    }
    if (a.do_f) {
        u64 (*sh)[16] = (u64 (*)[16]) & ftab[0][0];
#pragma unroll
        for (int i = 0; i < ICNT; i++) {
            u64 r = sum_over_row_lanes(mod_p_192(accp_to_u192(fa[i])), sh, tid);
            if (tid < 16) part[(u64)a.k * a.kappa * 256 + (a.i0 + i) * D + tid] = r;
        }
    }
}

// second pass over A: C_Mf = A tau (scalar multiples), cm_mtau = A exp(tau) (rotation = 16-lane shuffle); thread (jl, t), rows jl + 16 m
template <int ICNT>
__global__ __launch_bounds__(256) void k_rg_phase2(Phase2Args a) {
    __shared__ u64 sh[4][16];
    const int tid = threadIdx.x, t = tid & 15, jl = tid >> 4;
    const u32 blk = blockIdx.x;
    const u64 jb = (u64)blk * a.J;
    const u32 nout = 2 * a.kappa * 16;
    AccP ca[ICNT];
    u32 t0[ICNT], t1[ICNT], t2[ICNT];
#pragma unroll
    for (int i = 0; i < ICNT; i++) { accp_zero(ca[i]); t0[i] = t1[i] = t2[i] = 0; }
    constexpr int U = ICNT == 4 ? 4 : 8;
    for (u32 it = 0; it < a.J; it += 16 * U) {
        u64 tv[U], av[U][ICNT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            u64 j = jb + it + 16 * u + jl;
            bool in = it + 16 * u + jl < a.J && j < a.n;
            tv[u] = in ? a.tau[j] : 0;
#pragma unroll
            for (int i = 0; i < ICNT; i++) av[u][i] = in ? a.A[((u64)(a.i0 + i) * a.n + j) * D + t] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            u64 j = jb + it + 16 * u + jl;
            int64_t tc = centre(tv[u]);
            if (tc <= -(D / 2) || tc >= D / 2) { atomicOr(a.err, 2u); tc = 0; }
            if (a.i0 == 0 && t == 0 && it + 16 * u + jl < a.J && j < a.n) a.mtau[j] = (int8_t)tc;
            const int src = t - ((int)tc & 15);
#pragma unroll
            for (int i = 0; i < ICNT; i++) {
                accp_mad(ca[i], av[u][i], tv[u]);
                u64 r = __shfl(av[u][i], src & 15, 16);
                add96(t0[i], t1[i], t2[i], src < 0 ? P - r : r);
            }
        }
    }
    u64 *part = a.part + (u64)blk * nout;
#pragma unroll
    for (int i = 0; i < ICNT; i++) {
        u64 c = sum_over_row_lanes(mod_p_192(accp_to_u192(ca[i])), sh, tid);
        u64 m = sum_over_row_lanes(red_word(t2[i], ((u64)t1[i] << 32) | t0[i]), sh, tid);
        if (tid < 16) {
            part[(a.i0 + i) * D + tid] = c;                      // C_Mf
            part[a.kappa * 16 + (a.i0 + i) * D + tid] = m;       // cm_mtau
        }
    }
}

// out[o] = sum over blocks of part[blk][o] mod p.  With split_l != 0 the first nsplit outputs are comM_f coefficients (k, kappa, 16, 16)
// and the thread that finishes one also writes its l gadget digits into tau = split(hconcat(comM_f), n, base, l) (utils.rs:12-43:
// element e of row i -> positions [e l, (e + 1) l) of the decomposed row, digit j of coefficient t at (e l + j) 16 + t).
__global__ __launch_bounds__(64 * RED_WAVES) void k_reduce(const u64 *part, u32 nblk, u32 nout, u64 *out, u32 nsplit, u32 kappa, u32 k, u64 base,
                                                           int sh, u32 l, u64 *tau) {
    __shared__ u64 ps[RED_WAVES][64][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u32 o = blockIdx.x * 64 + lane;
    u64 s0 = 0, s1 = 0;
    if (o < nout) {
        u32 b = wv;
        for (; b + 7 * RED_WAVES < nblk; b += 8 * RED_WAVES) {
            u64 x[8];
#pragma unroll
            for (int q = 0; q < 8; q++) x[q] = part[(u64)(b + q * RED_WAVES) * nout + o];
#pragma unroll
            for (int q = 0; q < 8; q++) { s0 += x[q]; s1 += s0 < x[q]; }
        }
        for (; b < nblk; b += RED_WAVES) {
            u64 x = part[(u64)b * nout + o];
            s0 += x; s1 += s0 < x;
        }
    }
    ps[wv][lane][0] = s0;
    ps[wv][lane][1] = s1;
    __syncthreads();
    if (wv == 0 && o < nout) {
        for (int v = 1; v < RED_WAVES; v++) {
            u64 x = ps[v][lane][0];
            s0 += x; s1 += (s0 < x) + ps[v][lane][1];
        }
        u64 r = red_word(s1, s0);
        out[o] = r;
        if (l && o < nsplit) {
            u32 t = o & 15, c = (o >> 4) & 15, ik = o >> 8, i = ik % kappa, ki = ik / kappa;
            int64_t cur = centre(r);
            u64 pos = (((u64)i * k + ki) * D + c) * (u64)l * D + t;
            for (u32 j = 0; j < l; j++) {
                int64_t dg = digit_step(cur, base, sh);
                tau[pos + (u64)j * D] = dg >= 0 ? (u64)dg : P - (u64)(-dg);
            }
        }
    }
}

// ---- Decomp::decompose (decomp.rs:32-99) ---------------------------------------------------------------------------------------
// two balanced base-B digits of every coefficient: f = F0 + B F1
__global__ void __launch_bounds__(256) k_decompose2(const u64 *f, size_t words, u64 B, int sh, u64 *F0, u64 *F1) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    int64_t cur = centre(f[i]);
    int64_t d0 = digit_step(cur, B, sh), d1 = digit_step(cur, B, sh);
    F0[i] = d0 >= 0 ? (u64)d0 : P - (u64)(-d0);
    F1[i] = d1 >= 0 ? (u64)d1 : P - (u64)(-d1);
}
// coefficient t of a * b in Z_p[X]/(X^16 + 1) for the 16 lanes of a group: lane t holds b[t], aM = a in Montgomery form (a 2^64 mod p)
__device__ __forceinline__ u64 ring_mul_lane(const u64 *aM, u64 b_t, int t) {
    u64 acc = 0;
#pragma unroll
    for (int s = 0; s < D; s++) {
        const u64 bv = __shfl(b_t, (t - s) & 15, 16);
        const u64 pr = mont_mul(aM[s], bv);
        acc = (t - s) < 0 ? (acc >= pr ? acc - pr : acc + (P - pr)) : add_p(acc, pr);
    }
    return acc;
}
// one fix_variables step of `ntab` tables of `len` ring elements each (table tab evaluates at point tab & 1):
// out[tab][j] = in[tab][2j] + r * (in[tab][2j+1] - in[tab][2j]),  rM = [2][16] Montgomery words of this variable's coordinate
__global__ void __launch_bounds__(256) k_ring_fix(const u64 *in, u64 *out, u32 ntab, size_t len, const u64 *rM) {
    __shared__ u64 r_s[2][D];
    if (threadIdx.x < 2 * D) r_s[threadIdx.x / D][threadIdx.x % D] = rM[threadIdx.x];
    __syncthreads();
    const size_t g = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4), half = len / 2;
    const int t = threadIdx.x & 15;
    if (g >= (size_t)ntab * half) return;
    const u32 tab = (u32)(g / half);
    const size_t j = g % half;
    const u64 *src = in + ((size_t)tab * len + 2 * j) * D;
    const u64 lo = src[t], hi = src[D + t];
    const u64 diff = hi >= lo ? hi - lo : hi + (P - lo);
    out[((size_t)tab * half + j) * D + t] = add_p(lo, ring_mul_lane(r_s[tab & 1], diff, t));
}
// the same when both points' coordinate r is a constant polynomial (PlusProver's points are transcript challenges embedded as constants): a scalar product
__global__ void __launch_bounds__(256) k_ring_fix_const(const u64 *in, u64 *out, u32 ntab, size_t len, const u64 *rM) {
    const size_t g = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4), half = len / 2;
    const int t = threadIdx.x & 15;
    if (g >= (size_t)ntab * half) return;
    const u32 tab = (u32)(g / half);
    const size_t j = g % half;
    const u64 *src = in + ((size_t)tab * len + 2 * j) * D;
    const u64 lo = src[t], hi = src[D + t];
    const u64 diff = hi >= lo ? hi - lo : hi + (P - lo);
    out[((size_t)tab * half + j) * D + t] = add_p(lo, mont_mul(rM[(tab & 1) * D], diff));
}
// y = M x for a CSR matrix with ring-element coefficients (valM in Montgomery form); 16 lanes per row
__global__ void __launch_bounds__(256) k_spmv_ring(const u32 *rowptr, const u32 *col, const u64 *valM, const u64 *x, size_t nrows, u64 *y) {
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int t = threadIdx.x & 15;
    if (row >= nrows) return;
    u64 acc = 0;
    for (u32 k = rowptr[row]; k < rowptr[row + 1]; k++) {
        const u64 xv = x[(size_t)col[k] * D + t];
        u64 aM[D];
#pragma unroll
        for (int s = 0; s < D; s++) aM[s] = valM[(size_t)k * D + s];
        acc = add_p(acc, ring_mul_lane(aM, xv, t));
    }
    y[row * D + t] = acc;
}
// the same for a matrix whose coefficients are all CONSTANT polynomials (every R1CS the reference's benches and tests build: identity rows times gadget
// powers b^i): one Montgomery product per non-zero and lane instead of the 16 of a negacyclic product
__global__ void __launch_bounds__(256) k_spmv_ring_const(const u32 *rowptr, const u32 *col, const u64 *valM, const u64 *x, size_t nrows, u64 *y) {
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int t = threadIdx.x & 15;
    if (row >= nrows) return;
    u64 acc = 0;
    for (u32 k = rowptr[row]; k < rowptr[row + 1]; k++) acc = add_p(acc, mont_mul(valM[k], x[(size_t)col[k] * D + t]));      // valM: one word per non-zero (LfpMatrix::valMc)
    y[row * D + t] = acc;
}
// Constant-coefficient matrix times a vector in COMPACT form (Cm::prove's tau and m_tau: one word / one exponent byte per row instead of a 128-byte ring element):
// y[row] = sum_k M[row][col_k] tau[col_k] (a scalar: the product is a constant polynomial) ...
__global__ void __launch_bounds__(256) k_spmv_scalar_const(const u32 *rowptr, const u32 *col, const u64 *valM, const u64 *x, size_t nrows, u64 *y) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    u64 acc = 0;
    for (u32 k = rowptr[row]; k < rowptr[row + 1]; k++) acc = add_p(acc, mont_mul(valM[k], x[col[k]]));
    y[row] = acc;
}
// ... and y[row] = sum_k M[row][col_k] X^e(dig[col_k]) (a ring element: coefficient t collects the non-zeros whose monomial is X^t); thread = (row, coefficient)
__global__ void __launch_bounds__(256) k_spmv_mono_const(const u32 *rowptr, const u32 *col, const u64 *valM, const int8_t *dig, size_t nrows, u64 *y) {
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int t = threadIdx.x & 15;
    if (row >= nrows) return;
    u64 acc = 0;
    for (u32 k = rowptr[row]; k < rowptr[row + 1]; k++) {
        const int8_t d = dig[col[k]];
        const int e = d >= 0 ? d : 16 + d;
        if (e == t) acc = add_p(acc, mont_mul(valM[k], 1));
    }
    y[row * D + t] = acc;
}
// dst[tab] = src for tab in 0..copies-1 (the tables of one vector, one per evaluation point)
__global__ void __launch_bounds__(256) k_replicate(const u64 *src, size_t words, u32 copies, u64 *dst) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    const u64 v = src[i];
    for (u32 c = 0; c < copies; c++) dst[(size_t)c * words + i] = v;
}

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
template <int ICNT>
static void launch_phase1_i(const Phase1Args &a, u32 nblk, hipStream_t s) {
    switch (a.kcnt) {
    case 0: hipLaunchKernelGGL((k_rg_phase1<ICNT, 0>), dim3(nblk), dim3(256), 0, s, a); break;
    case 1: hipLaunchKernelGGL((k_rg_phase1<ICNT, 1>), dim3(nblk), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_rg_phase1<ICNT, 2>), dim3(nblk), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((k_rg_phase1<ICNT, 4>), dim3(nblk), dim3(256), 0, s, a); break;
    }
}
// icnt in {1, 2, 4}, kcnt in {0, 1, 2, 4} (group_size() cuts kappa and k into such groups)
void launch_phase1(const Phase1Args &a, u32 nblk, hipStream_t s) {
    if (a.icnt == 1) launch_phase1_i<1>(a, nblk, s);
    else if (a.icnt == 2) launch_phase1_i<2>(a, nblk, s);
    else launch_phase1_i<4>(a, nblk, s);
}
// m_tau = exp(tau) as exponents for ALL n entries (a sharded prover: phase 2 writes only the rank's rows, the whole vector is the input of the M m_tau rows)
__global__ void __launch_bounds__(256) k_mtau_all(const u64 *tau, u64 n, int8_t *mtau, u32 *err) {
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    int64_t tc = centre(tau[j]);
    if (tc <= -(D / 2) || tc >= D / 2) { atomicOr(err, 2u); tc = 0; }
    mtau[j] = (int8_t)tc;
}
void launch_mtau_all(const u64 *tau, u64 n, int8_t *mtau, u32 *err, hipStream_t s) {
    hipLaunchKernelGGL(k_mtau_all, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tau, n, mtau, err);
}
void launch_phase2(const Phase2Args &a, u32 nblk, hipStream_t s) {
    if (a.icnt == 1) hipLaunchKernelGGL(k_rg_phase2<1>, dim3(nblk), dim3(256), 0, s, a);
    else if (a.icnt == 2) hipLaunchKernelGGL(k_rg_phase2<2>, dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_rg_phase2<4>, dim3(nblk), dim3(256), 0, s, a);
}
void launch_reduce(const u64 *part, u32 nblk, u32 nout, u64 *out, u32 nsplit, u32 kappa, u32 k, u64 base, u32 l, u64 *tau, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce, dim3((nout + 63) / 64), dim3(64 * RED_WAVES), 0, s, part, nblk, nout, out, nsplit, kappa, k, base, log2_exact(base), l, tau);
}
void launch_decompose2(const u64 *f, size_t words, u64 B, u64 *F0, u64 *F1, hipStream_t s) {
    hipLaunchKernelGGL(k_decompose2, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, f, words, B, log2_exact(B), F0, F1);
}
void launch_ring_fix(const u64 *in, u64 *out, u32 ntab, size_t len, const u64 *rM, hipStream_t s, int const_r) {
    const size_t groups = (size_t)ntab * (len / 2);
    if (const_r) hipLaunchKernelGGL(k_ring_fix_const, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, s, in, out, ntab, len, rM);
    else hipLaunchKernelGGL(k_ring_fix, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, s, in, out, ntab, len, rM);
}
void launch_spmv_ring(const u32 *rowptr, const u32 *col, const u64 *valM, const u64 *x, size_t nrows, u64 *y, hipStream_t s, int const_coef) {
    if (const_coef) hipLaunchKernelGGL(k_spmv_ring_const, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, s, rowptr, col, valM, x, nrows, y);
    else hipLaunchKernelGGL(k_spmv_ring, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, s, rowptr, col, valM, x, nrows, y);
}
void launch_spmv_scalar_const(const u32 *rowptr, const u32 *col, const u64 *valM, const u64 *x, size_t nrows, u64 *y, hipStream_t s) {
    hipLaunchKernelGGL(k_spmv_scalar_const, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, s, rowptr, col, valM, x, nrows, y);
}
void launch_spmv_mono_const(const u32 *rowptr, const u32 *col, const u64 *valM, const int8_t *dig, size_t nrows, u64 *y, hipStream_t s) {
    hipLaunchKernelGGL(k_spmv_mono_const, dim3((unsigned)((nrows + 15) / 16)), dim3(256), 0, s, rowptr, col, valM, dig, nrows, y);
}
void launch_replicate(const u64 *src, size_t words, u32 copies, u64 *dst, hipStream_t s) {
    hipLaunchKernelGGL(k_replicate, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, src, words, copies, dst);
}
void launch_tensor_level(const u64 *cur, u64 len, u64 r, u64 *nxt, hipStream_t s) {
    hipLaunchKernelGGL(k_tensor_level, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, cur, len, r, nxt);
}
void launch_tensor_product(const u64 *a, u64 m, const u64 *b, u64 n, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_tensor_product, dim3((unsigned)((m * n + 255) / 256)), dim3(256), 0, s, a, m, b, n, out);
}
}  // namespace lfp

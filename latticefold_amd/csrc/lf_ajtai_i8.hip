// lf_ajtai_i8.hip -- Ajtai commitments of the base-2 digit planes of a witness on the int8 matrix cores (gfx950 v_mfma_i32_16x16x64_i8).
//
// The decomposition step (nifs/decomposition.rs:178-201) commits to K-1 witnesses whose coefficients are the balanced binary digits
// (-1, 0, 1) of one coefficient vector.  y_k[i] = sum_j A[i][j] * f_k[j] in R = Z_p[X]/(X^24 - X^12 + 1) is a genuine matrix product --
// (kappa x N) . (N x (K-1)) over the ring -- and with f_k that small it needs no modular arithmetic at all until the very end:
//   * ring product in COEFFICIENT form: (a * f)[c_out] = sum_{c_in} a[c_in] * Rot(f)[c_in][c_out], Rot(f)[c_in] = X^c_in * f mod Phi,
//     whose entries are sums of at most three digits (|.| <= 3);
//   * the 64-bit coefficients of A are cut into 8 bytes (biased by -128 to fit int8): A becomes a (8 kappa) x (24 N) int8 matrix,
//     Rot(f_k) a (24 N) x (24 (K-1)) int8 matrix, the product an exact int32 GEMM per column chunk;
//   * y = sum_u 2^(8u) (C[(i,u)] + 128 * colsum) mod p, once per output (k_ajtai_i8_reduce), then one CRT of kappa (K-1) elements.
// kappa = 26, K = 16 at 2^20 columns: 208 x 360 outputs, 1.9e12 int8 MACs per decomposition instead of 1.6e10 lazy 64 x 64 MACs on the
// quarter-rate integer multiplier (the 64-bit VALU kernel of rounds 1-3 took 7.1 ms, issue-bound; removed in round 6).  No bit-plane NTTs are needed either.
// General vectors (commit_ntt, Witness::commit) take the same byte planes of A through lf_ajtai_i8g.hip.
//
// Operands.  A is repacked ONCE per matrix (k_ajtai_pack_i8) into MFMA operand order, so a tile of 8 columns (192 inner elements =
// 3 K-steps of 64) is one contiguous block copied to LDS.  Rot(f) is never materialised: X^c_in * f is Toeplitz in (c_out - c_in) apart
// from the two wrap rules of Phi, so per (digit plane, 8 columns) two 47-entry vectors of 8 packed bytes
//     H[d] = f[d] + f[d + 12]               (outputs c_out >= 12)
//     L[d] = f[d] - f[d + 24] - f[d + 36]   (outputs c_out <  12)        (terms outside 0..23 dropped)
// are built with SWAR byte arithmetic, and the B operand of a lane is the two adjacent entries d = c_out - c_in, c_out - c_in - 1.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include "lf_field.cuh"
#include "lf_kernels.h"
#include "lf_ajtai_i8.h"

namespace lf {
static inline size_t cdiv(size_t a, size_t b) { return (a + b - 1) / b; }
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned long long ull;

constexpr int I8_WAVES = 8;          // 2 row groups x 4 column groups
constexpr ull M7 = 0x7f7f7f7f7f7f7f7full, M8 = 0x8080808080808080ull;
__device__ __forceinline__ ull swar_add(ull a, ull b) { return ((a & M7) + (b & M7)) ^ ((a ^ b) & M8); }
__device__ __forceinline__ ull swar_sub(ull a, ull b) { return ((a | M8) - (b & M7)) ^ ((a ^ ~b) & M8); }
// workgroup barrier that waits for LDS traffic only: __syncthreads() also drains the vector-memory counter, which would stall on the
// prefetch of the next A tile at every barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int digit2_i8(int32_t v, u32 k) {
    const u32 m = v < 0 ? 0u - (u32)v : (u32)v;   // unsigned: v = INT32_MIN is a legal digit (B = 2^32)
    int d = (int)((m >> k) & 1);
    return v < 0 ? -d : d;
}

// ---- ring geometry -----------------------------------------------------------------------------------------------------------------
// RD = ring degree (24: Z_p[X]/(X^24 - X^12 + 1), 64-bit p;  72: Z_p[X]/(X^72 - X^36 + 1), 31-bit p), NL = bytes per coefficient (8 / 4).
// Both rings have the form X^RD = X^(RD/2) - 1, so X^c_in * f is Toeplitz in d = c_out - c_in up to the same two wrap rules:
//     H[d] = f[d] + f[d + RD/2]                (outputs c_out >= RD/2)
//     L[d] = f[d] - f[d + RD] - f[d + 3RD/2]   (outputs c_out <  RD/2)        (terms outside 0..RD-1 dropped)
// A tile = 8 columns = 8 RD inner elements = RD/8 K-steps of 64 (a K-step covers 8 consecutive c_in of the 8 columns).

// one launch per row of A: coefficients of row i (element (c, j) at coef[c * cs + j * js], canonical) -> bytes in operand order
// Ab[((T*KS + s)*MT + mt)*1024 + lane*16 + t],  row m = NL i + u = 16 mt + (lane & 15),  c_in = 8 s + 2 (lane >> 4) + (t >> 3),  column 8 T + (t & 7)
__global__ void __launch_bounds__(256) k_ajtai_pack_i8(const u64 *coef, size_t cs, size_t js, size_t n, u32 i, u32 MT, u32 KS, u32 NL, size_t ntiles,
                                                       unsigned char *Ab) {
    size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 per_tile = KS * 4 * NL;
    if (gid >= ntiles * per_tile) return;
    const u32 r = (u32)(gid % per_tile), u = r % NL, g = (r / NL) & 3, s = r / (4 * NL);
    const size_t T = gid / per_tile;
    const u32 m = NL * i + u, mt = m >> 4, lane = g * 16 + (m & 15), c0 = 8 * s + 2 * g;
    unsigned char b[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        size_t j = T * 8 + (t & 7);
        u64 v = j < n ? coef[(size_t)(c0 + (t >> 3)) * cs + j * js] : 0;
        b[t] = (unsigned char)(((v >> (8 * u)) & 0xFF) ^ 0x80);
    }
    uint4 w;
    w.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((u32)b[3] << 24);
    w.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((u32)b[7] << 24);
    w.z = b[8] | (b[9] << 8) | (b[10] << 16) | ((u32)b[11] << 24);
    w.w = b[12] | (b[13] << 8) | (b[14] << 16) | ((u32)b[15] << 24);
    *(uint4 *)(Ab + ((T * KS + s) * MT + mt) * 1024 + lane * 16) = w;
}
void launch_ajtai_pack_i8(const u64 *coef, size_t cs, size_t js, size_t n, u32 i, u32 MT, u32 RD, u32 NL, unsigned char *Ab, hipStream_t s) {
    size_t ntiles = (n + 7) / 8;
    const u32 KS = RD / 8;
    hipLaunchKernelGGL(k_ajtai_pack_i8, dim3((unsigned)cdiv(ntiles * KS * 4 * NL, 256)), dim3(256), 0, s, coef, cs, js, n, i, MT, KS, NL, ntiles, Ab);
}

// The Goldilocks set-up in one pass over a row: NTT form [24][n] -> coefficients (dense 24 x 24 inverse map, as k_icrt_dense) -> operand
// bytes.  Block = 32 columns (4 tiles): inputs and the 24 x 32 coefficients go through LDS, nothing but the bytes is written.
__global__ void __launch_bounds__(256) k_ajtai_icrt_pack_i8(const u64 *mat, const u64 *ntt, size_t n, u32 i, u32 MT, size_t ntiles, unsigned char *Ab) {
    constexpr u32 KS = 3, NL = 8, per_tile = KS * 4 * NL;
    __shared__ u64 M[24 * 24], X[24][32], Cf[24][33];
    for (int t = threadIdx.x; t < 576; t += 256) M[t] = mat[t];
    const size_t j0 = (size_t)blockIdx.x * 32;
    for (int t = threadIdx.x; t < 768; t += 256) {
        const u32 c = t >> 5, jj = t & 31;
        X[c][jj] = j0 + jj < n ? ntt[(size_t)c * n + j0 + jj] : 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 768; t += 256) {
        const u32 r = t >> 5, jj = t & 31;
        Acc a;
        acc_set(a, M[r * 24], X[0][jj]);
#pragma unroll
        for (int c = 1; c < 24; c++) acc_mad(a, M[r * 24 + c], X[c][jj]);
        Cf[r][jj] = acc_reduce(a);
    }
    __syncthreads();
    for (u32 w = threadIdx.x; w < 4 * per_tile; w += 256) {
        const u32 tl = w / per_tile, r = w % per_tile, u = r % NL, g = (r / NL) & 3, s = r / (4 * NL);
        const size_t T = (size_t)blockIdx.x * 4 + tl;
        if (T >= ntiles) continue;
        const u32 m = NL * i + u, mt = m >> 4, lane = g * 16 + (m & 15), c0 = 8 * s + 2 * g;
        u32 o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const u64 v = Cf[c0 + (t >> 3)][tl * 8 + (t & 7)];   // columns past n hold zeros
            o[t >> 2] |= (u32)((((v >> (8 * u)) & 0xFF) ^ 0x80)) << (8 * (t & 3));
        }
        *(uint4 *)(Ab + ((T * KS + s) * MT + mt) * 1024 + lane * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
void launch_ajtai_icrt_pack_i8(const u64 *icrt_mat, const u64 *ntt, size_t n, u32 i, u32 MT, unsigned char *Ab, hipStream_t s) {
    const size_t ntiles = (n + 7) / 8;
    hipLaunchKernelGGL(k_ajtai_icrt_pack_i8, dim3((unsigned)cdiv(ntiles, 4)), dim3(256), 0, s, icrt_mat, ntt, n, i, MT, ntiles, Ab);
}
struct AjtaiI8Args {
    const unsigned char *Ab;
    const int32_t *planes;   // [RD][ld], already offset to this rank's first column
    u32 sides, nchunks;      // specialised kernels: plane groups (1 / 2) and column chunks -- workgroup (group, chunk) writes slot group * nchunks + chunk of part / dsum
    const u32 *bits;         // optional (k_ajtai_i8s): bit-plane form of the witness (lf_sv_rounds.h launch_sv_bits: [RD][bits_rows][bits_nw] words,
    size_t bits_nw;          // row r < bits_rows - 1 = bit r of |v|, last row = sign bits; 32 positions per word) -- digits are then cut from two
    u32 bits_rows;           // words per (plane, coefficient) instead of eight int32 values
    size_t ld, n;
    u32 MT, NT, k0, NP;
    u32 ntiles, tiles_per_wg;
    int32_t *part;           // [wg][MT][NT][64][4]
    int32_t *dsum;           // [wg][NP][RD]
    u32 *sync;               // k_ajtai_i8s with two plane groups: one signed counter per column chunk (zeroed before the launch), see i8s_build; null = no coupling
    u32 couple_w;            // window of the coupling, in tiles
    u32 couple_e;            // the handshake runs every couple_e tiles (a power of two)
};

// dynamic LDS: A tiles 2 x a_lds | V 2 x NP x 2 x 2RD x 8 | D NP x RD x 8 | w 2 x RD x 8 x 4
//
// Per tile of 8 columns: stage A[T+1] (registers) and w[T+2], digits D[T+1], barrier, Toeplitz vectors V[T+1], RD/8 K-steps of MFMAs on
// A[T] / V[T], A[T+1] -> LDS, barrier.  Notes from the tuning (profiles/r02b_i8_notes.txt), Goldilocks shape (RG 2 x CG 4 waves, 7 x 6 tiles):
//  * the 168 accumulators + 32 operand registers + 20 staging registers of a wave fill the 256-register budget of 2 waves / SIMD; every
//    deeper pipeline tried (one barrier per tile, G / M order alternating between the two waves of a SIMD, AGPR-targeted asm loads, a
//    branch-free MFMA block over clamped tiles) spilled 27 - 480 registers: control flow around the MFMA block makes the compiler copy
//    the tied accumulators;
//  * LDS-DMA (global_load_lds) sustained 2.8 B/clk/CU on this stream against 5.4 TB/s for dwordx4 loads, so A is staged through registers;
//  * loads carry no control flow (a guarded load is waited for at its join): the copy of a tile is unconditional and padded, tiles /
//    columns out of range are clamped to valid addresses and masked after the load;
//  * the staging registers are scalars and macros, not an array captured by lambdas (that array lived in scratch: 3.5 GB of writes / launch).
// Template: ring degree RD, row groups RG (x 8/RG column groups = 8 waves), MTW x NTW tiles per wave, ACH 16-byte chunks of an A tile per thread.
// EXACT: every wave that takes this instantiation owns exactly MTW row tiles (and computes all NTW column tiles, clamped past the end):
// the MFMA block is branch-free -- the compiler may then hoist operand loads across it, which sched_barriers after every second row tile
// keep within the register budget (without them: 62 - 85 spilled registers).
// PROF: per-phase shader-clock totals of every wave of workgroup 0 into g_i8_prof (LF_I8_PROF=1; a measurement instantiation, tools/i8_prof.py)
__device__ unsigned long long g_i8_prof[8][8];
// Round 3: ONE barrier per tile and nothing but the stores of the staged tile outside the MFMA block.  In-kernel clocks of the round-2 loop
// (profiles/r03_i8prof_before.txt, cycles per tile and wave at C4): issuing the tile copy's loads 2000 (the eight waves issue them together and
// block on the full vector-memory queue), barrier 500, Toeplitz vectors 2700 (index divisions + LDS latency chains), MFMA block 3750 - 4600,
// LDS stores 400, barrier 100 - 1100: the matrix pipe was busy 43 % of the time.  Now
//   * digits D are double-buffered and built two tiles ahead, so the vectors of tile T+1 (from D[T+1]) and the digits of tile T+2 need no
//     barrier between them and both sit BETWEEN the K-steps of tile T (the partner wave of the SIMD keeps the matrix pipe busy meanwhile);
//   * every (plane, H/L, entry) of the vectors belongs to a fixed thread: its three source offsets (a zero word stands in for "no term") and its
//     sign are computed once, outside the tile loop;
//   * the waves 0-3 issue their share of the next tile's loads at the top of the iteration, the waves 4-7 (the other wave of each SIMD) after the
//     first K-step (LATE): one wave of a SIMD computes while the other waits in the memory queue.
// MTT: row tiles of the whole workgroup when EXACT (a compile-time constant: LDS operand offsets become instruction immediates), else 0.
template <int RD, int RG, int MTW, int NTW, int ACH, bool EXACT, bool LATE, int MTT, bool PROF = false>
__device__ __forceinline__ void i8_run(const AjtaiI8Args &a, unsigned char *smem) {
    constexpr int KS = RD / 8, VS = 2 * RD, HALF = RD / 2, CG = I8_WAVES / RG;
    constexpr int WR = (RD * 8 + 511) / 512;                    // staged witness words per thread and tile
    constexpr int MAXNP = RD == 24 ? 15 : 8;                    // digit planes per launch (ajtai_i8_max_planes): 3 / 8 full rounds of the vector build
    constexpr int DS = RD + 1;                                  // a row of D: RD packed digit words + one zero word
    constexpr int EPP = 2 * VS;                                 // vector entries per plane (H and L)
    // The operand build (digits, vectors) is the work of all 512 threads, in KS pieces between the K-steps.  (Measured and dropped,
    // profiles/r03_i8_notes.txt: the build by the waves 4-7 alone, or split between the wave groups with the waves 4-7 building BEFORE their
    // K-steps so that one wave of a SIMD builds while the other has the matrix pipe: 8300 - 10400 cycles per tile against 7000 -- the wave
    // that starts its K-steps later gets the leftovers of the pipe -- and every branch-free / batched form of the build spilled registers.)
    constexpr int PPR = 512 / EPP;                              // planes per round of the vector build (5 / 1)
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mg = wave / CG, ng = wave % CG;
    const u32 MT = EXACT ? (u32)MTT : a.MT, NT = a.NT, NP = a.NP;
    const u32 chunk = blockIdx.x, slot = chunk;
    const int32_t *planes = a.planes;
    const size_t a_tile = (size_t)KS * MT * 1024;              // bytes of a tile in HBM
    constexpr size_t a_lds = (size_t)ACH * 512 * 16;           // ... and its padded stride in LDS
    unsigned char *Al = smem;                                   // [2][a_lds]
    // (strides of the MAXNP-plane shape whatever NP is: buffer offsets are buffer index x compile-time constant)
    constexpr int VB = MAXNP * 2 * VS, DB = MAXNP * DS;        // words of one V / D buffer
    ull *V = (ull *)(smem + 2 * a_lds);                         // [2][MAXNP][2][VS]
    ull *Dl = V + 2 * VB;                                       // [2][MAXNP][DS]
    int32_t *wl = (int32_t *)(Dl + 2 * DB);                     // [3][RD][8]
    // Per-thread loop constants and the digit sums live in LDS, not in registers: the MFMA block owns the register file (168 accumulators +
    // 24 + 4 operand + 20 staging registers of 256), and what the compiler spills goes to scratch, whose reloads wait -- in order -- for the
    // tile copy's loads in flight.  LDS reads in the operand-build gaps cost a few cycles.
    u32 *cst = (u32 *)(wl + 3 * RD * 8);                        // [GDR][512]: digit sums
    // this wave's tiles
    const u32 mh = (MT + RG - 1) / RG, m_lo = mg * mh, mcnt = m_lo >= MT ? 0 : (MT - m_lo < mh ? MT - m_lo : mh);
    const u32 n_lo = ng * NTW, ncnt = n_lo >= NT ? 0 : (NT - n_lo < NTW ? NT - n_lo : NTW);
    // B operand base offsets (bytes into one V buffer) at K-step 0: entry e0 = RD-1 - c_out + 2 g of (plane, H / L); columns past the end clamp
    // (two 16-bit offsets per register: a V buffer has 2 NP VS 8 <= 36 KB)
    static_assert(2 * MAXNP * VS * 8 < 65536, "16-bit B operand offsets");
    u32 vbp[(NTW + 1) / 2];
#pragma unroll
    for (int ni = 0; ni < NTW; ni++) {
        u32 n = (n_lo + ni) * 16 + (lane & 15);
        if (n >= RD * NP) n = RD * NP - 1;
        u32 p = n / RD, co = n % RD;
        const u32 off = ((p * 2 + (co >= HALF ? 0u : 1u)) * VS + (RD - 1 - co + 2 * (lane >> 4))) * 8;
        if (ni & 1) vbp[ni >> 1] |= off << 16; else vbp[ni >> 1] = off;
    }
    const u32 ab0 = (m_lo * 64 + lane) * 16;   // A operand: this lane in row tile m_lo of a K-step; tile mi adds mi KB
    v4i acc[MTW][NTW];
#pragma unroll
    for (int mi = 0; mi < MTW; mi++)
#pragma unroll
        for (int ni = 0; ni < NTW; ni++) acc[mi][ni] = v4i{0, 0, 0, 0};
    constexpr int GDR = (MAXNP * RD + 511) / 512;               // rounds of the digit build (1 / 2)
    constexpr int C_DACC = 0;                                   // rows of cst

    const u32 T0 = chunk * a.tiles_per_wg;
    const u32 T1 = T0 + a.tiles_per_wg < a.ntiles ? T0 + a.tiles_per_wg : a.ntiles;
    uint4 ar0 = {0, 0, 0, 0}, ar1 = ar0, ar2 = ar0, ar3 = ar0, ar4 = ar0;
    int32_t wreg0 = 0, wreg1 = 0;
    const u32 Tlast = a.ntiles - 1;
#define LF_I8_LOAD_A(T_)                                                                                            \
    do {                                                                                                            \
        const unsigned char *src_ = a.Ab + (size_t)((T_) < Tlast ? (T_) : Tlast) * a_tile + (size_t)tid * 16;       \
        ar0 = *(const uint4 *)(src_);                                                                               \
        ar1 = *(const uint4 *)(src_ + 8192);                                                                        \
        if (ACH > 2) ar2 = *(const uint4 *)(src_ + 2 * 8192);                                                       \
        if (ACH > 3) ar3 = *(const uint4 *)(src_ + 3 * 8192);                                                       \
        if (ACH > 4) ar4 = *(const uint4 *)(src_ + 4 * 8192);                                                       \
    } while (0)
#define LF_I8_STORE_A(buf_)                                                                                         \
    do {                                                                                                            \
        unsigned char *dst_ = Al + (buf_) * a_lds + (size_t)tid * 16;                                               \
        *(uint4 *)(dst_) = ar0;                                                                                     \
        *(uint4 *)(dst_ + 8192) = ar1;                                                                              \
        if (ACH > 2) *(uint4 *)(dst_ + 2 * 8192) = ar2;                                                             \
        if (ACH > 3) *(uint4 *)(dst_ + 3 * 8192) = ar3;                                                             \
        if (ACH > 4) *(uint4 *)(dst_ + 4 * 8192) = ar4;                                                             \
    } while (0)
    // staged witness words: word idx = tid (+ 512) of the tile's [RD][8] block
    const u32 wc0 = (tid >> 3) < (u32)RD ? (tid >> 3) : RD - 1, wc1 = ((tid + 512) >> 3) < (u32)RD ? ((tid + 512) >> 3) : RD - 1;
    // (32-bit byte offsets from the uniform base: a plane set is at most RD ld 4 < 2^32 bytes, checked by the launcher)
    const u32 wo0 = (u32)((size_t)wc0 * a.ld * 4), wo1 = (u32)((size_t)wc1 * a.ld * 4);
    auto load_w = [&](u32 T) {
        size_t j = (size_t)T * 8 + (tid & 7);
        const bool ok = T < T1 && j < a.n;
        const u32 jo = ok ? (u32)j * 4 : 0;
        int32_t v0 = *(const int32_t *)((const char *)planes + (wo0 + jo));
        wreg0 = ok ? v0 : 0;
        if (WR > 1) {
            int32_t v1 = *(const int32_t *)((const char *)planes + (wo1 + jo));
            wreg1 = ok ? v1 : 0;
        }
    };
    auto store_w = [&](u32 buf) {
        if (tid < (u32)RD * 8) wl[buf * RD * 8 + tid] = wreg0;
        if (WR > 1 && tid + 512 < (u32)RD * 8) wl[buf * RD * 8 + tid + 512] = wreg1;
    };
    // digits of a tile: thread idx = tid + 512 r owns (plane p, coefficient c) = (idx / RD, idx % RD) -- recomputed where needed: a
    // division by a constant is a handful of instructions, cheaper than an LDS round trip, and there is no register to keep it in
#pragma unroll
    for (int r = 0; r < GDR; r++) cst[(C_DACC + r) * 512 + tid] = 0;   // digit sums of (plane, c) = idx over this workgroup's columns
    auto gen_d = [&](u32 wbuf, u32 dbuf) {
#pragma unroll
        for (int r = 0; r < GDR; r++) {
            if (tid + 512 * r < NP * RD) {
                u32 idx = tid + 512 * r;
                asm volatile("" : "+v"(idx));   // (opaque: keeps the compiler from hoisting p, c out of the tile loop into registers it does not have)
                const u32 gp = idx / RD, gc = idx % RD;
                const int32_t *wp = wl + wbuf * RD * 8 + gc * 8;
                const int4 w0 = *(const int4 *)(wp), w1 = *(const int4 *)(wp + 4);
                const int32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                ull d = 0;
                int sacc = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    int dg = digit2_i8(w[q], a.k0 + gp);
                    sacc += dg;
                    d |= (ull)(unsigned char)dg << (8 * q);
                }
                Dl[dbuf * DB + gp * DS + gc] = d;
                cst[(C_DACC + r) * 512 + tid] += (u32)sacc;
            }
        }
    };
    // vectors of a tile: thread tid < PPR * EPP owns entry (H / L, e) = (gr / VS, gr % VS), gr = tid % EPP, of the planes tid / EPP + PPR * round;
    // v = D[o0] + D[o1] (H) or D[o0] - D[o1] - D[o2] (L), a missing term reads the zero word at index RD
    auto gen_v_round = [&](u32 dbuf, u32 vbuf, u32 round) {
        // planes tid / EPP + PPR * round < NP; the destination entry (tid / EPP) * EPP + tid % EPP is tid itself
        u32 tv = tid;
        asm volatile("" : "+v"(tv));            // (opaque: as in gen_d)
        const u32 gv_p = tv / EPP, gv_pd = gv_p * DS * 8;
        if (!(tv < (u32)(PPR * EPP) && gv_pd < (NP - PPR * round) * (u32)(DS * 8))) return;
        const u32 gv_r = tv - gv_p * EPP;
        u32 gv_o0 = RD, gv_o1 = RD, gv_o2 = RD;
        const bool gv_sub = gv_r >= (u32)VS;
        const int dl = RD - 1 - (int)(gv_r % VS);
        if (!gv_sub) {          // H: outputs c_out >= RD/2
            if (dl >= -(HALF - 1) && dl <= RD - 1) {
                if (dl >= 0) gv_o0 = dl;
                if (dl <= HALF - 1) gv_o1 = dl + HALF;
            }
        } else {                // L: outputs c_out < RD/2
            if (dl >= -(RD - 1) && dl <= HALF - 1) {
                if (dl >= 0) gv_o0 = dl;
                if (dl <= -1) gv_o1 = dl + RD;
                if (dl <= -(HALF + 1)) gv_o2 = dl + RD + HALF;
            }
        }
        const unsigned char *D = (const unsigned char *)(Dl + dbuf * DB + round * PPR * DS) + gv_pd;
        const ull x0 = *(const ull *)(D + gv_o0 * 8), x1 = *(const ull *)(D + gv_o1 * 8), x2 = *(const ull *)(D + gv_o2 * 8);
        const ull t = swar_add(x1, x2);
        V[vbuf * VB + round * PPR * EPP + tv] = gv_sub ? swar_sub(x0, t) : swar_add(x0, t);
    };
    const u32 gv_rounds = (NP + PPR - 1) / PPR;
    auto gen_v_gap = [&](u32 dbuf, u32 vbuf, u32 gap) {   // the rounds of K-step gap `gap`
        for (u32 round = gap; round < gv_rounds; round += KS) gen_v_round(dbuf, vbuf, round);
    };
    if (T0 < T1) {
        // ---- prologue: A[T0], w[T0 .. T0+2] -> LDS; zero words; D[T0], D[T0+1]; V[T0]
        LF_I8_LOAD_A(T0);
        load_w(T0);
        store_w(0);
        load_w(T0 + 1);
        store_w(1);
        load_w(T0 + 2);
        store_w(2);
        LF_I8_STORE_A(0);
        if (tid < 2 * MAXNP) Dl[tid * DS + RD] = 0;
        lds_barrier();
        gen_d(0, 0);
        gen_d(1, 1);
        lds_barrier();
        for (u32 g = 0; g < (u32)KS; g++) gen_v_gap(0, 0, g);
        lds_barrier();
        unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0;
#define LF_I8_STAMP(i_)                                                                  \
    if (PROF) {                                                                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
        pt[i_] += now_ - pc;                                                             \
        pc = now_;                                                                       \
    }
        if (PROF) pc = __builtin_amdgcn_s_memtime();
        u32 w3 = 0;                                            // (T - T0) % 3: the w slot of tile T; T+2 is in slot (w3 + 2) % 3, T+3 replaces T
        for (u32 T = T0; T < T1; T++) {
            const u32 cur = (T - T0) & 1, nxt = cur ^ 1;
            // state: A[cur], V[cur] = tile T;  D[nxt] = digits of tile T+1;  w slots hold T, T+1, T+2
            if (!LATE) { LF_I8_LOAD_A(T + 1); load_w(T + 3); }
            LF_I8_STAMP(0);     // early loads

            // ---- RD/8 K-steps of 64 inner elements, the next tile's operand build between them
            const unsigned char *Ac = Al + cur * a_lds;
            const unsigned char *Vc = (const unsigned char *)(V + cur * VB);
            // the operand build of the next tiles, cut into KS pieces, piece s after K-step s.  (Running the late waves' pieces BEFORE their
            // K-steps, so that the two waves of a SIMD alternate between the matrix pipe and the VALU / LDS work, cost 22 spilled registers.)
            auto build = [&](int s) {
                if (s == 0) gen_d(w3 >= 1 ? w3 - 1 : 2, cur);       // digits of tile T+2 (w slot (w3 + 2) % 3) replace those of tile T
                gen_v_gap(nxt, nxt, (u32)s);                          // vectors of tile T+1
            };
#pragma unroll
            for (int s = 0; s < KS; s++) {
                v4i b[NTW];
#pragma unroll
                for (int ni = 0; ni < NTW; ni++) {
                    const ull *q = (const ull *)(Vc + ((ni & 1) ? vbp[ni >> 1] >> 16 : vbp[ni >> 1] & 0xFFFF) + s * 64);
                    ull lo = q[0], hi = q[1];
                    b[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
                }
                // EXACT: the A operand of row tile mi+1 is read while the MFMAs of row tile mi run (a rolling pair of operand registers; the
                // sched_barrier after every row tile pins that order and keeps the compiler from hoisting more than one read ahead)
                v4i avn = v4i{0, 0, 0, 0};
                if (EXACT) avn = *(const v4i *)(Ac + (size_t)s * MT * 1024 + ab0);
#pragma unroll
                for (int mi = 0; mi < MTW; mi++) {
                    if (EXACT) {
                        const v4i av = avn;
                        if (mi + 1 < MTW) avn = *(const v4i *)(Ac + (size_t)s * MT * 1024 + ab0 + (mi + 1) * 1024);
#pragma unroll
                        for (int ni = 0; ni < NTW; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[ni], acc[mi][ni], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    } else if (mi < (int)mcnt) {   // (wave-uniform guards)
                        v4i av = *(const v4i *)(Ac + (size_t)s * MT * 1024 + ab0 + mi * 1024);
#pragma unroll
                        for (int ni = 0; ni < NTW; ni++)
                            if (ni < (int)ncnt) acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[ni], acc[mi][ni], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s == 0) LF_I8_STAMP(1);     // K-step 0
                build(s);
                if (s == 0 && LATE) { LF_I8_LOAD_A(T + 1); load_w(T + 3); }   // (after the build: the staging registers are dead until here)
                if (s == 0) LF_I8_STAMP(2);     // first build piece (+ late loads)
                __builtin_amdgcn_sched_barrier(0);
            }
            LF_I8_STAMP(3);     // K-steps 1.. + vectors
            if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            LF_I8_STAMP(4);     // wait for the tile copy's loads
            LF_I8_STORE_A(nxt);
            store_w(w3);        // w[T+3] goes where w[T] was
            LF_I8_STAMP(5);     // LDS stores
            lds_barrier();
            LF_I8_STAMP(6);     // barrier
            w3 = w3 == 2 ? 0 : w3 + 1;
        }
        if (PROF && blockIdx.x == 0 && lane == 0) {
            for (int i = 0; i < 7; i++) g_i8_prof[wave][i] = pt[i];
            g_i8_prof[wave][7] = T1 - T0;
        }
    }
#undef LF_I8_STAMP
#undef LF_I8_LOAD_A
#undef LF_I8_STORE_A
    // ---- partial results of the workgroup
#pragma unroll
    for (int mi = 0; mi < MTW; mi++)
#pragma unroll
        for (int ni = 0; ni < NTW; ni++)
            if (mi < (int)mcnt && ni < (int)ncnt)
                *(v4i *)(a.part + ((((size_t)slot * MT + m_lo + mi) * NT + n_lo + ni) * 64 + lane) * 4) = acc[mi][ni];
#pragma unroll
    for (int r = 0; r < GDR; r++)
        if (tid + 512 * r < NP * RD) a.dsum[(size_t)slot * NP * RD + tid + 512 * r] = (int)cst[(C_DACC + r) * 512 + tid];
}

// Kernel: waves 0-3 run the (MTWA, early loads) instantiation, waves 4-7 -- the second wave of every SIMD -- the (MTWB, late loads) one: two
// copies of the whole loop, each with its own accumulators (an if / else INSIDE the loop would make the compiler copy them; s_barrier counts
// waves, not program counters).  With two row groups of four waves the split is the row-group split (7 + 6 row tiles).
template <int RD, int RG, int MTWA, int MTWB, int NTW, int ACH, bool EXACT, bool PROF = false>
__global__ void __launch_bounds__(64 * I8_WAVES) k_ajtai_i8(AjtaiI8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    static_assert(RG == 1 || RG == 2, "waves 0-3 / 4-7 are the two row groups");
    static_assert(RG == 2 || MTWA == MTWB, "one row group: one tile count");
    constexpr int MTT = EXACT ? (RG == 2 ? MTWA + MTWB : MTWA) : 0;
    if ((threadIdx.x >> 6) < I8_WAVES / 2) i8_run<RD, RG, MTWA, NTW, ACH, EXACT, false, MTT, PROF>(a, smem);
    else i8_run<RD, RG, MTWB, NTW, ACH, EXACT, true, MTT, PROF>(a, smem);
}
// =====================================================================================================================================
// k_ajtai_i8s -- the 24-ring, 13-row-tile shape (kappa 25 / 26) with SPECIALISED waves.  In-kernel clocks of k_ajtai_i8 above
// (profiles/r03_i8prof_after.txt): the matrix pipe is busy only during the K-steps (3 x 1250 of 7000 cycles per tile); the operand build
// and the tile copy cannot hide behind MFMAs of the same wave (in-order issue), and two MFMA waves per SIMD need all 256 registers for
// accumulators, which leaves no room for a deeper pipeline.  Here a workgroup commits at most 8 digit planes (12 column tiles), so
//   * waves 0-3 (one per SIMD) ONLY multiply: (7 | 6) x 6 tiles = 168 | 144 accumulators, B operands double-buffered across the K-steps,
//     A operand rolling;
//   * waves 4-7 (the other wave of each SIMD) ONLY produce: tile copy global -> registers -> LDS two tiles ahead (no accumulators, so the
//     staging registers are free), digits two tiles ahead, Toeplitz vectors one tile ahead;
// one s_barrier per tile hands the buffers over.  The K - 1 = 15 planes of a decomposition are two such workgroups (8 + 7 planes) that stream
// the same tiles of A -- placed on one XCD eight dispatch slots apart, like the two witnesses of the paired launch, so A leaves HBM once.
// =====================================================================================================================================
constexpr int S_NPG = 8;                 // planes per workgroup
constexpr int S_NT = S_NPG * 24 / 16;    // 12 column tiles
constexpr int S_MT = 13;
constexpr int S_ALDS = 10 * 256 * 16;    // padded tile (39 936 bytes) in LDS
constexpr int S_VB = S_NPG * 2 * 48, S_DB = S_NPG * 25;
constexpr int S_NBA = 3;                 // LDS ring of A tiles: tile T is multiplied while tiles T + 1, T + 2 land (LDS-DMA from the copy wave)
constexpr int S_NCH = 3 * S_MT;          // 1 KiB pieces of a tile
constexpr int S_NVT = 192;               // threads of the three vector waves (4 - 6)
size_t ajtai_i8s_lds_bytes() { return S_NBA * (size_t)S_ALDS + 2 * (size_t)S_VB * 8 + 2 * (size_t)S_DB * 8 + 3 * 24 * 8 * 4 + 256 * 4; }
static_assert(S_NBA * S_ALDS + 2 * S_VB * 8 + 2 * S_DB * 8 + 3 * 24 * 8 * 4 + 256 * 4 <= 160 * 1024 && (S_NBA - 2) * S_NCH <= 63, "LDS of a CU / vmcnt range");

#define LF_S_STAMP(i_)                                                                   \
    if (PROF) {                                                                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
        pt[i_] += now_ - pc;                                                             \
        pc = now_;                                                                       \
    }
// NTU <= NTW: column tiles of this wave that hold planes of the launch (the last wave of a 7-plane group: 2 of 3 -- the MFMAs of a tile nobody reads are not issued:
// the kernel runs at the power-managed rate of the matrix pipe, fewer MFMAs are the one thing that buys time)
template <int MTW, int NTW, bool PROF, int NTU = NTW>
__device__ __forceinline__ void i8s_mma(const AjtaiI8Args &a, unsigned char *smem, u32 mg, u32 ng, u32 T0, u32 T1, u32 slot) {
    constexpr int RD = 24, KS = 3, VS = 48, HALF = 12;
    const u32 lane = threadIdx.x & 63;
    const unsigned char *Al = smem;
    const ull *V = (const ull *)(smem + S_NBA * S_ALDS);
    u32 vb[NTU];
#pragma unroll
    for (int ni = 0; ni < NTU; ni++) {
        u32 n = (ng * NTW + ni) * 16 + (lane & 15);       // < 192 = 8 planes x 24: planes past NP hold zero digits
        const u32 p = n / RD, co = n % RD;
        vb[ni] = ((p * 2 + (co >= HALF ? 0u : 1u)) * VS + (RD - 1 - co + 2 * (lane >> 4))) * 8;
    }
    const u32 m_lo = mg * 7, ab0 = (m_lo * 64 + lane) * 16;
    v4i acc[MTW][NTU];
#pragma unroll
    for (int mi = 0; mi < MTW; mi++)
#pragma unroll
        for (int ni = 0; ni < NTU; ni++) acc[mi][ni] = v4i{0, 0, 0, 0};
    if (T0 < T1) { lds_barrier(); lds_barrier(); }          // (the producers' prologue has two barriers of its own: every wave must arrive)
    lds_barrier();                                          // hand-over: A[T0], V[T0] are in buffer 0
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0, pc0 = 0, pr0 = 0;
    if (PROF) { pc = pc0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
    for (u32 T = T0; T < T1; T++) {
        const u32 cur = (T - T0) & 1;
        const unsigned char *Ac = Al + ((T - T0) % S_NBA) * S_ALDS;
        const unsigned char *Vc = (const unsigned char *)(V + cur * S_VB);
        v4i b[NTU], bn[NTU];
#pragma unroll
        for (int ni = 0; ni < NTU; ni++) {
            const ull *q = (const ull *)(Vc + vb[ni]);
            const ull lo = q[0], hi = q[1];
            b[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
        }
        if (MTW == S_MT) {
            // all 13 row tiles in one wave (the column split): the tile image is dense, piece q = s MTW + mi at 1024 q, and the A operand rolls through two registers
            // ACROSS the K-steps -- the read of piece q + 2 is issued before the MFMAs of piece q (an LDS read takes longer than one piece's MFMAs while the producer
            // waves store the next tile), only the first two reads of a tile are waited for
            v4i avn = *(const v4i *)(Ac + ab0), avnn = *(const v4i *)(Ac + ab0 + 1024);
#pragma unroll
            for (int q = 0; q < KS * MTW; q++) {
                const int s = q / MTW, mi = q % MTW;
                const v4i av = avn;
                avn = avnn;
                if (q + 2 < KS * MTW) avnn = *(const v4i *)(Ac + ab0 + (q + 2) * 1024);
                if (mi == 1 && s + 1 < KS) {               // the next K-step's B operands, behind the first row tiles of this one
#pragma unroll
                    for (int ni = 0; ni < NTU; ni++) {
                        const ull *qv = (const ull *)(Vc + vb[ni] + (s + 1) * 64);
                        const ull lo = qv[0], hi = qv[1];
                        bn[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
                    }
                }
#pragma unroll
                for (int ni = 0; ni < NTU; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (mi == MTW - 1 && s + 1 < KS) {
#pragma unroll
                    for (int ni = 0; ni < NTU; ni++) b[ni] = bn[ni];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                // A operand: three rolling registers -- the read of row tile mi + 2 is issued before the MFMAs of row tile mi (an LDS read takes
                // longer than the 96 cycles of one row tile's MFMAs while the producer waves store the next tile)
                v4i avn = *(const v4i *)(Ac + (size_t)s * S_MT * 1024 + ab0), avnn = *(const v4i *)(Ac + (size_t)s * S_MT * 1024 + ab0 + 1024);
#pragma unroll
                for (int mi = 0; mi < MTW; mi++) {
                    const v4i av = avn;
                    avn = avnn;
                    if (mi + 2 < MTW) avnn = *(const v4i *)(Ac + (size_t)s * S_MT * 1024 + ab0 + (mi + 2) * 1024);
                    if (mi == 1 && s + 1 < KS) {               // the next K-step's B operands, behind the first row tiles of this one
#pragma unroll
                        for (int ni = 0; ni < NTU; ni++) {
                            const ull *q = (const ull *)(Vc + vb[ni] + (s + 1) * 64);
                            const ull lo = q[0], hi = q[1];
                            bn[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < NTU; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[ni], acc[mi][ni], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s + 1 < KS) {
#pragma unroll
                    for (int ni = 0; ni < NTU; ni++) b[ni] = bn[ni];
                }
            }
        }
        LF_S_STAMP(3);     // K-steps
        lds_barrier();
        LF_S_STAMP(6);     // barrier
    }
    if (PROF && blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 7; i++) if (i != 4) g_i8_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
    if (PROF && threadIdx.x == 0 && T0 < T1) {
        // column 4 (unused by this kernel's stamps) carries per-workgroup loop statistics: row 0 = workgroup 0's loop in 100 MHz ticks, row 1 / 2 = max / (2^62 - min)
        // of the loop's shader cycles over the workgroups, row 3 = their sum, row 4 = max of the 100 MHz ticks, row 5 = workgroups counted (the tool zeroes the table first)
        const unsigned long long dc = __builtin_amdgcn_s_memtime() - pc0, dr = __builtin_amdgcn_s_memrealtime() - pr0;
        if (blockIdx.x == 0) g_i8_prof[0][4] = dr;
        atomicMax(&g_i8_prof[1][4], dc);
        atomicMax(&g_i8_prof[2][4], (1ull << 62) - dc);
        atomicAdd(&g_i8_prof[3][4], dc);
        atomicMax(&g_i8_prof[4][4], dr);
        atomicAdd(&g_i8_prof[5][4], 1ull);
    }
#pragma unroll
    for (int mi = 0; mi < MTW; mi++)
#pragma unroll
        for (int ni = 0; ni < NTU; ni++)
            *(v4i *)(a.part + ((((size_t)slot * S_MT + m_lo + mi) * S_NT + ng * NTW + ni) * 64 + lane) * 4) = acc[mi][ni];
}

// ---- copy wave (7): every tile of A straight from HBM / L2 into the LDS ring (global_load_lds_dwordx4: 1 KiB per instruction, no staging registers, no ds_write
// pass -- the register-staged copy cost the four producer waves ~1 000 of their 2 500 busy cycles per tile and the power that goes with them).  The wave touches LDS
// through nothing else: the compiler makes every ds_read of a wave wait for that wave's LDS-DMA in flight.  Counted waits: tile T + 1 has landed when at most one tile's
// pieces are outstanding.
typedef __attribute__((address_space(1))) const void *i8s_gptr;
typedef __attribute__((address_space(3))) void *i8s_lptr;
template <bool PROF>
__device__ __forceinline__ void i8s_copy(const AjtaiI8Args &a, unsigned char *smem, u32 T0, u32 T1) {
    const u32 lane = threadIdx.x & 63;
    const size_t a_tile = (size_t)3 * S_MT * 1024;
    const u32 Tlast = a.ntiles - 1;
    auto issue = [&](u32 T, u32 buf) {
        const unsigned char *src = a.Ab + (size_t)(T < Tlast ? T : Tlast) * a_tile + lane * 16;
        unsigned char *dst = smem + (size_t)buf * S_ALDS;
#pragma unroll
        for (int c = 0; c < S_NCH; c++) __builtin_amdgcn_global_load_lds((i8s_gptr)(src + c * 1024), (i8s_lptr)(dst + c * 1024), 16, 0, 0);
    };
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0;
    if (T0 < T1) {
        issue(T0, 0);
        issue(T0 + 1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S_NBA - 2) * S_NCH) : "memory");   // tile T0 has landed
        lds_barrier();                                      // (the vector waves' two prologue barriers)
        lds_barrier();
    }
    lds_barrier();                                          // hand-over of buffer 0
    if (PROF) pc = __builtin_amdgcn_s_memtime();
    u32 buf = S_NBA - 1;
    for (u32 T = T0; T < T1; T++) {
        issue(T + S_NBA - 1, buf);                          // into the buffer the multipliers left at the last barrier (tile T - 1)
        buf = buf + 1 == (u32)S_NBA ? 0 : buf + 1;
        LF_S_STAMP(0);     // LDS-DMA issue
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S_NBA - 2) * S_NCH) : "memory");   // tile T + 1 has landed
        LF_S_STAMP(5);     // wait for the tile
        lds_barrier();
        LF_S_STAMP(6);     // barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // nothing may land in LDS after the workgroup has gone
    if (PROF && blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 7; i++) if (i != 4) g_i8_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
}

// ---- vector waves (4-6): btid = 0 .. 191 -- digits two tiles ahead, Toeplitz vectors one tile ahead, the pair handshake
template <bool PROF, bool BITS>
__device__ __forceinline__ void i8s_build(const AjtaiI8Args &a, unsigned char *smem, const int32_t *planes, u32 k0, u32 NP, u32 T0, u32 T1, u32 slot) {
    constexpr int RD = 24, VS = 48, HALF = 12, DS = 25, EPP = 96, NVT = S_NVT, NVR = S_VB / S_NVT;
    static_assert(S_VB % S_NVT == 0, "vector entries per thread");
    const u32 btid = threadIdx.x - 256;
    ull *V = (ull *)(smem + S_NBA * S_ALDS);
    ull *Dl = V + 2 * S_VB;
    int32_t *wl = (int32_t *)(Dl + 2 * S_DB);               // [3][24][8]
    u32 *dsl = (u32 *)(wl + 3 * RD * 8);                    // digit sums [192]
    // staged witness words: thread btid < 192 owns word (coefficient btid / 8, column btid % 8) of a tile
    const u32 wo = (u32)((size_t)(btid < 192 ? btid >> 3 : 23) * a.ld * 4);
    int32_t wreg = 0;
    auto load_w = [&](u32 T) {
        const size_t j = (size_t)T * 8 + (btid & 7);
        const bool ok = T < T1 && j < a.n;
        const int32_t v = *(const int32_t *)((const char *)planes + (wo + (ok ? (u32)j * 4 : 0)));
        wreg = ok ? v : 0;
    };
    auto store_w = [&](u32 buf) { if (btid < 192) wl[buf * RD * 8 + btid] = wreg; };
    // digits: thread btid < 192 owns (plane, coefficient) = (btid / 24, btid % 24); planes >= NP get zero digits
    const u32 gp = btid / RD, gc = btid % RD;
    if (btid < 192) dsl[btid] = 0;
    // BITS: the magnitude bits of plane k0 + gp and the sign bits of coefficient gc, 8 columns = one byte of a bit-plane word each
    u32 bmag = 0, bsgn = 0, bsh = 0;   // raw words + shift: cut at use time, so that nothing waits for the loads where they are issued
    const size_t brow_m = ((size_t)(btid < 192 ? gc : 0) * a.bits_rows + (k0 + (gp < NP ? gp : 0))) * a.bits_nw;
    const size_t brow_s = ((size_t)(btid < 192 ? gc : 0) * a.bits_rows + (a.bits_rows - 1)) * a.bits_nw;
    auto load_bits = [&](u32 T) {     // tile T: byte T & 3 of word T >> 2 (tiles past the end: zero digits)
        const u32 Tc = T < T1 ? T : T0;
        bmag = a.bits[brow_m + (Tc >> 2)];
        bsgn = a.bits[brow_s + (Tc >> 2)];
        bsh = T < T1 && gp < NP ? 8 * (T & 3) : 32;          // (32: no digit)
    };
    auto gen_d_bits = [&](u32 dbuf) {
        if (btid < 192) {
            // 4 bits -> 4 bytes: (x * 0x204081) & 0x01010101;  biased digit byte = 4 + bit - 2 (bit & sign)
            const u32 mg8 = bsh < 32 ? (bmag >> bsh) & 0xFF : 0, neg = mg8 & (bsgn >> (bsh & 31));
            const u32 lo = ((mg8 & 15) * 0x204081u) & 0x01010101u, hi = ((mg8 >> 4) * 0x204081u) & 0x01010101u;
            const u32 nlo = ((neg & 15) * 0x204081u) & 0x01010101u, nhi = ((neg >> 4) * 0x204081u) & 0x01010101u;
            const u32 dlo = 0x04040404u + lo - 2 * nlo, dhi = 0x04040404u + hi - 2 * nhi;
            Dl[dbuf * S_DB + gp * DS + gc] = (ull)dlo | ((ull)dhi << 32);
            dsl[btid] += (u32)(__popc(mg8) - 2 * __popc(neg));
        }
    };
    auto gen_d = [&](u32 wbuf, u32 dbuf) {
        if (BITS) { gen_d_bits(dbuf); return; }
        if (btid < 192) {
            const int32_t *wp = wl + wbuf * RD * 8 + gc * 8;
            const int4 w0 = *(const int4 *)(wp), w1 = *(const int4 *)(wp + 4);
            const int32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            ull d = 0x0404040404040404ull;                  // digits are kept BIASED by 4 (a byte in 3 .. 5): sums of up to three need no byte-wise carries
            int sacc = 0;
            if (gp < NP) {
                d = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int dg = digit2_i8(w[q], k0 + gp);
                    sacc += dg;
                    d |= (ull)(unsigned)(dg + 4) << (8 * q);
                }
            }
            Dl[dbuf * S_DB + gp * DS + gc] = d;
            dsl[btid] += (u32)sacc;
        }
    };
    // vectors: entry idx = btid + 192 round (round < 4) = (plane, H / L, e) = (idx / 96, (idx % 96) / 48, idx % 48); with biased digit bytes
    //   H:  v = ((D[o0] + D[o1] + D[o2]) + 0x74..) ^ 0x80..      (bytes 12 + true value + 0x74 = 0x80 + true value)
    //   L:  v = ((D[o0] + 0x84..) - (D[o1] + D[o2])) ^ 0x80..    (a missing term reads the biased zero word)
    // plain 64-bit adds: no byte overflows (every byte stays between 0x7d and 0x8e), and the XOR turns 0x80 + t into the int8 t.
    u32 vo[NVR];        // o0 | o1 << 8 | o2 << 16 | L << 24, offsets in words from the D buffer
#pragma unroll
    for (int r = 0; r < NVR; r++) {
        const u32 idx = btid + NVT * r, vp = idx / EPP, vr = idx % EPP;
        u32 o0 = RD, o1 = RD, o2 = RD;
        const bool vsub = vr >= (u32)VS;
        const int dl = RD - 1 - (int)(vr % VS);
        if (!vsub) {
            if (dl >= -(HALF - 1) && dl <= RD - 1) { if (dl >= 0) o0 = dl; if (dl <= HALF - 1) o1 = dl + HALF; }
        } else if (dl >= -(RD - 1) && dl <= HALF - 1) {
            if (dl >= 0) o0 = dl;
            if (dl <= -1) o1 = dl + RD;
            if (dl <= -(HALF + 1)) o2 = dl + RD + HALF;
        }
        vo[r] = (vp * DS + o0) | ((vp * DS + o1) << 8) | ((vp * DS + o2) << 16) | (vsub ? 1u << 24 : 0u);
    }
    static_assert(S_NPG * 25 <= 256, "8-bit vector source offsets");
    auto gen_v = [&](u32 dbuf, u32 vbuf) {
        const ull *D = Dl + dbuf * S_DB;
        ull xa[NVR], xb[NVR], xc[NVR];
#pragma unroll
        for (int r = 0; r < NVR; r++) { xa[r] = D[vo[r] & 0xFF]; xb[r] = D[(vo[r] >> 8) & 0xFF]; xc[r] = D[(vo[r] >> 16) & 0xFF]; }
#pragma unroll
        for (int r = 0; r < NVR; r++) {
            const ull t = xb[r] + xc[r];
            const ull v = (vo[r] >> 24) ? (xa[r] + 0x8484848484848484ull) - t : (xa[r] + 0x7474747474747474ull) + t;
            V[vbuf * S_VB + btid + NVT * r] = v ^ 0x8080808080808080ull;
        }
    };
    // Coupling of the two plane-group workgroups of a column chunk (blocks b and b ^ 8: same XCD, same tiles of A).  Uncoupled they drift apart -- the
    // L2-resident tiles between a pair are gone after ~5 tile times (14 pairs share the 4 MB L2 of an XCD) -- and the one behind fetches A from HBM a second
    // time: 6.1-7.7 GB per launch measured against the 5.36 GB of ONE pass.  The pair shares a signed counter lead = (tiles issued by group 0) - (tiles issued by
    // group 1): before issuing a tile's loads a workgroup adds +-1 with ONE atomic performed in the XCD's L2 (workgroup scope: no trip to memory -- the agent-scope
    // form of this handshake cost a memory round trip per tile and 0.4 ms per launch), and the value the atomic returns tells it how far AHEAD it is: beyond
    // `couple_w` tiles it waits for its partner.  The returned value is consumed one tile later (the memory counter is in order: a use waits for every load issued
    // before it, so the atomic goes ahead of the tile's ten copy loads).  Only ONE producer wave does this (the last: see CPL0), the others follow through the per-tile barrier.
    // The workgroup BEHIND never waits, so the pair cannot deadlock; a workgroup that is done (or gives up after a bounded spin: the coupling is a traffic
    // optimisation, never a correctness condition -- e.g. if the pair did not share an L2) moves the counter 2^20 tiles to its partner's side.
    // Measured (C4, 224 workgroups, profiles/r04_i8_couple.txt): handshake every tile, window 3: 5.34 GB per launch, 2.14-2.18 ms against 1.99 uncoupled (the
    // atomic's result is waited for in wave 4 every tile); every 4 tiles, window 4 (the default): 5.35 GB, 2.10 against 2.07-2.09 ms on the same box; every 8
    // tiles: 5.7-6.8 GB (the pair drifts out of the L2 between handshakes).  So the second pass over A costs HBM traffic, not kernel time: the loop is bound by
    // the matrix-pipe issue rate and the producers (profiles/r03_i8_notes.txt), and a launch moves 2.6 TB/s either way.
    const u32 c_grp = (blockIdx.x >> 3) & 1, c_chunk = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7);
    constexpr u32 CPL0 = 128;     // the handshake runs in the LAST vector wave (wave 6): its barrier holds the whole workgroup -- the copy wave included -- back when the pair has to wait
    const bool cpl = a.sync != nullptr && a.sides == 2 && btid >= CPL0;
    int *const lead_p = (int *)a.sync + c_chunk;
    const int c_sgn = c_grp ? -1 : 1;
    bool coupled = cpl;
    int c_old = 0;           // (lane 0) the counter before this workgroup's last increment
    bool c_pending = false;
    u32 c_tick = 0;          // tiles since the loop began: the handshake runs every couple_e tiles and counts couple_e tiles at once
#define LF_S_COUPLE_RELEASE()                                                                                                        \
    if (cpl && btid == CPL0) (void)__hip_atomic_fetch_add(lead_p, c_sgn * (1 << 20), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define LF_S_COUPLE_STEP()                                                                                                           \
    if (coupled && ((c_tick++) & (a.couple_e - 1)) == 0) {                                                                           \
        if (c_pending) {                                                                                                             \
            int ahead_ = c_sgn * __builtin_amdgcn_readfirstlane(c_old) + (int)a.couple_e;                                            \
            int spins_ = 0;                                                                                                          \
            while (ahead_ > (int)a.couple_w) {                                                                                       \
                __builtin_amdgcn_s_sleep(8);                                                                                         \
                ahead_ = c_sgn * __hip_atomic_load(lead_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                              \
                if (++spins_ > 8192) { coupled = false; break; }                                                                     \
            }                                                                                                                        \
        }                                                                                                                            \
        if (coupled) {                                                                                                               \
            if (btid == CPL0) c_old = __hip_atomic_fetch_add(lead_p, c_sgn * (int)a.couple_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            c_pending = true;                                                                                                        \
        } else {                                                                                                                     \
            LF_S_COUPLE_RELEASE();                                                                                                   \
        }                                                                                                                            \
    }
    if (cpl && T0 >= T1) { LF_S_COUPLE_RELEASE(); }
    if (btid < 2 * S_NPG) Dl[btid * DS + RD] = 0x0404040404040404ull;   // the (biased) zero words
    if (T0 < T1) {
        // prologue: w[T0 .. T0+2], D[T0], D[T0+1], V[T0]  (two barriers: the multiplier and copy waves arrive at them too)
        if (!BITS) {
            load_w(T0); store_w(0);
            load_w(T0 + 1); store_w(1);
            load_w(T0 + 2); store_w(2);
        }
        lds_barrier();
        if (BITS) { load_bits(T0); gen_d(0, 0); load_bits(T0 + 1); gen_d(0, 1); load_bits(T0 + 2); }
        else { gen_d(0, 0); gen_d(1, 1); }
        lds_barrier();
        gen_v(0, 0);
    }
    lds_barrier();                                          // hand-over of buffer 0 (matches the multipliers' first barrier)
    u32 w3 = 0;
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0;
    if (PROF) pc = __builtin_amdgcn_s_memtime();
    for (u32 T = T0; T < T1; T += 2) {
        // even tile of the pair
        LF_S_COUPLE_STEP();
        if (!BITS) load_w(T + 3);
        LF_S_STAMP(0);     // handshake
        gen_d(w3 >= 1 ? w3 - 1 : 2, 0);                     // digits of tile T+2 -> D[0] (held tile T); BITS: from the words loaded one tile ago
        if (BITS) load_bits(T + 3);
        LF_S_STAMP(1);     // digits
        gen_v(1, 1);                                        // vectors of tile T+1 from D[1]
        LF_S_STAMP(2);     // vectors
        if (!BITS) store_w(w3);
        w3 = w3 == 2 ? 0 : w3 + 1;
        lds_barrier();
        LF_S_STAMP(6);     // barrier
        if (T + 1 >= T1) break;
        // odd tile
        LF_S_COUPLE_STEP();
        if (!BITS) load_w(T + 4);
        LF_S_STAMP(0);
        gen_d(w3 >= 1 ? w3 - 1 : 2, 1);
        if (BITS) load_bits(T + 4);
        LF_S_STAMP(1);
        gen_v(0, 0);
        LF_S_STAMP(2);
        if (!BITS) store_w(w3);
        w3 = w3 == 2 ? 0 : w3 + 1;
        lds_barrier();
        LF_S_STAMP(6);
    }
    if (PROF && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 7; i++) if (i != 4) g_i8_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
    if (coupled) { LF_S_COUPLE_RELEASE(); }   // done: the partner never waits for this workgroup again (a workgroup that gave up has released already)
#undef LF_S_COUPLE_STEP
#undef LF_S_COUPLE_RELEASE
    if (btid < 192) a.dsum[(size_t)slot * 192 + btid] = (int)dsl[btid];   // (stride of 8 planes whatever NP is)
}

// The four multiplier waves split the 12 column tiles: 13 x 3 tiles each, 39 MFMAs per K-step and wave.  (Rounds 3-5 also carried a 2 x 2 split into (7 | 6) x 6
// tiles -- 42 / 36 MFMAs, the 7-row waves bound the tile -- and a build that cut digits from the int32 planes when the bit planes were at hand; the column split
// with bit-plane digits was the fastest form at C4 by 0.1 ms per step, profiles/r05c_i8_ab.txt, and is the only one left.)
template <bool PROF, bool BITS>
__global__ void __launch_bounds__(512) k_ajtai_i8s(AjtaiI8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // plane group g = planes [8 g, 8 g + 8) of the launch: block ids 16 q + 8 g + x, like the two witnesses of the paired launch
    const u32 grp = a.sides == 2 ? (blockIdx.x >> 3) & 1 : 0;
    const u32 chunk = a.sides == 2 ? ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7) : blockIdx.x;
    const u32 slot = grp * a.nchunks + chunk;
    const u32 T0 = chunk * a.tiles_per_wg;
    const u32 T1 = T0 + a.tiles_per_wg < a.ntiles ? T0 + a.tiles_per_wg : a.ntiles;
    const u32 np_g = a.NP - S_NPG * grp < (u32)S_NPG ? a.NP - S_NPG * grp : (u32)S_NPG;
    const u32 wave = threadIdx.x >> 6;
    if (wave == 7) i8s_copy<PROF>(a, smem, T0, T1);
    else if (wave >= 4) i8s_build<PROF, BITS>(a, smem, a.planes, a.k0 + S_NPG * grp, np_g, T0, T1, slot);
    else if (wave == 3 && np_g * 24 <= (3 * 3 + 2) * 16) i8s_mma<13, 3, PROF, 2>(a, smem, 0, wave, T0, T1, slot);   // 7 planes = 168 columns: the group's twelfth column tile is never read
    else i8s_mma<13, 3, PROF>(a, smem, 0, wave, T0, T1, slot);
}

// =====================================================================================================================================
// k_ajtai_i8x -- the specialised-wave design of k_ajtai_i8s with the geometry as template parameters; instantiated for the 72-ring with 4 row
// tiles (BabyBear, kappa 13..16).  In-kernel clocks of the unspecialised kernel at C3 (profiles/r05b_i8prof_c3_before.txt): 12 570 cycles per
// 8-column tile for 360 MFMAs per SIMD (7 344 at the measured issue rate): loads, digits, vectors and stores (~3 000 cycles) run in lock step
// in all eight waves and leave the matrix pipe idle.  Here waves 0-3 (one per SIMD) only multiply -- 4 row tiles x 9 of the 36 column tiles
// each, 144 accumulators -- and waves 4-7 only produce (tile copy two tiles ahead, digits two ahead, Toeplitz vectors one ahead), one s_barrier
// per tile; a workgroup commits 8 planes, the 15 planes of a decomposition are two workgroups paired on one XCD, coupled through an L2 counter.
template <int RD, int MT>
struct SX {
    static constexpr int NPG = 8, KS = RD / 8, VS = 2 * RD, HALF = RD / 2, DS = RD + 1, EPP = 4 * RD;
    static constexpr int NT = NPG * RD / 16;                  // column tiles of a plane group
    static constexpr int TILE = KS * MT * 1024;               // bytes of a tile of A
    static constexpr int NLD = (TILE + 4095) / 4096;          // 16-byte loads per producer thread and tile
    static constexpr int ALDS = NLD * 4096;
    static constexpr int VB = NPG * 2 * VS, DB = NPG * DS;    // 64-bit words
    static constexpr int NW = RD * 8, NWR = (NW + 255) / 256; // staged witness words per tile, rounds of 256 threads
    static constexpr int NDI = NPG * RD, NDR = (NDI + 255) / 256;   // (plane, coefficient) digit items
    static constexpr int NVR = VB / 256;                      // rounds of the vector build
    static_assert(VB % 256 == 0 && NPG * DS < 1024, "vector rounds / 10-bit source offsets");
    static constexpr size_t lds_bytes() { return 2 * (size_t)ALDS + 2 * (size_t)VB * 8 + 2 * (size_t)DB * 8 + 3 * (size_t)NW * 4 + (size_t)NDI * 4; }
};
template <int RD, int MT, int NTW, bool PROF>
__device__ __forceinline__ void i8x_mma(const AjtaiI8Args &a, unsigned char *smem, u32 ng, u32 T0, u32 T1, u32 slot) {
    typedef SX<RD, MT> G;
    constexpr int KS = G::KS, VS = G::VS, HALF = G::HALF;
    const u32 lane = threadIdx.x & 63;
    const unsigned char *Al = smem;
    const ull *V = (const ull *)(smem + 2 * G::ALDS);
    u32 vb[NTW];
#pragma unroll
    for (int ni = 0; ni < NTW; ni++) {
        u32 n = (ng * NTW + ni) * 16 + (lane & 15);       // < 8 planes x RD: planes past NP hold zero digits
        const u32 p = n / RD, co = n % RD;
        vb[ni] = ((p * 2 + (co >= (u32)HALF ? 0u : 1u)) * VS + (RD - 1 - co + 2 * (lane >> 4))) * 8;
    }
    const u32 ab0 = lane * 16;
    v4i acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; mi++)
#pragma unroll
        for (int ni = 0; ni < NTW; ni++) acc[mi][ni] = v4i{0, 0, 0, 0};
    if (T0 < T1) { lds_barrier(); lds_barrier(); }          // (the producers' prologue has two barriers of its own: every wave must arrive)
    lds_barrier();                                          // hand-over: A[T0], V[T0] are in buffer 0
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0;
    if (PROF) pc = __builtin_amdgcn_s_memtime();
    for (u32 T = T0; T < T1; T++) {
        const u32 cur = (T - T0) & 1;
        const unsigned char *Ac = Al + cur * G::ALDS;
        const unsigned char *Vc = (const unsigned char *)(V + cur * G::VB);
        v4i b[NTW], bn[NTW];
#pragma unroll
        for (int ni = 0; ni < NTW; ni++) {
            const ull *q = (const ull *)(Vc + vb[ni]);
            const ull lo = q[0], hi = q[1];
            b[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
        }
        {   // the A operand rolls through two registers ACROSS the K-steps (the tile image is dense: piece q = s MT + mi at 1024 q): the read of piece q + 2 is issued
            // before the MFMAs of piece q, so only the first two reads of a tile are waited for -- restarting the roll at every K-step exposed an LDS latency KS times per tile
            v4i avn = *(const v4i *)(Ac + ab0), avnn = *(const v4i *)(Ac + ab0 + 1024);
#pragma unroll
            for (int q = 0; q < KS * MT; q++) {
                const int s = q / MT, mi = q % MT;
                const v4i av = avn;
                avn = avnn;
                if (q + 2 < KS * MT) avnn = *(const v4i *)(Ac + ab0 + (q + 2) * 1024);
                if (mi == 1 && s + 1 < KS) {               // the next K-step's B operands, behind the first row tiles of this one
#pragma unroll
                    for (int ni = 0; ni < NTW; ni++) {
                        const ull *qv = (const ull *)(Vc + vb[ni] + (s + 1) * 64);
                        const ull lo = qv[0], hi = qv[1];
                        bn[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
                    }
                }
#pragma unroll
                for (int ni = 0; ni < NTW; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[ni], acc[mi][ni], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (mi == MT - 1 && s + 1 < KS) {
#pragma unroll
                    for (int ni = 0; ni < NTW; ni++) b[ni] = bn[ni];
                }
            }
        }
        LF_S_STAMP(3);     // K-steps
        lds_barrier();
        LF_S_STAMP(6);     // barrier
    }
    if (PROF && blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 7; i++) g_i8_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
#pragma unroll
    for (int mi = 0; mi < MT; mi++)
#pragma unroll
        for (int ni = 0; ni < NTW; ni++)
            *(v4i *)(a.part + ((((size_t)slot * MT + mi) * G::NT + ng * NTW + ni) * 64 + lane) * 4) = acc[mi][ni];
}
// the producer waves: btid = 0 .. 255
template <int RD, int MT, bool PROF>
__device__ __forceinline__ void i8x_build(const AjtaiI8Args &a, unsigned char *smem, const int32_t *planes, u32 k0, u32 NP, u32 T0, u32 T1, u32 slot) {
    typedef SX<RD, MT> G;
    constexpr int VS = G::VS, HALF = G::HALF, DS = G::DS, EPP = G::EPP, NLD = G::NLD, NWR = G::NWR, NDR = G::NDR, NVR = G::NVR, NW = G::NW, NDI = G::NDI;
    constexpr int LA = NLD < 2 ? NLD : 2, LB = NLD < 6 ? NLD : 6;      // the tile copy's loads in three groups: [0, LA), [LA, LB), [LB, NLD)
    const u32 btid = threadIdx.x - 256;
    unsigned char *Al = smem;
    ull *V = (ull *)(smem + 2 * G::ALDS);
    ull *Dl = V + 2 * G::VB;
    int32_t *wl = (int32_t *)(Dl + 2 * G::DB);               // [3][RD][8]
    u32 *dsl = (u32 *)(wl + 3 * NW);                        // digit sums [NDI]
    const size_t a_tile = (size_t)G::TILE;
    const u32 Tlast = a.ntiles - 1;
    // (named scalars, not arrays: a staging array that a lambda or a loop can address ends up in scratch memory)
    uint4 x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, y0, y1, y2, y3, y4, y5, y6, y7, y8, y9;
    static_assert(NLD <= 10, "ten staging registers per tile copy");
    const u32 voff = btid * 16;
#define LF_X_LD1(P_, q_, I0_, I1_) if ((q_) >= (I0_) && (q_) < (I1_)) P_##q_ = *(const uint4 *)(src_ + (q_) * 4096);
#define LF_X_LOAD(P_, T_, I0_, I1_)                                                                                \
    do {                                                                                                           \
        const unsigned char *src_ = a.Ab + (size_t)((T_) < Tlast ? (T_) : Tlast) * a_tile + voff;                  \
        LF_X_LD1(P_, 0, I0_, I1_) LF_X_LD1(P_, 1, I0_, I1_) LF_X_LD1(P_, 2, I0_, I1_) LF_X_LD1(P_, 3, I0_, I1_) LF_X_LD1(P_, 4, I0_, I1_)      \
        LF_X_LD1(P_, 5, I0_, I1_) LF_X_LD1(P_, 6, I0_, I1_) LF_X_LD1(P_, 7, I0_, I1_) LF_X_LD1(P_, 8, I0_, I1_) LF_X_LD1(P_, 9, I0_, I1_)      \
    } while (0)
#define LF_X_ST1(P_, q_) if ((q_) < NLD) *(uint4 *)(dst_ + (q_) * 4096) = P_##q_;
#define LF_X_STORE(P_, buf_)                                                                                       \
    do {                                                                                                           \
        unsigned char *dst_ = Al + (buf_) * G::ALDS + (size_t)btid * 16;                                           \
        LF_X_ST1(P_, 0) LF_X_ST1(P_, 1) LF_X_ST1(P_, 2) LF_X_ST1(P_, 3) LF_X_ST1(P_, 4) LF_X_ST1(P_, 5) LF_X_ST1(P_, 6) LF_X_ST1(P_, 7) LF_X_ST1(P_, 8) LF_X_ST1(P_, 9) \
    } while (0)
    // staged witness words: item = btid + 256 r < 8 RD is word (coefficient item / 8, column item % 8) of a tile
    static_assert(NWR <= 3, "three staged words per thread");
    int32_t wr0 = 0, wr1 = 0, wr2 = 0;
#define LF_X_LW1(r_, W_)                                                                                           \
    if ((r_) < NWR) {                                                                                              \
        const u32 item_ = btid + 256 * (r_), cf_ = item_ < (u32)NW ? item_ >> 3 : RD - 1;                          \
        const int32_t v_ = *(const int32_t *)((const char *)planes + ((u32)((size_t)cf_ * a.ld * 4) + (ok_ ? (u32)j_ * 4 : 0))); \
        W_ = ok_ ? v_ : 0;                                                                                         \
    }
#define LF_X_LOADW(T_)                                                                                             \
    do {                                                                                                           \
        const size_t j_ = (size_t)(T_) * 8 + (btid & 7);                                                           \
        const bool ok_ = (T_) < T1 && j_ < a.n;                                                                    \
        LF_X_LW1(0, wr0) LF_X_LW1(1, wr1) LF_X_LW1(2, wr2)                                                         \
    } while (0)
#define LF_X_SW1(r_, W_) if ((r_) < NWR && btid + 256 * (r_) < (u32)NW) wl[(buf_) * NW + btid + 256 * (r_)] = W_;
#define LF_X_STOREW(B_)                                                                                            \
    do {                                                                                                           \
        const u32 buf_ = (B_);                                                                                     \
        LF_X_SW1(0, wr0) LF_X_SW1(1, wr1) LF_X_SW1(2, wr2)                                                         \
    } while (0)
    // digits: item = btid + 256 r < 8 RD is (plane, coefficient) = (item / RD, item % RD); planes >= NP get zero digits
#pragma unroll
    for (int r = 0; r < NDR; r++) { const u32 item = btid + 256 * r; if (item < (u32)NDI) dsl[item] = 0; }
    auto gen_d = [&](u32 wbuf, u32 dbuf) {
#pragma unroll
        for (int r = 0; r < NDR; r++) {
            const u32 item = btid + 256 * r;
            if (item < (u32)NDI) {
                const u32 gp = item / RD, gc = item % RD;
                const int32_t *wp = wl + wbuf * NW + gc * 8;
                const int4 w0 = *(const int4 *)(wp), w1 = *(const int4 *)(wp + 4);
                const int32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
                ull d = 0x0404040404040404ull;                  // digits are kept BIASED by 4 (a byte in 3 .. 5): sums of up to three need no byte-wise carries
                int sacc = 0;
                if (gp < NP) {
                    d = 0;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int dg = digit2_i8(w[q], k0 + gp);
                        sacc += dg;
                        d |= (ull)(unsigned)(dg + 4) << (8 * q);
                    }
                }
                Dl[dbuf * G::DB + gp * DS + gc] = d;
                dsl[item] += (u32)sacc;
            }
        }
    };
    // vectors: entry idx = btid + 256 r = (plane, H / L, e) = (idx / EPP, (idx % EPP) / VS, idx % VS), see i8s_build
    u32 vo[NVR];          // o0 | o1 << 10 | o2 << 20 | L << 30, offsets in words from the D buffer
#pragma unroll
    for (int r = 0; r < NVR; r++) {
        const u32 idx = btid + 256 * r, vp = idx / EPP, vr = idx % EPP;
        u32 o0 = RD, o1 = RD, o2 = RD;
        const bool vsub = vr >= (u32)VS;
        const int dl = RD - 1 - (int)(vr % VS);
        if (!vsub) {
            if (dl >= -(HALF - 1) && dl <= RD - 1) { if (dl >= 0) o0 = dl; if (dl <= HALF - 1) o1 = dl + HALF; }
        } else if (dl >= -(RD - 1) && dl <= HALF - 1) {
            if (dl >= 0) o0 = dl;
            if (dl <= -1) o1 = dl + RD;
            if (dl <= -(HALF + 1)) o2 = dl + RD + HALF;
        }
        vo[r] = (vp * DS + o0) | ((vp * DS + o1) << 10) | ((vp * DS + o2) << 20) | (vsub ? 1u << 30 : 0u);
    }
    auto gen_v = [&](u32 dbuf, u32 vbuf) {
        const ull *D = Dl + dbuf * G::DB;
#pragma unroll
        for (int r = 0; r < NVR; r++) {
            const ull xa = D[vo[r] & 0x3FF], xb = D[(vo[r] >> 10) & 0x3FF], xc = D[(vo[r] >> 20) & 0x3FF];
            const ull t = xb + xc;
            const ull v = (vo[r] >> 30) ? (xa + 0x8484848484848484ull) - t : (xa + 0x7474747474747474ull) + t;
            V[vbuf * G::VB + btid + 256 * r] = v ^ 0x8080808080808080ull;
        }
    };
    // coupling of the two plane-group workgroups of a column chunk: as in i8s_build (the handshake runs in the last producer wave)
    constexpr u32 CPL0 = 192;
    const u32 c_grp = (blockIdx.x >> 3) & 1, c_chunk = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7);
    const bool cpl = a.sync != nullptr && a.sides == 2 && btid >= CPL0;
    int *const lead_p = (int *)a.sync + c_chunk;
    const int c_sgn = c_grp ? -1 : 1;
    bool coupled = cpl;
    int c_old = 0;
    bool c_pending = false;
    u32 c_tick = 0;
#define LF_X_COUPLE_RELEASE()                                                                                                        \
    if (cpl && btid == CPL0) (void)__hip_atomic_fetch_add(lead_p, c_sgn * (1 << 20), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define LF_X_COUPLE_STEP()                                                                                                           \
    if (coupled && ((c_tick++) & (a.couple_e - 1)) == 0) {                                                                           \
        if (c_pending) {                                                                                                             \
            int ahead_ = c_sgn * __builtin_amdgcn_readfirstlane(c_old) + (int)a.couple_e;                                            \
            int spins_ = 0;                                                                                                          \
            while (ahead_ > (int)a.couple_w) {                                                                                       \
                __builtin_amdgcn_s_sleep(8);                                                                                         \
                ahead_ = c_sgn * __hip_atomic_load(lead_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                              \
                if (++spins_ > 8192) { coupled = false; break; }                                                                     \
            }                                                                                                                        \
        }                                                                                                                            \
        if (coupled) {                                                                                                               \
            if (btid == CPL0) c_old = __hip_atomic_fetch_add(lead_p, c_sgn * (int)a.couple_e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            c_pending = true;                                                                                                        \
        } else {                                                                                                                     \
            LF_X_COUPLE_RELEASE();                                                                                                   \
        }                                                                                                                            \
    }
    if (cpl && T0 >= T1) { LF_X_COUPLE_RELEASE(); }
    if (btid < 2 * G::NPG) Dl[btid * DS + RD] = 0x0404040404040404ull;   // the (biased) zero words of both D buffers
    if (T0 < T1) {
        // prologue: A[T0] -> buffer 0, A[T0+1] in flight (x), w[T0 .. T0+2], D[T0], D[T0+1], V[T0]
        LF_X_LOAD(y, T0, 0, NLD);
        LF_X_LOADW(T0); LF_X_STOREW(0);
        LF_X_LOADW(T0 + 1); LF_X_STOREW(1);
        LF_X_LOADW(T0 + 2); LF_X_STOREW(2);
        LF_X_STORE(y, 0);
        LF_X_LOAD(x, T0 + 1, 0, NLD);
        lds_barrier();
        gen_d(0, 0); gen_d(1, 1);
        lds_barrier();
        gen_v(0, 0);
    }
    lds_barrier();                                          // hand-over of buffer 0 (matches the multipliers' first barrier)
    u32 w3 = 0;
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0;
    if (PROF) pc = __builtin_amdgcn_s_memtime();
    for (u32 T = T0; T < T1; T += 2) {
        // even tile of the pair: x holds A[T+1]; load A[T+2] into y
        LF_X_COUPLE_STEP();
        LF_X_LOADW(T + 3);
        LF_X_LOAD(y, T + 2, 0, LA);
        LF_S_STAMP(0);     // load issue
        gen_d(w3 >= 1 ? w3 - 1 : 2, 0);                     // digits of tile T+2 -> D[0]
        LF_X_LOAD(y, T + 2, LA, LB);
        LF_S_STAMP(1);     // digits
        gen_v(1, 1);                                        // vectors of tile T+1 from D[1]
        LF_X_LOAD(y, T + 2, LB, NLD);
        LF_S_STAMP(2);     // vectors
        LF_X_STORE(x, 1);                                   // A[T+1] -> buffer 1
        LF_X_STOREW(w3);
        w3 = w3 == 2 ? 0 : w3 + 1;
        LF_S_STAMP(5);     // wait + LDS stores
        lds_barrier();
        LF_S_STAMP(6);     // barrier
        if (T + 1 >= T1) break;
        // odd tile: y holds A[T+2]; load A[T+3] into x
        LF_X_COUPLE_STEP();
        LF_X_LOADW(T + 4);
        LF_X_LOAD(x, T + 3, 0, LA);
        LF_S_STAMP(0);
        gen_d(w3 >= 1 ? w3 - 1 : 2, 1);
        LF_X_LOAD(x, T + 3, LA, LB);
        LF_S_STAMP(1);
        gen_v(0, 0);
        LF_X_LOAD(x, T + 3, LB, NLD);
        LF_S_STAMP(2);
        LF_X_STORE(y, 0);
        LF_X_STOREW(w3);
        w3 = w3 == 2 ? 0 : w3 + 1;
        LF_S_STAMP(5);
        lds_barrier();
        LF_S_STAMP(6);
    }
    if (PROF && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 7; i++) g_i8_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
    if (coupled) { LF_X_COUPLE_RELEASE(); }
#undef LF_X_COUPLE_STEP
#undef LF_X_COUPLE_RELEASE
#undef LF_X_LOAD
#undef LF_X_STORE
#undef LF_X_LD1
#undef LF_X_ST1
#undef LF_X_LW1
#undef LF_X_LOADW
#undef LF_X_SW1
#undef LF_X_STOREW
#pragma unroll
    for (int r = 0; r < NDR; r++) { const u32 item = btid + 256 * r; if (item < (u32)NDI) a.dsum[(size_t)slot * NDI + item] = (int)dsl[item]; }
}
template <int RD, int MT, bool PROF>
__global__ void __launch_bounds__(512) k_ajtai_i8x(AjtaiI8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef SX<RD, MT> G;
    const u32 grp = a.sides == 2 ? (blockIdx.x >> 3) & 1 : 0;
    const u32 chunk = a.sides == 2 ? ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7) : blockIdx.x;
    const u32 slot = grp * a.nchunks + chunk;
    const u32 T0 = chunk * a.tiles_per_wg;
    const u32 T1 = T0 + a.tiles_per_wg < a.ntiles ? T0 + a.tiles_per_wg : a.ntiles;
    const u32 np_g = a.NP - G::NPG * grp < (u32)G::NPG ? a.NP - G::NPG * grp : (u32)G::NPG;
    const u32 wave = threadIdx.x >> 6;
    if (wave >= 4) i8x_build<RD, MT, PROF>(a, smem, a.planes, a.k0 + G::NPG * grp, np_g, T0, T1, slot);
    else i8x_mma<RD, MT, G::NT / 4, PROF>(a, smem, wave, T0, T1, slot);
}

// copies the per-phase clock totals of the last PROF launch: out[wave][0..6] cycles per phase, out[wave][7] = tiles
int ajtai_i8_read_prof(unsigned long long *out64) { return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_i8_prof), sizeof(g_i8_prof)) == hipSuccess ? 0 : -1; }

// stage 1 of the reduction: element-wise sum of the workgroups' partial tiles (and of their digit sums) -- coalesced across threads
// (blockIdx.y = side: the slots of side s start at s * nwg, its sums at s * (per_wg + per_wg_d))
__global__ void __launch_bounds__(256) k_ajtai_i8_sum(const int32_t *part, size_t per_wg, const int32_t *dsum, u32 per_wg_d, u32 nwg, long long *sum) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= per_wg + per_wg_d) return;
    part += (size_t)blockIdx.y * nwg * per_wg;
    dsum += (size_t)blockIdx.y * nwg * per_wg_d;
    sum += (size_t)blockIdx.y * (per_wg + per_wg_d);
    const int32_t *src = e < per_wg ? part + e : dsum + (e - per_wg);
    const size_t stride = e < per_wg ? per_wg : per_wg_d;
    long long s = 0;
    u32 w = 0;
    for (; w + 8 <= nwg; w += 8) {
        int32_t x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = src[(size_t)(w + q) * stride];
#pragma unroll
        for (int q = 0; q < 8; q++) s += x[q];
    }
    for (; w < nwg; w++) s += src[(size_t)w * stride];
    sum[e] = s;
}
// signed 128-bit value mod p for a modulus below 2^32
__device__ __forceinline__ u64 s128_mod_small(__int128 v, u64 p) {
    const bool neg = v < 0;
    unsigned __int128 t = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const u64 lo = (u64)t, hi = (u64)(t >> 64);
    const u64 two64 = ((0xFFFFFFFFFFFFFFFFull % p) + 1) % p;
    u64 r = ((hi % p) * two64 + lo % p) % p;
    return neg ? (p - r) % p : r;
}
// stage 2: y[plane][row][c_out] in coefficient form, canonical.  Element index e = plane * kappa_total + row0 + i;
// soa != 0: out[c_out * NE + e] (NE = NP * kappa_total), else out[e * RD + c_out].  p_small = 0: the Goldilocks modulus.
__global__ void __launch_bounds__(256) k_ajtai_i8_finish(const long long *sum, size_t per_wg, u32 MT, u32 NT, u32 NP, u32 kappa, u32 row0, u32 kappa_total,
                                                         u32 RD, u32 NL, u64 p_small, int soa, u64 *coef_out, u32 np_total, u32 grp_planes) {
    const u32 o = blockIdx.x * 256 + threadIdx.x;
    if (o >= NP * kappa * RD) return;
    // blockIdx.y = 1: the second plane group of the specialised kernels (planes grp_planes .. np_total - 1 of the same output)
    if (blockIdx.y) sum += per_wg + (size_t)NP * RD;
    const u32 co = o % RD, i = (o / RD) % kappa, p = o / (RD * kappa), HALF = RD / 2;
    const u32 p_glob = p + blockIdx.y * grp_planes;
    if (grp_planes && p_glob >= np_total) return;
    const u32 n = p * RD + co, nt = n >> 4, col = n & 15;
    // T = sum over inner elements of Rot(F)[.][c_out], F = sum over columns of the digit polynomials: the "-128" bias of the bytes of A
    const long long *F = sum + per_wg + (size_t)p * RD;
    long long Tsum = 0;
    for (int ci = 0; ci < (int)RD; ci++) {
        int d = (int)co - ci;
        if (co >= HALF) {
            if (d >= 0) Tsum += F[d];
            if (d <= (int)HALF - 1) Tsum += F[d + HALF];
        } else {
            if (d >= 0) Tsum += F[d];
            if (d <= -1) Tsum -= F[d + RD];
            if (d <= -(int)HALF - 1) Tsum -= F[d + RD + HALF];
        }
    }
    __int128 tot = 0;
    for (u32 u = 0; u < NL; u++) {
        const u32 m = NL * i + u, mt = m >> 4, r = m & 15, ln = col + 16 * (r >> 2), reg = r & 3;
        tot += (__int128)(sum[(((size_t)mt * NT + nt) * 64 + ln) * 4 + reg] + 128 * Tsum) << (8 * u);
    }
    const u64 val = p_small ? s128_mod_small(tot, p_small) : fq_from_s128((u64)tot, (int64_t)(tot >> 64));
    const size_t e = (size_t)(grp_planes ? p_glob : p) * kappa_total + row0 + i;
    if (soa) coef_out[(size_t)co * ((size_t)(grp_planes ? np_total : NP) * kappa_total) + e] = val;
    else coef_out[e * RD + co] = val;
}

static u32 ach_for(u32 RD, u32 MT) {   // 16-byte chunks of an A tile per thread, instantiated values only
    const size_t bytes = (size_t)(RD / 8) * MT * 1024;
    return bytes <= 2 * 8192 ? 2 : (bytes <= 3 * 8192 ? 3 : 5);
}
u32 ajtai_i8_max_planes(const AjtaiI8Ring &R) { return R.RD == 24 ? 15 : 8; }     // digit planes per launch (accumulators, LDS, rounds of the vector build)
// ... for a given row-tile count: the specialised kernels (k_ajtai_i8s: 24-ring, 13 row tiles; k_ajtai_i8x: 72-ring, 4 row tiles) run two plane groups of 8 in one launch
u32 ajtai_i8_max_planes_mt(const AjtaiI8Ring &R, u32 MT) { return R.RD == 72 && MT == 4 ? 15u : ajtai_i8_max_planes(R); }
size_t ajtai_i8_lds_bytes(const AjtaiI8Ring &R, u32 MT, u32 NP) {
    (void)NP;   // the buffers have the strides of the largest plane count
    const size_t maxnp = ajtai_i8_max_planes(R);
    return 2 * (size_t)ach_for(R.RD, MT) * 8192 + 2 * maxnp * 2 * (2 * R.RD) * 8 + 2 * maxnp * (R.RD + 1) * 8 + 3 * (size_t)R.RD * 8 * 4 + 6 * 512 * 4;
}
size_t ajtai_i8_slack_bytes() { return 5 * 8192; }   // readable bytes required behind the packed matrix
u32 ajtai_i8_row_tiles(const AjtaiI8Ring &R, u32 kappa) { return (R.NL * kappa + 15) / 16; }
u32 ajtai_i8_col_tiles(const AjtaiI8Ring &R, u32 NP) { return (R.RD * NP + 15) / 16; }
u32 ajtai_i8_max_rows(const AjtaiI8Ring &R) { return R.RD == 24 ? 26 : 16; }      // rows of A per launch (13 / 4 row tiles)
// partial buffer words (int32) for nwg workgroups
size_t ajtai_i8_part_words(u32 nwg, u32 MT, u32 NT) { return (size_t)nwg * MT * NT * 256; }
size_t ajtai_i8_sum_words(const AjtaiI8Ring &R, u32 MT, u32 NT, u32 NP) { return 2 * ((size_t)MT * NT * 256 + (size_t)NP * R.RD); }   // (two sides)

int launch_ajtai_i8(const AjtaiI8Ring &R, const unsigned char *Ab, u32 MT, const int32_t *planes, size_t ld, size_t n, u32 kappa, u32 row0, u32 kappa_total,
                    u32 k0, u32 NP, u32 nwg, int32_t *part, int32_t *dsum, long long *sum, u64 *coef_out, hipStream_t s, const u32 *bits, size_t bits_nw, u32 bits_rows) {
    AjtaiI8Args a;
    a.bits = nullptr; a.bits_nw = 0; a.bits_rows = 0;
    a.Ab = Ab; a.planes = planes; a.ld = ld; a.n = n;
    a.MT = MT; a.NT = ajtai_i8_col_tiles(R, NP); a.k0 = k0; a.NP = NP;
    a.ntiles = (u32)((n + 7) / 8);
    a.part = part; a.dsum = dsum; a.sync = nullptr; a.couple_w = 0; a.couple_e = 1;
    a.sides = 1;
    if ((size_t)R.RD * ld * 4 >= ((size_t)1 << 32)) return -1;   // 32-bit plane offsets in the kernel
    if ((R.RD != 24 && R.RD != 72) || kappa > ajtai_i8_max_rows(R) || R.NL * kappa > 16 * MT || MT > 13 || NP > ajtai_i8_max_planes_mt(R, MT) || NP == 0 || nwg == 0) return -1;
    static const bool prof = getenv("LF_I8_PROF") != nullptr;
    static const int cw = getenv("LF_I8_COUPLE_W") ? atoi(getenv("LF_I8_COUPLE_W")) : 4;      // window of the pair coupling in tiles (0 switches the coupling off)
    const u32 ce = 4;      // the handshake runs every 4 tiles (measured: every tile costs 0.1 ms per launch, every 8 lets the pair drift out of the L2)
    // The specialised-wave kernels: the 13-row-tile shape of the 24-ring (k_ajtai_i8s) and the 4-row-tile shape of the 72-ring (k_ajtai_i8x).  Plane groups of 8;
    // two groups = paired workgroups on one XCD, coupled through an L2 counter (i8s_build).  Every workgroup of the grid must be resident for the coupling to make
    // progress: at most one per CU.
    const bool spec_s = R.RD == 24 && MT == 13, spec_x = R.RD == 72 && MT == 4;
    if (spec_s || spec_x) {
        typedef SX<72, 4> G;
        const u32 NPG = 8, ndi = spec_s ? 192u : (u32)G::NDI, NTg = spec_s ? (u32)S_NT : (u32)G::NT;
        const u32 groups = NP > NPG ? 2 : 1;
        // column chunks per plane group; two groups: a multiple of 8 (the block id -> (group, chunk) map of the kernels; trailing chunks run empty), so a launch
        // has up to 16 workgroups even when fewer were asked for -- the caller sizes part / dsum for max(nwg, 16) slots
        u32 per = nwg / groups;
        if (groups == 2 && per >= 8) per &= ~7u;
        if (per < 1) per = 1;
        {
            a.tiles_per_wg = (a.ntiles + per - 1) / per;
            u32 nch = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
            if (groups == 2) nch = (nch + 7) & ~7u;
            a.sides = groups; a.nchunks = nch; a.NT = NTg;
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void *)k_ajtai_i8x<72, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_ajtai_i8x<72, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_ajtai_i8s<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_ajtai_i8s<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_ajtai_i8s<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void *)k_ajtai_i8s<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_set = true;
            }
            const dim3 g(groups * nch), b(512);
            // the coupling counters live behind the digit sums this path uses (ndi words per workgroup; the caller sizes dsum for the largest plane count)
            if (groups == 2 && cw > 0 && g.x <= nwg && (size_t)nwg * ndi + g.x <= (size_t)nwg * ajtai_i8_max_planes_mt(R, MT) * R.RD && g.x <= 256) {
                a.sync = (u32 *)(dsum + (size_t)nwg * ndi);
                a.couple_w = (u32)cw;
                a.couple_e = ce;
                (void)hipMemsetAsync(a.sync, 0, (size_t)nch * 4, s);      // one signed counter per column chunk (pair of workgroups)
            }
            if (prof) {
                static const unsigned long long zeros[64] = {0};
                (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_i8_prof), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice, s);
            }
            if (spec_x) {
                if (prof) hipLaunchKernelGGL((k_ajtai_i8x<72, 4, true>), g, b, G::lds_bytes(), s, a);
                else hipLaunchKernelGGL((k_ajtai_i8x<72, 4, false>), g, b, G::lds_bytes(), s, a);
            } else {
                // digits from the bit-plane form when the caller has it and digit plane k0 + NP - 1 is one of its magnitude rows
                a.bits = bits; a.bits_nw = bits_nw; a.bits_rows = bits_rows;
                const bool ub = bits != nullptr && k0 + NP <= bits_rows - 1;
                const size_t lds_s = ajtai_i8s_lds_bytes();
                if (prof && ub) hipLaunchKernelGGL((k_ajtai_i8s<true, true>), g, b, lds_s, s, a);
                else if (prof) hipLaunchKernelGGL((k_ajtai_i8s<true, false>), g, b, lds_s, s, a);
                else if (ub) hipLaunchKernelGGL((k_ajtai_i8s<false, true>), g, b, lds_s, s, a);
                else hipLaunchKernelGGL((k_ajtai_i8s<false, false>), g, b, lds_s, s, a);
            }
            const size_t per_wg = (size_t)MT * NTg * 256;
            hipLaunchKernelGGL(k_ajtai_i8_sum, dim3((unsigned)cdiv(per_wg + ndi, 256), groups), dim3(256), 0, s, part, per_wg, dsum, ndi, nch, sum);
            hipLaunchKernelGGL(k_ajtai_i8_finish, dim3((unsigned)cdiv((size_t)NPG * kappa * R.RD, 256), groups), dim3(256), 0, s, sum, per_wg, MT, NTg, NPG, kappa, row0,
                               kappa_total, R.RD, R.NL, R.p_small, R.soa_out, coef_out, NP, NPG);
            return (int)(groups * nch);
        }
    }
    // every other shape: the generic kernel (all eight waves build and multiply), guarded instantiations by row-tile count
    if (NP > ajtai_i8_max_planes(R)) return -1;
    a.tiles_per_wg = (a.ntiles + nwg - 1) / nwg;
    const u32 nchunks = (a.ntiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    a.nchunks = nchunks;
    const u32 grid = nchunks;
    const size_t lds = ajtai_i8_lds_bytes(R, MT, NP);
#define LF_I8_LAUNCH(RD, RG, MTWA, MTWB, NTW, ACH)                                                                                  \
    do {                                                                                                                           \
        static bool attr_set = false;                                                                                              \
        if (!attr_set) { (void)hipFuncSetAttribute((const void *)k_ajtai_i8<RD, RG, MTWA, MTWB, NTW, ACH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
        hipLaunchKernelGGL((k_ajtai_i8<RD, RG, MTWA, MTWB, NTW, ACH, false>), dim3(grid), dim3(64 * I8_WAVES), lds, s, a);         \
    } while (0)
    if (R.RD == 24) {   // 2 row groups x 4 column groups of 6 tiles (24 x 16 planes = 24 tiles)
        const u32 mh = (MT + 1) / 2;
        if (mh <= 2) LF_I8_LAUNCH(24, 2, 2, 2, 6, 2);
        else if (mh <= 4) LF_I8_LAUNCH(24, 2, 4, 4, 6, 3);
        else LF_I8_LAUNCH(24, 2, 7, 7, 6, 5);
    } else {            // 1 row group (<= 4 row tiles) x 8 column groups of 5 tiles (72 x 8 planes = 36 tiles)
        if (MT <= 1) LF_I8_LAUNCH(72, 1, 1, 1, 5, 2);
        else if (MT <= 2) LF_I8_LAUNCH(72, 1, 2, 2, 5, 3);
        else if (MT <= 4) LF_I8_LAUNCH(72, 1, 4, 4, 5, 5);
        else return -1;
    }
#undef LF_I8_LAUNCH
    const size_t per_wg = (size_t)MT * a.NT * 256;
    hipLaunchKernelGGL(k_ajtai_i8_sum, dim3((unsigned)cdiv(per_wg + NP * R.RD, 256), 1), dim3(256), 0, s, part, per_wg, dsum, NP * R.RD, nchunks, sum);
    hipLaunchKernelGGL(k_ajtai_i8_finish, dim3((unsigned)cdiv((size_t)NP * kappa * R.RD, 256), 1), dim3(256), 0, s, sum, per_wg, MT, a.NT, NP, kappa, row0,
                       kappa_total, R.RD, R.NL, R.p_small, R.soa_out, coef_out, NP, 0u);
    return (int)grid;
}

// =====================================================================================================================================
// MLE evaluations of the witness digit planes on the same matrix cores (Goldilocks):  v[k][c][q] = sum_j eq[q][j] * digit_k(planes[c][j])
// (decomposition.rs:204-211 v_s, and the linearization's v with the coefficients themselves as digits).  An int8 GEMM again: rows =
// (coefficient c, plane k) -- one 16-row MFMA tile per coefficient, its rows the planes -- inner dimension = columns j, columns = the 8
// bytes of the three eq words (biased by -128) plus a column of ones that yields the row sums the bias needs.  No LDS: the A operand is cut
// from 16 consecutive plane entries in registers (the 16 plane lanes of a tile share the loads), the B operand is one 16-byte load from
// the byte-packed eq table (k_eq_pack_i8, once per evaluation point).  Was k_coef_eval: masked +-eq additions on the VALU, 0.72 ms per call
// at 2^20 columns.
// =====================================================================================================================================
// EB[(8 q + u)][j] = byte u of eq[q][j] ^ 0x80 (0x80 = biased zero beyond n, up to the padded length ldb).  wave = (word q, 8 blocks of 64
// columns): coalesced 8-byte loads, the 64 x 8 byte transpose through LDS, 64 contiguous bytes per byte plane out.  (The first version --
// one thread = 16 columns of one word -- read 128 bytes per lane from 64 different cache lines per instruction: 0.8 TB/s, 62 us at 2^20.)
__global__ void __launch_bounds__(256) k_eq_pack_i8(const u64 *eq, size_t ld, size_t n, size_t ldb, unsigned char *EB) {
    __shared__ u64 sm[4][64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr u32 PER_WAVE = 8;
    const size_t blocks = (ldb + 63) / 64, groups = (blocks + PER_WAVE - 1) / PER_WAVE, wid = (size_t)blockIdx.x * 4 + wave;
    if (wid >= 3 * groups) return;                                // (no block-wide barrier below: a wave only reads what it wrote)
    const u32 q = (u32)(wid / groups);
    const u32 u = lane >> 3, c0 = 8 * (lane & 7);                 // lane L writes byte plane u = L / 8, columns 8 (L % 8) .. + 7 of the block
    const unsigned char *src = (const unsigned char *)&sm[wave][0];
    for (u32 k = 0; k < PER_WAVE; k++) {
        const size_t blk = (wid % groups) * PER_WAVE + k;
        if (blk >= blocks) break;
        const size_t j0 = blk * 64, j = j0 + lane;
        sm[wave][lane] = (j < n ? eq[(size_t)q * ld + j] : 0) ^ 0x8080808080808080ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u64 o = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) o |= (u64)src[(c0 + t) * 8 + u] << (8 * t);
        if (j0 + c0 + 8 <= ldb) *(u64 *)(EB + (size_t)(8 * q + u) * ldb + j0 + c0) = o;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// digit of a row: MODE 1 = balanced binary digit k0 + k of |v| with the sign of v (row = plane k, one tile per coefficient);
// MODE 0 = balanced base-256 digit (row & 3) of v: v = sum_k b_k 256^k with b_k in [-128, 127] (4 digits cover |v| <= 127 (256^4 - 1) / 255
// = 2139062143 < 2^31: larger values take the VALU kernel, see `cap` in the launcher; a tile = 4 coefficients x 4 digits)
template <int MODE>
__device__ __forceinline__ int ce_digit(int32_t v, u32 k, u32 k0) {
    if (MODE) return digit2_i8(v, k0 + k);
    int x = v, d = 0;
#pragma unroll
    for (u32 i = 0; i < 4; i++) {
        const int lb = x & 0xFF;
        const int b = lb >= 128 ? lb - 256 : lb;
        d = i == k ? b : d;
        x = (x >> 8) + (lb >= 128 ? 1 : 0);
    }
    return d;
}
struct CoefEvalI8Args {
    const int32_t *planes;      // [24][ldp], offset to the first column of this call
    size_t ldp, n;
    const unsigned char *EB;    // [24][ldb] packed eq bytes of the same columns
    size_t ldb;
    u32 k0, rows;               // first plane of this launch, rows used (<= 16)
    u32 steps_per_wg;           // K-steps (64 columns) per workgroup
    int32_t *part;              // [wg][24][2][64][4]
};
template <int MODE>
__global__ void __launch_bounds__(256) k_coef_eval_i8(CoefEvalI8Args a) {
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 15, g = lane >> 4;
    const size_t nsteps = (a.n + 63) / 64;
    const size_t s0 = (size_t)blockIdx.x * a.steps_per_wg, s1 = s0 + a.steps_per_wg < nsteps ? s0 + a.steps_per_wg : nsteps;
    v4i acc[6][2];
#pragma unroll
    for (int mi = 0; mi < 6; mi++) { acc[mi][0] = v4i{0, 0, 0, 0}; acc[mi][1] = v4i{0, 0, 0, 0}; }
    const bool full16 = (a.ldp & 3) == 0 && (((size_t)a.planes) & 15) == 0;
    for (size_t st = s0; st < s1; st++) {
        const size_t j0 = st * 64 + 16 * g;           // this lane's 16 columns
        // B operands: tile 0 = eq byte planes 0..15, tile 1 = planes 16..23, then the ones column, then zeros
        v4i b0, b1;
        {
            const bool in = j0 + 16 <= a.n;
            const u32 c1 = 16 + row;
            if (in) {
                b0 = *(const v4i *)(a.EB + (size_t)row * a.ldb + j0);
                b1 = c1 < 24 ? *(const v4i *)(a.EB + (size_t)c1 * a.ldb + j0) : (c1 == 24 ? v4i{0x01010101, 0x01010101, 0x01010101, 0x01010101} : v4i{0, 0, 0, 0});
            } else {   // ragged end: the packed table is padded to a multiple of 16 columns with biased zeros; columns past n carry digit 0 anyway
                const size_t jc = j0 < a.ldb ? j0 : 0;
                const bool ok = j0 < a.ldb;
                b0 = ok ? *(const v4i *)(a.EB + (size_t)row * a.ldb + jc) : v4i{0, 0, 0, 0};
                b1 = c1 < 24 ? (ok ? *(const v4i *)(a.EB + (size_t)c1 * a.ldb + jc) : v4i{0, 0, 0, 0}) : (c1 == 24 ? v4i{0x01010101, 0x01010101, 0x01010101, 0x01010101} : v4i{0, 0, 0, 0});
            }
        }
#pragma unroll
        for (int mi = 0; mi < (MODE ? 6 : 2); mi++) {
            // MODE 1: tile = coefficient c, rows = planes;  MODE 0: tile = wave + 4 mi (6 tiles of 4 coefficients x 4 byte digits)
            const u32 tile = MODE ? wave * 6 + mi : wave + 4 * mi;
            if (!MODE && tile >= 6) continue;
            const u32 c = MODE ? tile : tile * 4 + (row >> 2);
            const int32_t *pl = a.planes + (size_t)c * a.ldp + j0;
            int32_t v[16];
            if (full16 && j0 + 16 <= a.n) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int4 w = *(const int4 *)(pl + 4 * q);
                    v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++) v[q] = j0 + q < a.n ? pl[q] : 0;
            }
            u32 w4[4] = {0, 0, 0, 0};
            if (MODE ? row < a.rows : (row & 3) < a.rows) {
#pragma unroll
                for (int q = 0; q < 16; q++) w4[q >> 2] |= (u32)(unsigned char)ce_digit<MODE>(v[q], MODE ? row : (row & 3), a.k0) << (8 * (q & 3));
            }
            const v4i av = v4i{(int)w4[0], (int)w4[1], (int)w4[2], (int)w4[3]};
            acc[mi][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b0, acc[mi][0], 0, 0, 0);
            acc[mi][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b1, acc[mi][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mi = 0; mi < (MODE ? 6 : 2); mi++) {
        const u32 tile = MODE ? wave * 6 + mi : wave + 4 * mi;
        if (!MODE && tile >= 6) continue;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) *(v4i *)(a.part + ((((size_t)blockIdx.x * 24 + tile) * 2 + nt) * 64 + lane) * 4) = acc[mi][nt];
    }
}
// out[(k*24 + c)*3 + q] (MODE 1, k = k0 + row) or out[c*3 + q] accumulated over the byte rows (MODE 0), canonical Goldilocks words
__global__ void __launch_bounds__(256) k_coef_eval_i8_finish(const long long *sum, u32 k0, u32 rows, int mode_bits, u32 nbytes, u64 *out) {
    const u32 o = blockIdx.x * 256 + threadIdx.x;
    const u32 total = mode_bits ? rows * 72 : 72;
    if (o >= total) return;
    const u32 q = o % 3, c = (o / 3) % 24, r = mode_bits ? o / 72 : 0;
    auto cell = [&](u32 rr, u32 col) {   // C[row][col] in the tile layout: mode_bits: tile c, row rr;  mode 0: tile c / 4, row 4 (c % 4) + rr
        const u32 tile = mode_bits ? c : c >> 2, trow = mode_bits ? rr : 4 * (c & 3) + rr;
        const u32 nt = col >> 4, cl = col & 15, ln = cl + 16 * (trow >> 2), reg = trow & 3;
        return sum[(((size_t)tile * 2 + nt) * 64 + ln) * 4 + reg];
    };
    u64 val = 0, pw = 1;                              // mode 0: value = sum over the byte rows of 256^row * (row's sum), in the field
    const u32 r_lo = mode_bits ? r : 0, r_hi = mode_bits ? r + 1 : nbytes;
    for (u32 rr = r_lo; rr < r_hi; rr++) {
        const long long ones = cell(rr, 24);          // sum of the digits of this row: the "-128" bias of the eq bytes
        __int128 t = 0;
        for (u32 u = 0; u < 8; u++) t += (__int128)(cell(rr, 8 * q + u) + 128 * ones) << (8 * u);
        val = fq_add(val, fq_mul(fq_from_s128((u64)t, (int64_t)(t >> 64)), pw));
        pw = fq_mul(pw, 256);
    }
    if (mode_bits) out[((size_t)(k0 + r) * 24 + c) * 3 + q] = val;
    else out[(size_t)c * 3 + q] = val;
}
size_t coef_eval_i8_eb_bytes(size_t n) { return 24 * ((n + 15) / 16 * 16) + 64; }
size_t coef_eval_i8_part_words(u32 nwg) { return (size_t)nwg * 24 * 2 * 256; }
// planes [24][ldp] (n columns from the pointer), eq [3][ldeq] (same columns), K planes (mode_bits) or the coefficients themselves (mode 0, bound =
// max |coefficient|).  EB / part / sum: scratch (coef_eval_i8_eb_bytes, coef_eval_i8_part_words(nwg), 24*2*256 words).  Returns 0, or -1 if the
// shape is not handled (caller falls back to k_coef_eval).
int launch_coef_eval_i8(const int32_t *planes, size_t ldp, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits, u64 bound, unsigned char *EB, u32 nwg,
                        int32_t *part, long long *sum, u64 *out, hipStream_t s) {
    if (!n || (mode_bits && K == 0)) return -1;
    u32 nbytes = 0;
    if (!mode_bits) {   // balanced base-256 digits needed for |v| <= bound
        u64 cap = 127;
        nbytes = 1;
        while (cap < bound && nbytes < 4) { cap = cap * 256 + 127; nbytes++; }
        if (cap < bound) return -1;
    }
    const size_t ldb = (n + 15) / 16 * 16;
    hipLaunchKernelGGL(k_eq_pack_i8, dim3((unsigned)cdiv(3 * cdiv(cdiv(ldb, 64), 8), 4)), dim3(256), 0, s, eq, ldeq, n, ldb, EB);
    const size_t nsteps = (n + 63) / 64;
    if (nwg > nsteps) nwg = (u32)nsteps;
    CoefEvalI8Args a;
    a.planes = planes; a.ldp = ldp; a.n = n; a.EB = EB; a.ldb = ldb;
    a.steps_per_wg = (u32)((nsteps + nwg - 1) / nwg);
    const u32 grid = (u32)((nsteps + a.steps_per_wg - 1) / a.steps_per_wg);
    a.part = part;
    const size_t per_wg = 24 * 2 * 256;
    const u32 total_rows = mode_bits ? K : nbytes;
    for (u32 k0 = 0; k0 < total_rows; k0 += 16) {
        a.k0 = k0;
        a.rows = total_rows - k0 < 16 ? total_rows - k0 : 16;
        if (mode_bits) hipLaunchKernelGGL(k_coef_eval_i8<1>, dim3(grid), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_coef_eval_i8<0>, dim3(grid), dim3(256), 0, s, a);
        hipLaunchKernelGGL(k_ajtai_i8_sum, dim3((unsigned)cdiv(per_wg, 256)), dim3(256), 0, s, part, per_wg, part, 0u, grid, sum);
        hipLaunchKernelGGL(k_coef_eval_i8_finish, dim3((unsigned)cdiv(mode_bits ? a.rows * 72 : 72, 256)), dim3(256), 0, s, sum, k0, a.rows, mode_bits, nbytes, out);
    }
    return 0;
}
}  // namespace lf

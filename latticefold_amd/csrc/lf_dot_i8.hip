// lf_dot_i8.hip -- batched inner products of F_{p^3}-slot vectors on the int8 matrix cores (gfx950 v_mfma_i32_16x16x64_i8):
//     out[a][b][slot] = sum_i X_a[slot][i] * Y_b[slot][i]        (a < na <= 16 vectors X, b < nb <= 3 vectors Y, 8 slots, n columns)
// -- the u_s and eta evaluations of a fold step (decomposition.rs:214-256, folding.rs:236-256: <z_k, M_j^T eq(r)>), was k_dot_batch:
// 48 lazy 64-bit F_{p^3} multiply-accumulates per column and slot on the quarter-rate integer multiplier, 0.48 ms per call at C4.
//
// Every 64-bit word of both operands is written with balanced base-256 digits (d_u in [-128, 127], sum_u d_u 256^u = the word or the word - p:
// byte_u(w + 0x80..80) ^ 0x80, see lf_sv_rounds.hip), so a product of two words is sum_{u,v} 256^(u+v) d_u e_v and the sums over the columns
// of all digit products are an exact int8 GEMM: per (slot, component cz of X) rows = (digit u, vector a) -- 8 row tiles of 16 vectors --,
// inner dimension = columns, matrix columns = (vector b, component cq, digit v) of Y -- 72 = 5 column tiles.  The X digits are cut in
// registers from the 16 words a lane loads (every byte of every word is used once: X streams from HBM exactly once); the Y digits are packed
// once per call (k_dot_pack_y: Y is 1/5 of X).  F_{p^3} structure, powers of 256 and the reduction mod p happen once per output.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "lf_field.cuh"
#include "lf_kernels.h"

namespace lf {
static inline size_t dcdiv(size_t a, size_t b) { return (a + b - 1) / b; }
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 dot_perm(u32 hi, u32 lo, u32 sel) {   // v_perm_b32: selector values 0-3 = bytes of lo, 4-7 = bytes of hi, 12 = 0x00
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    return 0;
#endif
}
// balanced digits of a word: bytes of the result are the digits as int8
__device__ __forceinline__ u64 dot_digits(u64 w) {
    u64 s = w + 0x8080808080808080ull;
    if (s < w) s += 0xFFFFFFFFull;   // wrapped past 2^64: digits of w - p (2^64 - p = 2^32 - 1)
    return s ^ 0x8080808080808080ull;
}

// YB[slot][(b*3 + cq)*8 + v][.] = digit v of Y_b[3 slot + cq][i] in the operand order of dot_load_x per block of 64 columns, zero beyond n
// (ldq columns per row); wave = (word plane, 64 columns)
__global__ void __launch_bounds__(256) k_dot_pack_y(const u64 *Y, size_t ldy, u32 nb, size_t n, size_t lead, size_t ldq, unsigned char *YB) {
    __shared__ u64 sm[4][64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr u32 PER_WAVE = 8;                                   // blocks of 64 columns per wave
    const size_t blocks = ldq / 64, groups = (blocks + PER_WAVE - 1) / PER_WAVE, wid = (size_t)blockIdx.x * 4 + wave;
    if (wid >= (size_t)nb * 24 * groups) return;                 // (no block-wide barrier below: a wave only reads what it wrote)
    const u32 wp = (u32)(wid / groups);                           // word plane b*24 + 3*slot + cq
    const u32 b = wp / 24, slot = (wp % 24) / 3, cq = wp % 3;
    const u64 *src_row = Y + ((size_t)b * 24 + 3 * slot + cq) * ldy;
    unsigned char *dst_rows = YB + ((size_t)slot * 72 + (b * 3 + cq) * 8) * ldq;
    // lane L writes digit plane v = L / 8, operand positions 8 (L % 8) .. + 7
    const u32 v = lane >> 3, c0 = 8 * (lane & 7);
    const unsigned char *src = (const unsigned char *)&sm[wave][0];
    for (u32 k = 0; k < PER_WAVE; k++) {
        const size_t blk = (wid % groups) * PER_WAVE + k;
        if (blk >= blocks) break;
        const size_t i0 = blk * 64, i = i0 + lane;
        sm[wave][lane] = (i >= lead && i < n) ? dot_digits(src_row[i]) : 0;   // (the first `lead` columns belong to the neighbour slice: zero digits)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u64 o = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const u32 pos = c0 + t, col = 8 * ((pos & 15) >> 1) + 2 * (pos >> 4) + (pos & 1);   // operand position 16 g + 2 t + h holds column 8 t + 2 g + h
            o |= (u64)src[col * 8 + v] << (8 * t);
        }
        *(u64 *)(dst_rows + (size_t)v * ldq + i0 + c0) = o;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

struct DotI8Args {
    const u64 *X;               // [na][24][ldx]
    size_t ldx, n;
    u32 na;                     // <= 16
    const unsigned char *YB;    // [8][72][ldq]
    size_t ldq;
    u32 nrows_y;                // 24 nb (<= 72)
    u32 nsteps, steps_per_chunk, chunks;
    int32_t *part;              // [unit 24][chunk][a 8][nt 5][64][4]
};

// Inner-dimension order of a K-step of 64 columns: element e = 2t + h of lane group g is column 8t + 2g + h -- so that ONE load instruction
// (16 bytes per lane) reads 64 contiguous bytes per vector row (the four lane groups of a row side by side) instead of 16 bytes from each of
// 64 different cache lines (one lane = 128 contiguous bytes: 2 TB/s, L1-tag-bound).  k_dot_pack_y stores the Y digits in the same order.
__device__ __forceinline__ void dot_load_x(const u64 *xrow, size_t iw, u32 g, size_t n, bool xlive, u64 (&dst)[16]) {
    if (xlive && iw + 64 <= n) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const ulonglong2 p = *(const ulonglong2 *)(xrow + iw + 8 * t + 2 * g);
            dst[2 * t] = p.x; dst[2 * t + 1] = p.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const size_t i = iw + 8 * (e >> 1) + 2 * g + (e & 1);
            dst[e] = (xlive && i < n) ? xrow[i] : 0;
        }
    }
}
__device__ __forceinline__ void dot_load_y(const unsigned char *yb, size_t ldq, size_t i0, u32 row, u32 nrows_y, v4i (&b)[5]) {
#pragma unroll
    for (int nt = 0; nt < 5; nt++) {
        const u32 r = 16 * nt + row;
        b[nt] = r < nrows_y ? *(const v4i *)(yb + (size_t)r * ldq + i0) : v4i{0, 0, 0, 0};
    }
}
// one K-step: digits of the 16 words, the 8 digit-plane operands (register j of digit u = bytes u of words 4j .. 4j+3), 40 MFMAs
__device__ __forceinline__ void dot_step(v4i (&acc)[8][5], const u64 (&raw)[16], const v4i (&b)[5]) {
    u64 w[16];
#pragma unroll
    for (int t = 0; t < 16; t++) w[t] = dot_digits(raw[t]);
    constexpr u32 SEL_LO = 0x0C0C0400u, SEL_PAIR = 0x05040100u;   // v_perm_b32: byte 0 of the low source, byte 0 of the high source | two bytes of each
#pragma unroll
    for (int u = 0; u < 8; u++) {
        u32 op[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const u32 ub = u & 3;
            const u32 h0 = u < 4 ? (u32)w[4 * j] : (u32)(w[4 * j] >> 32), h1 = u < 4 ? (u32)w[4 * j + 1] : (u32)(w[4 * j + 1] >> 32);
            const u32 h2 = u < 4 ? (u32)w[4 * j + 2] : (u32)(w[4 * j + 2] >> 32), h3 = u < 4 ? (u32)w[4 * j + 3] : (u32)(w[4 * j + 3] >> 32);
            const u32 sel = SEL_LO + ub * 0x0101u;                    // bytes ub of both sources
            const u32 p01 = dot_perm(h1, h0, sel), p23 = dot_perm(h3, h2, sel);
            op[j] = dot_perm(p23, p01, SEL_PAIR);
        }
        const v4i av = v4i{(int)op[0], (int)op[1], (int)op[2], (int)op[3]};
#pragma unroll
        for (int nt = 0; nt < 5; nt++) acc[u][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[nt], acc[u][nt], 0, 0, 0);
    }
}
// wave = one (slot, cz) unit x one chunk of columns.  One wave per SIMD (160 accumulator registers): the loads of the next K-step are in
// flight while this one is computed (two K-steps ahead needed more registers than the file has next to the accumulators).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_dot_i8(DotI8Args a) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, g = lane >> 4;
    // 1-D grid of 24 * nq blocks (nq = chunk quads).  Workgroups go round-robin to the 8 XCDs, and the three cz units of a (slot, chunk quad)
    // read the same Y digits: block L = x + 8 cz + 24 j handles combo x + 8 j, so the three land on one XCD, 8 apart in dispatch order,
    // and share those digits in its L2 (PMC: 1275 -> 1005 MB fetched per launch for 805 + 151 MB of operands; the duration did not move)
    const u32 L = blockIdx.x, combo = (L % 8) + 8 * (L / 24), cz = (L / 8) % 3;
    const u32 slot = combo % 8, unit = slot * 3 + cz;
    const u32 chunk = (combo / 8) * 4 + wave;
    v4i acc[8][5];
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
        for (int nt = 0; nt < 5; nt++) acc[u][nt] = v4i{0, 0, 0, 0};
    if (chunk >= a.chunks) return;
    const u32 s0 = chunk * a.steps_per_chunk, s1 = s0 + a.steps_per_chunk < a.nsteps ? s0 + a.steps_per_chunk : a.nsteps;
    const u64 *xrow = a.X + ((size_t)row * 24 + 3 * slot + cz) * a.ldx;
    const bool xlive = row < a.na;
    const unsigned char *yb = a.YB + (size_t)slot * 72 * a.ldq;
    const u32 last = s1 > s0 ? s1 - 1 : s0;
    auto colw = [&](u32 st) { return (size_t)(st < last ? st : last) * 64; };   // clamped: the tail re-loads the last step
    // TWO K-steps in flight (the accumulators live in AGPRs, so 256 - 153 vector registers were idle: one step ahead left a wave -- the only one of its SIMD --
    // with 8 KB outstanding, and the kernel at 2.5 TB/s)
    u64 xa[16], xn[16], xm[16];
    v4i ba[5], bn[5], bm[5];
    dot_load_x(xrow, colw(s0), g, a.n, xlive, xa); dot_load_y(yb, a.ldq, colw(s0) + 16 * g, row, a.nrows_y, ba);
    dot_load_x(xrow, colw(s0 + 1), g, a.n, xlive && s0 + 1 < s1, xn); dot_load_y(yb, a.ldq, colw(s0 + 1) + 16 * g, row, a.nrows_y, bn);
    // (three buffers in rotation, the loop unrolled by three: a register copy of a buffer would wait for its load at the end of the very step that issued it.  No
    // exits inside the trip -- control flow around the MFMA block makes the compiler copy the tied accumulators: 359 spilled registers --: the last trip is padded
    // with steps whose X words are zero)
    auto ldx = [&](u32 st, u64 (&dst)[16]) { dot_load_x(xrow, colw(st), g, a.n, xlive && st < s1, dst); };
    for (u32 st = s0; st < s1; st += 3) {
        ldx(st + 2, xm); dot_load_y(yb, a.ldq, colw(st + 2) + 16 * g, row, a.nrows_y, bm);   // two K-steps ahead
        dot_step(acc, xa, ba);
        ldx(st + 3, xa); dot_load_y(yb, a.ldq, colw(st + 3) + 16 * g, row, a.nrows_y, ba);
        dot_step(acc, xn, bn);
        ldx(st + 4, xn); dot_load_y(yb, a.ldq, colw(st + 4) + 16 * g, row, a.nrows_y, bn);
        dot_step(acc, xm, bm);
    }
    int32_t *o = a.part + ((size_t)unit * a.chunks + chunk) * (8 * 5 * 256);
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
        for (int nt = 0; nt < 5; nt++) *(v4i *)(o + ((size_t)u * 5 + nt) * 256 + lane * 4) = acc[u][nt];
}

// tot[unit][e] = sum over the chunks (64-bit)
__global__ void __launch_bounds__(256) k_dot_i8_sum(const int32_t *part, u32 chunks, long long *tot) {
    const u32 e = blockIdx.x * 256 + threadIdx.x, unit = blockIdx.y;   // e < 8*5*256
    long long s = 0;
    for (u32 ch = 0; ch < chunks; ch++) s += part[((size_t)unit * chunks + ch) * 10240 + e];
    tot[(size_t)unit * 10240 + e] = s;
}

// block = output (a, b, slot, comp), thread = (cz, digit u):  out[(a*nb + b)*24 + 3*slot + comp] = sum over (cz, cq) with cz + cq = comp (mod 3) of
// nu^[cz+cq >= 3] * sum_{u,v} 256^(u+v) tot[slot, cz][u][a][(b, cq, v)]
__global__ void __launch_bounds__(32) k_dot_i8_finish(const long long *tot, u32 na, u32 nb, u64 nu, u64 *out, u32 nb_out, u32 b0) {
    __shared__ u64 sm[24];
    const u32 o = blockIdx.x, t = threadIdx.x;
    const u32 comp = o % 3, slot = (o % 24) / 3, b = (o / 24) % nb, av = o / (24 * nb);
    if (t < 24) {
        const u32 cz = t >> 3, u = t & 7, cq = (comp + 3 - cz) % 3;
        const long long *tu = tot + (size_t)(slot * 3 + cz) * 10240;
        __int128 inner = 0;
#pragma unroll
        for (u32 v = 0; v < 8; v++) {
            const u32 col = (b * 3 + cq) * 8 + v, nt = col >> 4, cl = col & 15;
            inner += (__int128)tu[((size_t)u * 5 + nt) * 256 + (cl + 16 * (av >> 2)) * 4 + (av & 3)] << (8 * v);
        }
        u64 val = fq_from_s128((u64)inner, (int64_t)(inner >> 64));
        u64 pw = 1;
        for (u32 i = 0; i < u; i++) pw = fq_mul(pw, 256);
        val = fq_mul(val, pw);
        if (cz + cq >= 3) val = fq_mul(val, nu);
        sm[t] = val;
    }
    __syncthreads();
    if (t == 0) {
        u64 res = 0;
        for (int i = 0; i < 24; i++) res = fq_add(res, sm[i]);
        out[((size_t)av * nb_out + b0 + b) * 24 + 3 * slot + comp] = fq_canon(res);   // (nb_out, b0: this launch's Y vectors are b0 .. b0 + nb - 1 of nb_out)
    }
}

size_t dot_i8_yb_bytes(size_t n) { return (size_t)8 * 72 * (dcdiv(n, 64) * 64) + 64; }
static u32 dot_i8_chunks(size_t nsteps) {
    size_t want = 40;   // 24 units x 40 chunks = 240 blocks of 4 waves: ONE batch on 256 CUs at one wave per SIMD (44 chunks = 264 blocks ran as two batches: 2x the time)
    if (want > nsteps) want = nsteps;
    const size_t spc = dcdiv(nsteps, want);
    return (u32)dcdiv(nsteps, spc);
}
// sized for the largest chunk count: dot_i8_chunks is not monotone in the step count (80 steps -> 40 chunks, 81 -> 27), and the launcher may
// start a slice one column early
size_t dot_i8_part_words(size_t) { return (size_t)24 * 40 * 10240; }
size_t dot_i8_tot_words() { return (size_t)24 * 10240; }
// X [na][24][ldx], Y [nb][24][ldy], n columns; out[(a*nb + b)*24 + 3*slot + comp] canonical.  Returns 0, or -1 if the shape is not handled.
// the Y digits alone (k_dot_pack_y), for a following launch_dot_batch_i8(.., y_packed = true) on vectors X with the same alignment: two X sets against the
// same Y (the eta inner products of the two sides of a fold step) pack it once
int launch_dot_pack_y(const u64 *X, const u64 *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, hipStream_t s) {
    if (nb < 1 || nb > 3 || n < 64 || (((size_t)X) & 7)) return -1;
    const size_t lead = (((size_t)X) & 15) ? 1 : 0;
    Y -= lead; n += lead;
    const size_t ldq = dcdiv(n, 64) * 64;
    hipLaunchKernelGGL(k_dot_pack_y, dim3((unsigned)dcdiv((size_t)nb * 24 * dcdiv(ldq / 64, 8), 4)), dim3(256), 0, s, Y, ldy, nb, n, lead, ldq, YB);
    return 0;
}
int launch_dot_batch_i8(const DevCrt &t, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, int32_t *part,
                        long long *tot, u64 *out, hipStream_t s, bool y_packed, u32 nb_out, u32 b0) {
    if (!nb_out) nb_out = nb;
    if (na < 1 || na > 16 || nb < 1 || nb > 3 || n < 64 || (ldx & 1) || (((size_t)X) & 7)) return -1;
    // a column slice that starts at an odd column (a rank's slice of a sharded step): start one column earlier (16-byte aligned loads) and
    // give that column zero digits on the Y side
    const size_t lead = (((size_t)X) & 15) ? 1 : 0;
    X -= lead; Y -= lead; n += lead;
    const size_t ldq = dcdiv(n, 64) * 64;
    // exactness: a wave adds steps_per_chunk * 64 digit products of at most 2^14 into an int32 accumulator
    if (dcdiv(ldq / 64, dot_i8_chunks(ldq / 64)) >= 2048) return -1;
    if (!y_packed) hipLaunchKernelGGL(k_dot_pack_y, dim3((unsigned)dcdiv((size_t)nb * 24 * dcdiv(ldq / 64, 8), 4)), dim3(256), 0, s, Y, ldy, nb, n, lead, ldq, YB);
    DotI8Args a;
    a.X = X; a.ldx = ldx; a.n = n; a.na = na; a.YB = YB; a.ldq = ldq; a.nrows_y = 24 * nb;
    a.nsteps = (u32)(ldq / 64);
    a.chunks = dot_i8_chunks(a.nsteps);
    a.steps_per_chunk = (u32)dcdiv(a.nsteps, a.chunks);
    a.part = part;
    hipLaunchKernelGGL(k_dot_i8, dim3((unsigned)dcdiv(a.chunks, 4) * 24), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_dot_i8_sum, dim3(40, 24), dim3(256), 0, s, part, a.chunks, tot);
    hipLaunchKernelGGL(k_dot_i8_finish, dim3(na * nb * 24), dim3(32), 0, s, tot, na, nb, t.nu, out, nb_out, b0);
    return 0;
}
}  // namespace lf

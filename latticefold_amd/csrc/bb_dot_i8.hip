// bb_dot_i8.hip -- BabyBear: batched inner products of F_{p^9}-slot vectors on the int8 matrix cores (the BabyBear form of lf_dot_i8.hip):
//     out[a][b][slot] = sum_i X_a[slot][i] * Y_b[slot][i]        (a < na <= 16 vectors X, b < nb <= 3 vectors Y, 8 slots, n columns)
// -- u_s and eta of a fold step (<z_k, M_j^T eq(r)>), was lfbb::k_dot_batch: 27 F_{p^9} products per column and slot on v_mad_i64_i32.
//
// A residue is a centred Montgomery word x~ in [-H, H], H < 2^30 (bb_field.cuh).  Its four balanced base-256 digits d_u in [-128, 127],
// sum_u d_u 256^u = x~, are the bytes of (x~ + 0x80808080) ^ 0x80808080, so sum_i x~ y~ = sum_{u,v} 256^(u+v) sum_i d_u e_v: per
// (slot, component cx of X) an exact int8 GEMM with rows = (digit u, vector a) -- 4 row tiles --, inner dimension = columns, matrix columns =
// (vector b, component cy, digit v) of Y -- 108 = 7 column tiles.  X digits are cut in registers from coalesced loads (one instruction = 32
// contiguous bytes per vector row), Y digits are packed once per call in the same inner-dimension order.  The finish undoes the two
// Montgomery factors (sum x~ y~ = R^2 sum x y), applies Y^9 = nu and reduces mod p.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "bb_field.cuh"
#include "bb_kernels.h"

namespace lfbb {
static inline size_t bdiv(size_t a, size_t b) { return (a + b - 1) / b; }
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32 bbd_perm(u32 hi, u32 lo, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    return 0;
#endif
}
__device__ __forceinline__ u32 bbd_digits(fe w) { return ((u32)w + 0x80808080u) ^ 0x80808080u; }   // |w| < 2^30: no wrap

// inner-dimension order of a K-step of 64 columns: element e = 2t + h of lane group g is column 8t + 2g + h
// YB[slot][(b*9 + cy)*4 + v][.] = digit v of Y_b[9 slot + cy][i] in that order per block of 64 columns, zero beyond n and below `lead`
__global__ void __launch_bounds__(256) k_bbdot_pack_y(const fe *Y, size_t ldy, u32 nb, size_t n, size_t lead, size_t ldq, unsigned char *YB) {
    __shared__ u32 sm[4][64];
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr u32 PER_WAVE = 8;
    const size_t blocks = ldq / 64, groups = (blocks + PER_WAVE - 1) / PER_WAVE, wid = (size_t)blockIdx.x * 4 + wave;
    if (wid >= (size_t)nb * RE * groups) return;                 // (no block-wide barrier below: a wave only reads what it wrote)
    const u32 wp = (u32)(wid / groups);                           // word plane b*72 + 9*slot + cy
    const u32 b = wp / RE, slot = (wp % RE) / TAU, cy = wp % TAU;
    const fe *src_row = Y + ((size_t)b * RE + TAU * slot + cy) * ldy;
    unsigned char *dst_rows = YB + ((size_t)slot * 108 + (b * TAU + cy) * 4) * ldq;
    const u32 v = lane >> 4, c0 = 4 * (lane & 15);                // lane L writes digit plane v = L / 16, operand positions 4 (L % 16) .. + 3
    const unsigned char *src = (const unsigned char *)&sm[wave][0];
    for (u32 k = 0; k < PER_WAVE; k++) {
        const size_t blk = (wid % groups) * PER_WAVE + k;
        if (blk >= blocks) break;
        const size_t i0 = blk * 64, i = i0 + lane;
        sm[wave][lane] = (i >= lead && i < n) ? bbd_digits(src_row[i]) : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32 o = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u32 pos = c0 + t, col = 8 * ((pos & 15) >> 1) + 2 * (pos >> 4) + (pos & 1);   // position 16 g + 2 t + h holds column 8 t + 2 g + h
            o |= (u32)src[col * 4 + v] << (8 * t);
        }
        *(u32 *)(dst_rows + (size_t)v * ldq + i0 + c0) = o;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

struct BbDotArgs {
    const fe *X;                // [na][72][ldx]
    size_t ldx, n;
    u32 na;                     // <= 16
    const unsigned char *YB;    // [8][108][ldq]
    size_t ldq;
    u32 nrows_y;                // 36 nb (<= 108)
    u32 nsteps, steps_per_chunk, chunks;
    int32_t *part;              // [unit 72][chunk][u 4][nt 7][64][4]
};
// (rows of X start on 8-byte boundaries only -- n = wit_len + l + 1 is even, not a multiple of four -- hence 8-byte loads)
__device__ __forceinline__ void bbd_load_x(const fe *xrow, size_t iw, u32 g, size_t n, bool xlive, u32 (&dst)[16]) {
    if (xlive && iw + 64 <= n) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int2 p = *(const int2 *)(xrow + iw + 8 * t + 2 * g);
            dst[2 * t] = (u32)p.x; dst[2 * t + 1] = (u32)p.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const size_t i = iw + 8 * (e >> 1) + 2 * g + (e & 1);
            dst[e] = (xlive && i < n) ? (u32)xrow[i] : 0;
        }
    }
}
__device__ __forceinline__ void bbd_load_y(const unsigned char *yb, size_t ldq, size_t i0, u32 row, u32 nrows_y, v4i (&b)[7]) {
#pragma unroll
    for (int nt = 0; nt < 7; nt++) {
        const u32 r = 16 * nt + row;
        b[nt] = r < nrows_y ? *(const v4i *)(yb + (size_t)r * ldq + i0) : v4i{0, 0, 0, 0};
    }
}
// wave = one (slot, cx) unit x one chunk of columns; the loads of the next K-step are in flight while this one is computed
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_bbdot_i8(BbDotArgs a) {
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, g = lane >> 4;
    const u32 unit = blockIdx.y, slot = unit / TAU, cx = unit % TAU;
    const u32 chunk = blockIdx.x * 4 + wave;
    v4i acc[4][7];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int nt = 0; nt < 7; nt++) acc[u][nt] = v4i{0, 0, 0, 0};
    if (chunk >= a.chunks) return;
    const u32 s0 = chunk * a.steps_per_chunk, s1 = s0 + a.steps_per_chunk < a.nsteps ? s0 + a.steps_per_chunk : a.nsteps;
    const fe *xrow = a.X + ((size_t)row * RE + TAU * slot + cx) * a.ldx;
    const bool xlive = row < a.na;
    const unsigned char *yb = a.YB + (size_t)slot * 108 * a.ldq;
    const u32 last = s1 > s0 ? s1 - 1 : s0;
    u32 xa[16], xn[16];
    v4i ba[7], bn[7];
    bbd_load_x(xrow, (size_t)s0 * 64, g, a.n, xlive, xa); bbd_load_y(yb, a.ldq, (size_t)s0 * 64 + 16 * g, row, a.nrows_y, ba);
    for (u32 st = s0; st < s1; st++) {
        const size_t iwn = (size_t)(st + 1 < last ? st + 1 : last) * 64;
        bbd_load_x(xrow, iwn, g, a.n, xlive, xn); bbd_load_y(yb, a.ldq, iwn + 16 * g, row, a.nrows_y, bn);
        u32 w[16];
#pragma unroll
        for (int t = 0; t < 16; t++) w[t] = bbd_digits((fe)xa[t]);
        constexpr u32 SEL_LO = 0x0C0C0400u, SEL_PAIR = 0x05040100u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            u32 op[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 sel = SEL_LO + u * 0x0101u;   // bytes u of both sources
                const u32 p01 = bbd_perm(w[4 * j + 1], w[4 * j], sel), p23 = bbd_perm(w[4 * j + 3], w[4 * j + 2], sel);
                op[j] = bbd_perm(p23, p01, SEL_PAIR);
            }
            const v4i av = v4i{(int)op[0], (int)op[1], (int)op[2], (int)op[3]};
#pragma unroll
            for (int nt = 0; nt < 7; nt++) acc[u][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, ba[nt], acc[u][nt], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) xa[t] = xn[t];
#pragma unroll
        for (int nt = 0; nt < 7; nt++) ba[nt] = bn[nt];
    }
    int32_t *o = a.part + ((size_t)unit * a.chunks + chunk) * (4 * 7 * 256);
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int nt = 0; nt < 7; nt++) *(v4i *)(o + ((size_t)u * 7 + nt) * 256 + lane * 4) = acc[u][nt];
}
__global__ void __launch_bounds__(256) k_bbdot_sum(const int32_t *part, u32 chunks, long long *tot) {
    const u32 e = blockIdx.x * 256 + threadIdx.x, unit = blockIdx.y;   // e < 4*7*256
    long long s = 0;
    for (u32 ch = 0; ch < chunks; ch++) s += part[((size_t)unit * chunks + ch) * 7168 + e];
    tot[(size_t)unit * 7168 + e] = s;
}
// block = output (a, b, slot, c), thread = (cx, digit u): out[(a*nb + b)*72 + 9*slot + c] = canonical of
//   R^-2 * sum over (cx, cy) with cx + cy = c (mod 9) of nu^[cx+cy >= 9] * sum_{u,v} 256^(u+v) tot[slot, cx][u][a][(b, cy, v)]
__global__ void __launch_bounds__(64) k_bbdot_finish(const long long *tot, u32 na, u32 nb, u32 nu_canon, u32 rinv2, u64 *out) {
    __shared__ u64 sm[36];
    const u32 o = blockIdx.x, t = threadIdx.x;
    const u32 c = o % TAU, slot = (o % RE) / TAU, b = (o / RE) % nb, av = o / (RE * nb);
    const u64 P = BB_P;
    if (t < 36) {
        const u32 cx = t >> 2, u = t & 3, cy = (c + TAU - cx) % TAU;
        const long long *tu = tot + (size_t)(slot * TAU + cx) * 7168;
        u64 val = 0, pw = 1;
        for (u32 v = 0; v < 4; v++) {
            const u32 col = (b * TAU + cy) * 4 + v, nt = col >> 4, cl = col & 15;
            const long long cell = tu[((size_t)u * 7 + nt) * 256 + (cl + 16 * (av >> 2)) * 4 + (av & 3)];
            const u64 cm = (u64)((cell % (long long)P + (long long)P) % (long long)P);
            val = (val + cm * pw) % P;
            pw = pw * 256 % P;
        }
        u64 pu = 1;
        for (u32 i = 0; i < u; i++) pu = pu * 256 % P;
        val = val * pu % P;
        if (cx + cy >= TAU) val = val * nu_canon % P;
        sm[t] = val;
    }
    __syncthreads();
    if (t == 0) {
        u64 res = 0;
        for (int i = 0; i < 36; i++) res = (res + sm[i]) % P;
        out[o] = res * rinv2 % P;
    }
}

size_t bbdot_i8_yb_bytes(size_t n) { return (size_t)8 * 108 * (bdiv(n, 64) * 64) + 64; }
static u32 bbdot_chunks(size_t nsteps) {
    size_t want = 12;   // 72 units x 12 chunks = 216 blocks of 4 waves: one batch on 256 CUs at one wave per SIMD
    if (want > nsteps) want = nsteps;
    const size_t spc = bdiv(nsteps, want);
    return (u32)bdiv(nsteps, spc);
}
// sized for the largest chunk count (bbdot_chunks is not monotone in the step count; the launcher may start a slice one column early)
size_t bbdot_i8_part_words(size_t) { return (size_t)72 * 12 * 7168; }
size_t bbdot_i8_tot_words() { return (size_t)72 * 7168; }
static u64 bb_powmod(u64 a, u64 e) {
    u64 r = 1;
    a %= BB_P;
    while (e) { if (e & 1) r = r * a % BB_P; a = a * a % BB_P; e >>= 1; }
    return r;
}
// X [na][72][ldx], Y [nb][72][ldy] (centred Montgomery words), n columns; out[(a*nb + b)*72 + 9*slot + c] canonical.  0, or -1 (shape).
// the Y digits alone, for launch_dot_batch_i8(.., y_packed = true) calls on X vectors of the same alignment (the eta products of the two sides of a fold step)
int launch_dot_pack_y(const fe *X, const fe *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, hipStream_t s) {
    if (nb < 1 || nb > 3 || n < 64 || (((size_t)X) & 3)) return -1;
    const size_t lead = (((size_t)X) & 7) / 4;
    Y -= lead; n += lead;
    const size_t ldq = bdiv(n, 64) * 64;
    hipLaunchKernelGGL(k_bbdot_pack_y, dim3((unsigned)bdiv((size_t)nb * RE * bdiv(ldq / 64, 8), 4)), dim3(256), 0, s, Y, ldy, nb, n, lead, ldq, YB);
    return 0;
}
int launch_dot_batch_i8(const DevBb &t, const fe *X, size_t ldx, u32 na, const fe *Y, size_t ldy, u32 nb, size_t n, unsigned char *YB, int32_t *part,
                        long long *tot, u64 *out, hipStream_t s, bool y_packed) {
    if (na < 1 || na > 16 || nb < 1 || nb > 3 || n < 64 || (ldx & 1) || (((size_t)X) & 3)) return -1;
    // a column slice that does not start on an 8-byte boundary: start one column earlier and give that column zero digits on the Y side
    const size_t lead = (((size_t)X) & 7) / 4;
    X -= lead; Y -= lead; n += lead;
    const size_t ldq = bdiv(n, 64) * 64;
    // exactness: a wave adds steps_per_chunk * 64 digit products of at most 2^14 into an int32 accumulator
    if (bdiv(ldq / 64, bbdot_chunks(ldq / 64)) >= 2048) return -1;
    if (!y_packed) hipLaunchKernelGGL(k_bbdot_pack_y, dim3((unsigned)bdiv((size_t)nb * RE * bdiv(ldq / 64, 8), 4)), dim3(256), 0, s, Y, ldy, nb, n, lead, ldq, YB);
    BbDotArgs a;
    a.X = X; a.ldx = ldx; a.n = n; a.na = na; a.YB = YB; a.ldq = ldq; a.nrows_y = 36 * nb;
    a.nsteps = (u32)(ldq / 64);
    a.chunks = bbdot_chunks(a.nsteps);
    a.steps_per_chunk = (u32)bdiv(a.nsteps, a.chunks);
    a.part = part;
    hipLaunchKernelGGL(k_bbdot_i8, dim3((unsigned)bdiv(a.chunks, 4), 72), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_bbdot_sum, dim3(28, 72), dim3(256), 0, s, part, a.chunks, tot);
    const u64 rinv = bb_powmod(BB_R, BB_P - 2);
    hipLaunchKernelGGL(k_bbdot_finish, dim3(na * nb * RE), dim3(64), 0, s, tot, na, nb, (u32)to_canon(t.nu), (u32)(rinv * rinv % BB_P), out);
    return 0;
}

// ---- T[k][c] = sum_i eq[i] * digit_k(planes[c][i]) (v_s of a decomposition, the linearization's v) on the matrix cores -------------------------------
// The BabyBear form of lf::k_coef_eval_i8 (binary digit planes only): per coefficient c an exact int8 GEMM with rows = the K <= 16 planes (digits -1, 0, 1 cut
// from the int32 plane words in registers), inner dimension = columns, matrix columns = the 36 balanced base-256 digits of the nine words of eq (packed once
// per call).  Was lfbb::k_coef_eval: K * 9 masked 64-bit additions per column and coefficient on the VALU (0.65 ms per call at C3).
__global__ void __launch_bounds__(256) k_bbce_pack_eq(const fe *eq, size_t ldeq, size_t n, size_t ldb, unsigned char *EB) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;     // four columns per thread
    const u32 q = blockIdx.y;
    if (i0 >= ldb) return;
    u32 d[4];
#pragma unroll
    for (int t = 0; t < 4; t++) d[t] = i0 + t < n ? bbd_digits(eq[(size_t)q * ldeq + i0 + t]) : 0;
#pragma unroll
    for (int u = 0; u < 4; u++)
        *(u32 *)(EB + (size_t)(4 * q + u) * ldb + i0) = ((d[0] >> (8 * u)) & 0xFF) | (((d[1] >> (8 * u)) & 0xFF) << 8) | (((d[2] >> (8 * u)) & 0xFF) << 16) | (((d[3] >> (8 * u)) & 0xFF) << 24);
}
struct BbCeArgs {
    const int32_t *planes;      // [72][ldp]
    size_t ldp, n;
    const unsigned char *EB;    // [36][ldb]
    size_t ldb;
    u32 rows;                   // K <= 16
    u32 steps_per_wg;
    int32_t *part;              // [wg][72][3][64][4]
};
__device__ __forceinline__ int bbce_digit(int32_t v, u32 k) {
    const u32 m = v < 0 ? 0u - (u32)v : (u32)v;
    const int d = (int)((m >> k) & 1);
    return v < 0 ? -d : d;
}
// grid (workgroups over the columns, 3 groups of 24 coefficients); wave w of a workgroup owns the coefficients 24 y + 6 w .. + 5
__global__ void __launch_bounds__(256) k_bbce_i8(BbCeArgs a) {
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 15, g = lane >> 4;
    const size_t nsteps = (a.n + 63) / 64;
    const size_t s0 = (size_t)blockIdx.x * a.steps_per_wg, s1 = s0 + a.steps_per_wg < nsteps ? s0 + a.steps_per_wg : nsteps;
    v4i acc[6][3];
#pragma unroll
    for (int mi = 0; mi < 6; mi++)
#pragma unroll
        for (int nt = 0; nt < 3; nt++) acc[mi][nt] = v4i{0, 0, 0, 0};
    const bool full16 = (a.ldp & 3) == 0 && (((size_t)a.planes) & 15) == 0;
    for (size_t st = s0; st < s1; st++) {
        const size_t j0 = st * 64 + 16 * g;           // this lane's 16 columns (EB is zero-padded to ldb, a multiple of 64)
        v4i b[3];
#pragma unroll
        for (int nt = 0; nt < 3; nt++) {
            const u32 r = 16 * nt + row;
            b[nt] = r < 36 ? *(const v4i *)(a.EB + (size_t)r * a.ldb + j0) : v4i{0, 0, 0, 0};
        }
#pragma unroll
        for (int mi = 0; mi < 6; mi++) {
            const u32 c = 24 * blockIdx.y + wave * 6 + mi;
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
            if (row < a.rows) {
#pragma unroll
                for (int q = 0; q < 16; q++) w4[q >> 2] |= (u32)(unsigned char)bbce_digit(v[q], row) << (8 * (q & 3));
            }
            const v4i av = v4i{(int)w4[0], (int)w4[1], (int)w4[2], (int)w4[3]};
#pragma unroll
            for (int nt = 0; nt < 3; nt++) acc[mi][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[nt], acc[mi][nt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mi = 0; mi < 6; mi++) {
        const u32 c = 24 * blockIdx.y + wave * 6 + mi;
#pragma unroll
        for (int nt = 0; nt < 3; nt++) *(v4i *)(a.part + ((((size_t)blockIdx.x * 72 + c) * 3 + nt) * 64 + lane) * 4) = acc[mi][nt];
    }
}
// block = 64 outputs x 4 groups of workgroups (a thread walks a quarter of the partial buffers; one thread per output: 61 us of serial adds at 256 workgroups)
__global__ void __launch_bounds__(256) k_bbce_sum(const int32_t *part, u32 nwg, long long *tot) {
    const size_t per = (size_t)72 * 3 * 256, i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const u32 g = threadIdx.x >> 6;
    __shared__ long long sm[4][64];
    long long s = 0;
    if (i < per)
        for (u32 w = g; w < nwg; w += 4) s += part[(size_t)w * per + i];
    sm[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && i < per) tot[i] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
// out[(k*72 + c)*9 + q] canonical: sum_u 256^u C[k][4q + u] is the Montgomery word of the evaluation (integer scaling keeps the Montgomery form)
__global__ void __launch_bounds__(256) k_bbce_finish(const long long *tot, u32 K, u32 rinv, u64 *out) {
    const u32 o = blockIdx.x * 256 + threadIdx.x;
    if (o >= K * 72 * 9) return;
    const u32 q = o % 9, c = (o / 9) % 72, k = o / (9 * 72);
    long long v = 0;
#pragma unroll
    for (u32 u = 0; u < 4; u++) {
        const u32 col = 4 * q + u, nt = col >> 4, cl = col & 15, ln = cl + 16 * (k >> 2), reg = k & 3;
        v += tot[(((size_t)c * 3 + nt) * 64 + ln) * 4 + reg] << (8 * u);      // |tot| <= n * 128: < 2^57 for n < 2^26
    }
    const long long Pm = (long long)BB_P;
    out[o] = (u64)((v % Pm + Pm) % Pm) * rinv % BB_P;      // Montgomery word -> canonical: times 2^-32
}
size_t coef_eval_i8_eb_bytes(size_t n) { return 36 * (bdiv(n, 64) * 64) + 64; }
size_t coef_eval_i8_part_words(u32 nwg) { return (size_t)nwg * 72 * 3 * 256; }
size_t coef_eval_i8_tot_words() { return (size_t)72 * 3 * 256; }
// planes [72][ldp] (n columns), eq [9][ldeq]; K <= 16 binary digit planes.  out[(k*72 + c)*9 + q] canonical (the layout of launch_coef_eval, mode_bits).
// Returns 0, or -1 if the shape is not handled (the caller keeps lfbb::k_coef_eval).
int launch_coef_eval_i8(const int32_t *planes, size_t ldp, size_t n, const fe *eq, size_t ldeq, u32 K, unsigned char *EB, u32 nwg, int32_t *part, long long *tot,
                        u64 *out, hipStream_t s) {
    if (!n || K < 1 || K > 16 || n >= ((size_t)1 << 26) || (((size_t)EB) & 15)) return -1;
    const size_t ldb = bdiv(n, 64) * 64, nsteps = ldb / 64;
    hipLaunchKernelGGL(k_bbce_pack_eq, dim3((unsigned)bdiv(ldb / 4, 256), 9), dim3(256), 0, s, eq, ldeq, n, ldb, EB);
    if (nwg > nsteps) nwg = (u32)nsteps;
    BbCeArgs a;
    a.planes = planes; a.ldp = ldp; a.n = n; a.EB = EB; a.ldb = ldb; a.rows = K; a.part = part;
    a.steps_per_wg = (u32)bdiv(nsteps, nwg);
    const u32 grid = (u32)bdiv(nsteps, a.steps_per_wg);
    hipLaunchKernelGGL(k_bbce_i8, dim3(grid, 3), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_bbce_sum, dim3((unsigned)bdiv((size_t)72 * 3 * 256, 64)), dim3(256), 0, s, part, grid, tot);
    hipLaunchKernelGGL(k_bbce_finish, dim3((unsigned)bdiv((size_t)K * 72 * 9, 256)), dim3(256), 0, s, tot, K, (u32)bb_powmod(BB_R, BB_P - 2), out);
    return 0;
}
}  // namespace lfbb

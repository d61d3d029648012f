// lf_ajtai_i8g.hip -- GENERAL Ajtai commitments on the int8 matrix cores (gfx950 v_mfma_i32_16x16x64_i8):
// AjtaiCommitmentScheme::commit_ntt (commitment/commitment_scheme.rs:37-54,75-77; benched as "CommitNTT", benches/ajtai.rs:15-31) and
// Witness::commit (arith.rs:357-362) -- y[i] = sum_j A[i][j] * f[j] for an ARBITRARY ring vector f, from the same resident byte planes of A
// that the digit-plane kernels of lf_ajtai_i8.hip stream (no NTT-form copy of A is needed any more).
//
// The contraction.  Centred coefficients of f (|v| <= (p-1)/2 < 2^63; a witness handle's int32 planes: |v| < 2^31) are cut into balanced
// base-128 digits, v = sum_k 128^k d_k with d_k in [-64, 63]: 10 planes for a general Goldilocks element, 5 for any int32 witness or a
// BabyBear element.  Per plane the product with A is the exact int8 GEMM of lf_ajtai_i8.hip: A = (NL kappa) x (RD N) bytes (biased by
// -128), B = Rot(f_k), never materialised: X^c_in * f_k is Toeplitz in d = c_out - c_in up to the wrap rules of X^RD = X^(RD/2) - 1,
//     H[d]  = f[d] + f[d + RD/2]                                   (outputs c_out >= RD/2; the two terms exist for 0 <= d < RD/2 only)
//     L'[d] = d >= 0 ? -f[d] : f[d + RD] + f[d + 3 RD/2]            (outputs c_out <  RD/2, NEGATED: y = -C there)
// -- an entry never has more than two terms (f[d] and f[d + RD] exclude each other), so with digits in [-64, 63] every entry fits an
// int8: H in [-128, 126], L' in [-128, 126] / [-63, 64].  (The un-negated L would need 128.)  The planes are recombined exactly at the
// end: y = sum_k 128^k y_k mod p on kappa x RD outputs (k_ajtai_i8g_finish).
//
// Kernel shape.  Digits arrive precomputed (k_i8g_cut_*: one pass over f, biased bytes, 8 columns per 64-bit word), so the producer
// waves only copy tiles of A and build the Toeplitz vectors; a workgroup holds ALL planes of the commitment (up to 16 column tiles) and
// HALF of the row tiles of A (<= 7 x 16 accumulator tiles over four multiplier waves): the two row halves read disjoint bytes of A, so
// -- unlike the plane groups of k_ajtai_i8s -- nothing is fetched twice and no pair of workgroups has to be kept in step.  int32
// accumulators are flushed to a fresh partial slot every `ft` tiles: |a b| <= 2^14 per MAC, 8 RD MACs per tile and output.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "lf_field.cuh"
#include "lf_ajtai_i8.h"

namespace lf {
namespace {
inline size_t cdiv(size_t a, size_t b) { return (a + b - 1) / b; }
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned long long ull;
// workgroup barrier that waits for LDS traffic only (see lf_ajtai_i8.hip)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr ull ZW = 0x4040404040404040ull;          // eight biased zero digits
constexpr ull NEGC = 0xC0C0C0C0C0C0C0C1ull;        // ~x + NEGC = 0xC0.. - x: bytes 128 - digit
}  // namespace

struct AjtaiI8GArgs {
    const unsigned char *Ab;   // packed row chunk of A (lf_ajtai_i8.hip: k_ajtai_pack_i8), MT row tiles
    u32 MT;
    const ull *pre;            // digit words [NP][RD][ldw]: byte q of word T = 64 + digit of column 8 T + q
    size_t ldw;
    u32 NP, ntiles, ft;        // planes, column tiles, accumulator flush period (tiles)
    u32 m_lo[2], mth[2];       // row halves: first row tile, row tiles
    u32 nch[2], tpw[2], nsub[2];   // column chunks (= workgroups) of a half, tiles per chunk, partial slots per chunk
    int32_t *part[2];          // [chunk][sub][mth][NT][64][4]
    int32_t *dsum;             // [nch[0]][NPMAX * RD]: digit sums of the first half's chunks
};

// PROF: per-phase shader-clock totals of every wave of workgroup 0 (LF_I8G_PROF=1; a measurement instantiation, tools/i8g_prof.py): [wave][phase], [wave][7] = tiles
__device__ unsigned long long g_i8g_prof[8][8];
__device__ unsigned int g_i8g_wg[512];    // PROF: loop duration of every workgroup in 100 MHz ticks (tools/i8g_prof.py prints them by half and XCD)
#define LF_G_STAMP(i_)                                                                   \
    if (PROF) {                                                                          \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
        pt[i_] += now_ - pc;                                                             \
        pc = now_;                                                                       \
    }
int ajtai_i8g_read_prof(unsigned long long *out64) { return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_i8g_prof), sizeof(g_i8g_prof)) == hipSuccess ? 0 : -1; }
int ajtai_i8g_read_wg(unsigned int *out512) { return hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_i8g_wg), sizeof(g_i8g_wg)) == hipSuccess ? 0 : -1; }
// geometry: ring degree RD, at most MTW row tiles per workgroup, NTW column tiles per multiplier wave
template <int RD, int MTW, int NTW>
struct GX {
    static constexpr int KS = RD / 8, VS = 2 * RD, HALF = RD / 2, DS = RD + 2, EPP = 4 * RD;   // a row of D: RD digit words, the zero word, the negation constant
    static constexpr int NT = 4 * NTW;                            // column tiles of a workgroup
    static constexpr int NPMAX = NT * 16 / RD;                    // planes they hold
    static constexpr int NCH = KS * MTW;                          // 1 KiB pieces of the workgroup's part of a tile of A (one global_load_lds_dwordx4 of a wave each)
    static constexpr int ALDS = NCH * 1024;
    static constexpr int NBA = RD <= 24 ? 4 : 3;                  // LDS ring of A tiles: tile T is multiplied while tiles T + 1 .. T + NBA - 1 land
    static constexpr int NVT = 192;                               // threads of the three vector waves (4 - 6); wave 7 only copies
    static constexpr int VB = NPMAX * 2 * VS, DB = NPMAX * DS;    // 64-bit words of a V / D buffer
    static constexpr int NDI = NPMAX * RD, NDR = (NDI + NVT - 1) / NVT;
    static constexpr int NVR = (VB + NVT - 1) / NVT;
    static_assert((NBA - 2) * NCH <= 63 && NPMAX * DS < 1024 && 2 * NPMAX * VS * 8 < 65536, "vmcnt range / packed offsets");
    static constexpr size_t lds_bytes() { return (size_t)NBA * ALDS + 2 * (size_t)VB * 8 + 16 + 2 * (size_t)DB * 8 + 16; }   // (+ the spare words)
    static_assert(lds_bytes() <= 160 * 1024, "LDS of a CU");
};

// ---- multiplier waves (0-3): wave ng owns column tiles [ng NTW, (ng + 1) NTW) and all mth row tiles of the workgroup's half
// MTH: the half's row tiles at compile time (loops, operand strides and the accumulator array are exact), 0 = any number <= MTW at run time (wave-uniform guards)
// NTU <= NTW: column tiles of this wave that hold planes of the commitment (10 planes x 24 = 15 tiles: the last wave multiplies 3 of its 4)
template <int RD, int MTW, int NTW, int MTH, bool PROF, int NTU = NTW>
__device__ __forceinline__ void i8g_mma(const AjtaiI8GArgs &a, unsigned char *smem, u32 ng, u32 mth_rt, u32 nsub, u32 T0, u32 T1, int32_t *part) {
    typedef GX<RD, MTW, NTW> G;
    constexpr int KS = G::KS, VS = G::VS, HALF = G::HALF, NT = G::NT;
    constexpr bool EXACT = MTH > 0;
    constexpr int ML = EXACT ? MTH : MTW;
    const u32 lane = threadIdx.x & 63;
    const u32 mth = EXACT ? (u32)MTH : mth_rt;
    const unsigned char *Al = smem;
    const ull *V = (const ull *)(smem + G::NBA * G::ALDS);
    u32 vb[NTU];
#pragma unroll
    for (int ni = 0; ni < NTU; ni++) {
        const u32 n = (ng * NTW + ni) * 16 + (lane & 15);
        u32 p = n / RD;
        const u32 co = n % RD;
        if (p >= (u32)G::NPMAX) p = G::NPMAX - 1;                 // padding columns behind the last plane: computed, never read
        vb[ni] = ((p * 2 + (co >= (u32)HALF ? 0u : 1u)) * VS + (RD - 1 - co + 2 * (lane >> 4))) * 8;
    }
    const u32 ab0 = lane * 16;
    const size_t per_slot = (size_t)mth * NT * 256;
    if (T0 < T1) lds_barrier();                                 // the producers' prologue has a barrier of its own: every wave must arrive
    lds_barrier();                                              // hand-over: A[T0], V[T0] are in buffer 0
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0, pr0 = 0;
    if (PROF) { pc = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
    for (u32 sub = 0; sub < nsub; sub++) {
        const u32 Tb = T0 + sub * a.ft, Te = Tb + a.ft < T1 ? Tb + a.ft : T1;
        v4i acc[ML][NTU];
#pragma unroll
        for (int mi = 0; mi < ML; mi++)
#pragma unroll
            for (int ni = 0; ni < NTU; ni++) acc[mi][ni] = v4i{0, 0, 0, 0};
        for (u32 T = Tb; T < Te; T++) {
            const u32 cur = (T - T0) & 1;
            const unsigned char *Ac = Al + ((T - T0) % G::NBA) * G::ALDS + ab0;
            const unsigned char *Vc = (const unsigned char *)(V + cur * G::VB);
            v4i b[NTU], bn[NTU];
#pragma unroll
            for (int ni = 0; ni < NTU; ni++) {
                const ull *q = (const ull *)(Vc + vb[ni]);
                const ull lo = q[0], hi = q[1];
                b[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
            }
            if (EXACT) {
                // the image of the half tile is dense ([KS][mth][64 lanes][16]): row tile mi of K-step s is piece q = s ML + mi at Ac + 1024 q.  The A operand rolls
                // through two registers ACROSS the K-steps: the read of piece q + 2 is issued before the MFMAs of piece q, so only the first two reads of a tile
                // are waited for (restarting the roll at every K-step exposed an LDS latency three times per tile)
                v4i avn = *(const v4i *)(Ac), avnn = *(const v4i *)(Ac + (KS * ML > 1 ? 1024 : 0));
#pragma unroll
                for (int q = 0; q < KS * ML; q++) {
                    const int s = q / ML, mi = q % ML;
                    const v4i av = avn;
                    avn = avnn;
                    if (q + 2 < KS * ML) avnn = *(const v4i *)(Ac + (q + 2) * 1024);
                    if (mi == (ML > 1 ? 1 : 0) && s + 1 < KS) {   // the next K-step's B operands, behind the first row tiles of this one
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
                    if (mi == ML - 1 && s + 1 < KS) {
#pragma unroll
                        for (int ni = 0; ni < NTU; ni++) b[ni] = bn[ni];
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const unsigned char *As = Ac + (size_t)s * mth * 1024;
                    // A operand: rolling registers, the read of row tile mi + 2 is issued before the MFMAs of row tile mi
                    v4i avn = *(const v4i *)(As), avnn = *(const v4i *)(As + (mth > 1 ? 1024 : 0));
#pragma unroll
                    for (int mi = 0; mi < ML; mi++) {
                        const v4i av = avn;
                        avn = avnn;
                        if (mi + 2 < ML) avnn = *(const v4i *)(As + (EXACT || (u32)(mi + 2) < mth ? (mi + 2) * 1024 : 0));
                        if (mi == (ML > 1 ? 1 : 0) && s + 1 < KS) {   // the next K-step's B operands, behind the first row tiles of this one
#pragma unroll
                            for (int ni = 0; ni < NTU; ni++) {
                                const ull *q = (const ull *)(Vc + vb[ni] + (s + 1) * 64);
                                const ull lo = q[0], hi = q[1];
                                bn[ni] = v4i{(int)(u32)lo, (int)(u32)(lo >> 32), (int)(u32)hi, (int)(u32)(hi >> 32)};
                            }
                        }
                        if (EXACT || (u32)mi < mth) {                  // (wave-uniform)
#pragma unroll
                            for (int ni = 0; ni < NTU; ni++) acc[mi][ni] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[ni], acc[mi][ni], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (s + 1 < KS) {
#pragma unroll
                        for (int ni = 0; ni < NTU; ni++) b[ni] = bn[ni];
                    }
                }
            }
            LF_G_STAMP(3);     // K-steps
            lds_barrier();
            LF_G_STAMP(6);     // barrier
        }
        int32_t *dst = part + sub * per_slot;
#pragma unroll
        for (int mi = 0; mi < ML; mi++)
#pragma unroll
            for (int ni = 0; ni < NTU; ni++)
                if (EXACT || (u32)mi < mth) *(v4i *)(dst + (((size_t)mi * NT + ng * NTW + ni) * 64 + lane) * 4) = acc[mi][ni];
    }
    for (u32 pad = (T1 - T0) & 3; pad & 3; pad++) lds_barrier();   // the producers' loop runs whole trips of four tiles
    if (PROF && lane == 0 && ng == 0) {                          // slot [4] of waves 1 .. 3: the loop in 100 MHz ticks -- workgroup 0, the slowest workgroup, the sum over all
        const unsigned long long dr = __builtin_amdgcn_s_memrealtime() - pr0;
        if (blockIdx.x == 0) g_i8g_prof[1][4] = dr;
        if (blockIdx.x < 512) g_i8g_wg[blockIdx.x] = (unsigned int)dr;
        atomicMax(&g_i8g_prof[2][4], dr);
        atomicAdd(&g_i8g_prof[3][4], dr);
    }
    if (PROF && blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 7; i++) if (i != 4) g_i8g_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8g_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
}

// ---- copy wave (7): the workgroup's part of every tile of A straight from HBM into the LDS ring (global_load_lds_dwordx4: 1 KiB per instruction, no staging
// registers, no ds_write pass).  The wave touches LDS through nothing else: the compiler's wait-count pass makes any ds_read of a wave wait for that wave's
// LDS-DMA in flight (it cannot tell the buffers apart), which is why the vector waves below do not copy.  Counted waits: tile T + 1 has landed when at most
// (NBA - 2) tiles' pieces are outstanding.
typedef __attribute__((address_space(1))) const void *i8g_gptr;
typedef __attribute__((address_space(3))) void *i8g_lptr;
template <int RD, int MTW, int NTW, int MTH, bool PROF>
__device__ __forceinline__ void i8g_copy(const AjtaiI8GArgs &a, unsigned char *smem, u32 m_lo, u32 mth_rt, u32 T0, u32 T1) {
    typedef GX<RD, MTW, NTW> G;
    constexpr int KS = G::KS, NBA = G::NBA;
    constexpr int NCH = KS * (MTH > 0 ? MTH : MTW);             // pieces issued per tile (the counted waits need a compile-time number)
    const u32 mth = MTH > 0 ? (u32)MTH : mth_rt;
    const u32 lane = threadIdx.x & 63;
    const size_t a_tile = (size_t)KS * a.MT * 1024;
    const u32 Tlast = a.ntiles - 1;
    const u32 nch = KS * mth;                                   // pieces of this half (MTH = 0: a half with fewer row tiles than MTW re-copies piece 0 in its spare slots: same bytes, same place)
    // piece c = (K-step c / mth, row tile c % mth): LDS offset 1024 c (the image the multipliers read: [KS][mth][64 lanes][16]), global offset inside a tile:
    auto issue = [&](u32 T, u32 buf) {
        const unsigned char *src = a.Ab + (size_t)(T < Tlast ? T : Tlast) * a_tile + (size_t)m_lo * 1024 + lane * 16;
        unsigned char *dst = smem + (size_t)buf * G::ALDS;
        u32 s = 0, r = 0;                                       // (wave-uniform counters: no division per piece)
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const bool in = MTH > 0 || (u32)c < nch;
            const u32 go = in ? (s * a.MT + r) * 1024 : 0, lo = in ? (u32)c * 1024 : 0;
            __builtin_amdgcn_global_load_lds((i8g_gptr)(src + go), (i8g_lptr)(dst + lo), 16, 0, 0);
            if (++r == mth) { r = 0; s++; }
        }
    };
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0;
    if (T0 < T1) {
#pragma unroll
        for (int q = 0; q + 1 < NBA; q++) issue(T0 + q, q);     // tiles T0 .. T0 + NBA - 2 -> buffers 0 .. NBA - 2
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBA - 2) * NCH) : "memory");   // tile T0 has landed
        lds_barrier();                                          // (the vector waves' prologue barrier)
    }
    lds_barrier();                                              // hand-over of buffer 0
    if (PROF) pc = __builtin_amdgcn_s_memtime();
    u32 buf = NBA - 1;                                          // buffer of tile T + NBA - 1
    const u32 Tend = T0 + ((T1 - T0 + 3) & ~3u);                // whole trips of four tiles, like the vector waves
    for (u32 T = T0; T < Tend; T++) {
        issue(T + NBA - 1, buf);                                // into the buffer the multipliers left at the last barrier (tile T - 1)
        buf = buf + 1 == (u32)NBA ? 0 : buf + 1;
        LF_G_STAMP(1);     /* issue */
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBA - 2) * NCH) : "memory");   // tile T + 1 has landed
        LF_G_STAMP(5);     /* wait for the tile */
        lds_barrier();
        LF_G_STAMP(6);     /* barrier */
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // nothing may land in LDS after the workgroup has gone
    if (PROF && blockIdx.x == 0 && lane == 0) {
        for (int i = 0; i < 7; i++) g_i8g_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8g_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
}

// ---- vector waves (4-6): btid = 0 .. 191 -- digit words of the tile two ahead into D, Toeplitz vectors of the next tile into V
template <int RD, int MTW, int NTW, bool PROF>
__device__ __forceinline__ void i8g_build(const AjtaiI8GArgs &a, unsigned char *smem, u32 T0, u32 T1, int32_t *dsum_slot) {
    typedef GX<RD, MTW, NTW> G;
    constexpr int VS = G::VS, HALF = G::HALF, DS = G::DS, EPP = G::EPP, NDR = G::NDR, NVR = G::NVR, NDI = G::NDI, NVT = G::NVT;
    const u32 btid = threadIdx.x - 256;
    ull *V = (ull *)(smem + G::NBA * G::ALDS);
    ull *Dl = V + 2 * G::VB + 2;                               // (two pad words behind the V buffers)
    // digit words: item = btid + NVT r < NDI is (plane, coefficient) = (item / RD, item % RD); planes >= NP hold zero digits
    static_assert(NDR <= 2, "two digit words per thread");
    // raw words (the "no digit" select happens at use time, so that nothing waits for the loads where they are issued), FOUR named sets, one per unrolled
    // iteration: every set has exactly one load per trip of the loop (with two sets, each loaded twice per trip, the register allocator gave the two loads
    // different registers and joined them with a copy at the loop's back edge -- a copy that waits for the load).  The words of tile T + 4 are requested in
    // iteration T and used in iteration T + 2.
    ull dA0 = ZW, dA1 = ZW, dB0 = ZW, dB1 = ZW, dC0 = ZW, dC1 = ZW, dD0 = ZW, dD1 = ZW;
    bool kA0 = false, kA1 = false, kB0 = false, kB1 = false, kC0 = false, kC1 = false, kD0 = false, kD1 = false;
    int dsum0 = 0, dsum1 = 0;   // digit sums of this thread's (plane, coefficient) items over the workgroup's columns
    const u32 it0 = btid, it1 = btid + NVT;
    const bool ok0 = it0 < (u32)NDI && it0 / RD < a.NP, ok1 = NDR > 1 && it1 < (u32)NDI && it1 / RD < a.NP;
    // LDS word of an item inside a D buffer; threads without an item write the spare word behind the two buffers (no control flow around the stores)
    const bool has0 = it0 < (u32)NDI, has1 = NDR > 1 && it1 < (u32)NDI;
    const u32 dso0 = has0 ? (it0 / RD) * DS + it0 % RD : 0, dso1 = has1 ? (it1 / RD) * DS + it1 % RD : 0;
    const ull *row0 = a.pre + (size_t)(ok0 ? it0 : 0) * a.ldw, *row1 = a.pre + (size_t)(ok1 ? it1 : 0) * a.ldw;
    auto load_d = [&](u32 T, ull &r0, ull &r1, bool &k0, bool &k1) {     // (clamped address, no control flow)
        const bool in = T < T1;
        const u32 Tc = in ? T : T0;
        r0 = row0[Tc];
        k0 = in && ok0;
        if (NDR > 1) { r1 = row1[Tc]; k1 = in && ok1; }
    };
    auto bsum = [](ull w) { return (int)__builtin_amdgcn_sad_u8((u32)w, 0u, __builtin_amdgcn_sad_u8((u32)(w >> 32), 0u, 0u)) - 8 * 64; };
    auto gen_d = [&](u32 dbuf, ull r0, ull r1, bool k0, bool k1) {
        { const ull w = k0 ? r0 : ZW; Dl[has0 ? dbuf * G::DB + dso0 : 2 * G::DB] = w; dsum0 += bsum(w); }
        if (NDR > 1) { const ull w = k1 ? r1 : ZW; Dl[has1 ? dbuf * G::DB + dso1 : 2 * G::DB] = w; dsum1 += bsum(w); }
    };
#define LF_G_LOADD(S_, T_) load_d((T_), d##S_##0, d##S_##1, k##S_##0, k##S_##1)
#define LF_G_GEND(S_, buf_) gen_d((buf_), d##S_##0, d##S_##1, k##S_##0, k##S_##1)
    // vectors: entry idx = btid + NVT r < VB = (plane, H / L', e) = (idx / EPP, (idx % EPP) / VS, idx % VS), d = RD - 1 - e;
    //   H : v = D[oa] + D[ob]            (f[d], f[d + RD/2]; a missing term reads the biased zero word)
    //   L': v = ~D[oa] + NEGC            (d >= 0: 0xC0.. - D[oa] = -f[d])      or      D[oa] + D[ob]     (d < 0: f[d + RD], f[d + 3 RD/2])
    // bytes are 64 + digit: a sum is 128 + value, 0xC0 - byte is 128 - digit; no byte carries or borrows; the XOR with 0x80 gives the int8.
    // Branch-free: v = (D[oa] ^ mask) + D[ob] with ob -> the constant word NEGC of the row for a negated entry.  (The first version selected between the two
    // forms with a divergent branch per entry: reads, branch, write, four times in series -- 1 250 cycles per tile, the producers bound the loop.)
    u32 vo[NVR];          // oa | ob << 10 | neg << 20, offsets in words from the D buffer
#pragma unroll
    for (int r = 0; r < NVR; r++) {
        const u32 idx = btid + NVT * r, idc = idx < (u32)G::VB ? idx : 0, vp = idc / EPP, vr = idc % EPP;
        u32 oa = RD, ob = RD, neg = 0;
        const int dl = RD - 1 - (int)(vr % VS);
        if (vr < (u32)VS) {
            if (dl >= -(HALF - 1) && dl <= RD - 1) { if (dl >= 0) oa = dl; if (dl <= HALF - 1) ob = dl + HALF; }
        } else if (dl >= -(RD - 1) && dl <= HALF - 1) {
            if (dl >= 0) { oa = dl; ob = RD + 1; neg = 1; }
            else { oa = dl + RD; if (dl <= -(HALF + 1)) ob = dl + RD + HALF; }
        }
        vo[r] = (vp * DS + oa) | ((vp * DS + ob) << 10) | (neg << 20);
    }
    auto gen_v = [&](u32 dbuf, u32 vbuf) {
        const ull *D = Dl + dbuf * G::DB;
        ull xa[NVR], xb[NVR];
#pragma unroll
        for (int r = 0; r < NVR; r++) { xa[r] = D[vo[r] & 0x3FF]; xb[r] = D[(vo[r] >> 10) & 0x3FF]; }
#pragma unroll
        for (int r = 0; r < NVR; r++) {
            const ull mk = 0ull - (ull)(vo[r] >> 20);
            const ull v = ((xa[r] ^ mk) + xb[r]) ^ 0x8080808080808080ull;
            // (the last round of a buffer that is not a multiple of NVT words: the spare threads write the pad word behind the V buffers)
            V[(G::VB % NVT) == 0 || btid + NVT * r < (u32)G::VB ? vbuf * G::VB + btid + NVT * r : 2 * G::VB] = v;
        }
    };
    if (btid < 2 * G::NPMAX) { Dl[btid * DS + RD] = ZW; Dl[btid * DS + RD + 1] = NEGC; }   // the biased zero word and the negation constant of every row of both D buffers
    if (T0 < T1) {
        // prologue: D[T0], D[T0+1]; digit words of T0+2, T0+3 in flight (sets C, D); V[T0]
        LF_G_LOADD(A, T0);
        LF_G_LOADD(B, T0 + 1);
        LF_G_GEND(A, 0);
        LF_G_GEND(B, 1);
        LF_G_LOADD(C, T0 + 2);
        LF_G_LOADD(D, T0 + 3);
        lds_barrier();
        gen_v(0, 0);
        // Every load of the prologue lands before the loop starts: the compiler's wait-count pass merges the pending-load state of the loop's back edge with the
        // state at its entry and waits for the YOUNGER of the two at every use; with nothing pending at the entry only the steady-state distances remain.
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0), expcnt / lgkmcnt untouched
    }
    lds_barrier();                                              // hand-over of buffer 0 (matches the multipliers' first barrier)
    // iteration of tile Tt (buffer parity PB_ = (Tt - T0) & 1; digit-word sets DU_ used / DL_ loaded): digits of tile Tt + 2 -> D[PB_] (it held tile Tt), request
    // the digit words of tile Tt + 4, vectors of tile Tt + 1 from D[1 - PB_]
#define LF_G_ITER(DU_, DL_, Tt_, PB_)                                                                              \
    do {                                                                                                           \
        LF_G_GEND(DU_, PB_);                                                                                       \
        LF_G_STAMP(0);     /* digits (waits for the words requested two tiles ago) */                              \
        LF_G_LOADD(DL_, (Tt_) + 4);                                                                                \
        gen_v(1 - (PB_), 1 - (PB_));                                                                               \
        LF_G_STAMP(2);     /* vectors */                                                                           \
        lds_barrier();                                                                                             \
        LF_G_STAMP(6);     /* barrier */                                                                           \
    } while (0)
    unsigned long long pt[7] = {0, 0, 0, 0, 0, 0, 0}, pc = 0;
    if (PROF) pc = __builtin_amdgcn_s_memtime();
    for (u32 T = T0; T < T1; T += 4) {
        // NO early exit inside a trip: the structurizer funnels a `break` through the loop's latch, and the wait-count pass then merges "left after the
        // second iteration" into the state of the back edge.  A chunk runs a multiple of four iterations; the spare ones carry zero digits, and the
        // multiplier and copy waves take part in their barriers.
        LF_G_ITER(C, A, T, 0);
        LF_G_ITER(D, B, T + 1, 1);
        LF_G_ITER(A, C, T + 2, 0);
        LF_G_ITER(B, D, T + 3, 1);
    }
#undef LF_G_ITER
#undef LF_G_LOADD
#undef LF_G_GEND
    if (PROF && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        for (int i = 0; i < 7; i++) g_i8g_prof[threadIdx.x >> 6][i] = pt[i];
        g_i8g_prof[threadIdx.x >> 6][7] = T1 - T0;
    }
    if (dsum_slot) {
        if (has0) dsum_slot[it0] = dsum0;
        if (has1) dsum_slot[it1] = dsum1;
    }
}

template <int RD, int MTW, int NTW, bool PROF = false>
__global__ void __launch_bounds__(512) k_ajtai_i8g(AjtaiI8GArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef GX<RD, MTW, NTW> G;
    const u32 h = blockIdx.x >= a.nch[0] ? 1 : 0, chunk = blockIdx.x - (h ? a.nch[0] : 0);
    const u32 mth = a.mth[h], nsub = a.nsub[h];
    const u32 T0 = chunk * a.tpw[h], T1 = T0 + a.tpw[h] < a.ntiles ? T0 + a.tpw[h] : a.ntiles;
    const u32 wave = threadIdx.x >> 6;
    int32_t *part = a.part[h] + (size_t)chunk * nsub * mth * G::NT * 256;
    if (wave >= 4 && wave < 7) { i8g_build<RD, MTW, NTW, PROF>(a, smem, T0, T1, h == 0 ? a.dsum + (size_t)chunk * G::NDI : nullptr); return; }
    const u32 nt_used = (a.NP * RD + 15) / 16;                  // column tiles that hold planes (the rest of the workgroup's NT tiles is padding nobody reads)
    // the half's row-tile count at compile time where it is one of the usual ones (7 + 6 at kappa 26, 5 + 5 at kappa 20, 4 + 4 at kappa 16): exact loops, no spare pieces
#define LF_G_ROLE(MTH_)                                                                                             \
    do {                                                                                                           \
        if (wave == 7) i8g_copy<RD, MTW, NTW, MTH_, PROF>(a, smem, a.m_lo[h], mth, T0, T1);                        \
        else if (NTW > 1 && wave == 3 && nt_used == 4 * NTW - 1) i8g_mma<RD, MTW, NTW, MTH_, PROF, (NTW > 1 ? NTW - 1 : 1)>(a, smem, wave, mth, nsub, T0, T1, part);   \
        else i8g_mma<RD, MTW, NTW, MTH_, PROF>(a, smem, wave, mth, nsub, T0, T1, part);                            \
    } while (0)
    if (mth == (u32)MTW) LF_G_ROLE(MTW);
    else if (MTW > 1 && mth == (u32)(MTW - 1)) LF_G_ROLE((MTW > 1 ? MTW - 1 : 1));
    else if (MTW > 2 && mth == (u32)(MTW - 2)) LF_G_ROLE((MTW > 2 ? MTW - 2 : 1));
    else if (MTW > 3 && mth == (u32)(MTW - 3)) LF_G_ROLE((MTW > 3 ? MTW - 3 : 1));
    else LF_G_ROLE(0);
#undef LF_G_ROLE
}

// sum layout: [S0: mth0 NT 256][S1: mth1 NT 256][F: NDI] -- the partial tiles of each row half over its slots, the digit sums over the first half's chunks
__global__ void __launch_bounds__(256) k_ajtai_i8g_sum(const int32_t *p0, size_t per0, u32 n0, const int32_t *p1, size_t per1, u32 n1, const int32_t *dsum, u32 ndi,
                                                       u32 nd, long long *sum) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= per0 + per1 + ndi) return;
    const int32_t *src;
    size_t stride;
    u32 cnt;
    if (e < per0) { src = p0 + e; stride = per0; cnt = n0; }
    else if (e < per0 + per1) { src = p1 + (e - per0); stride = per1; cnt = n1; }
    else { src = dsum + (e - per0 - per1); stride = ndi; cnt = nd; }
    long long s = 0;
    u32 w = 0;
    for (; w + 8 <= cnt; w += 8) {
        int32_t x[8];
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = src[(size_t)(w + q) * stride];
#pragma unroll
        for (int q = 0; q < 8; q++) s += x[q];
    }
    for (; w < cnt; w++) s += src[(size_t)w * stride];
    sum[e] = s;
}
namespace {
__device__ __forceinline__ u64 g_s128_mod_small(__int128 v, u64 p) {
    const bool neg = v < 0;
    unsigned __int128 t = neg ? (unsigned __int128)(-v) : (unsigned __int128)v;
    const u64 lo = (u64)t, hi = (u64)(t >> 64);
    const u64 two64 = ((0xFFFFFFFFFFFFFFFFull % p) + 1) % p;
    u64 r = ((hi % p) * two64 + lo % p) % p;
    return neg ? (p - r) % p : r;
}
}  // namespace
// y[row][c_out] = sum_k 128^k * (+-)(C_k + 128 * colsum_k) mod p, canonical, coefficient form.  Element e = row0 + i of kappa_total:
// soa != 0: out[c_out * kappa_total + e], else out[e * RD + c_out].  p_small = 0: the Goldilocks modulus.
__global__ void __launch_bounds__(256) k_ajtai_i8g_finish(const long long *sum, u32 mth0, u32 mth1, u32 NT, u32 ndi, u32 NP, u32 kappa, u32 row0, u32 kappa_total, u32 RD,
                                                          u32 NL, u64 p_small, int soa, u64 *out) {
    const u32 o = blockIdx.x * 256 + threadIdx.x;
    if (o >= kappa * RD) return;
    const u32 co = o % RD, i = o / RD, HALF = RD / 2;
    const long long *S1 = sum + (size_t)mth0 * NT * 256, *Fs = S1 + (size_t)mth1 * NT * 256;
    (void)ndi;
    u64 val = 0, pw = 1;
    for (u32 p = 0; p < NP; p++) {
        const long long *F = Fs + (size_t)p * RD;
        // colsum of the operand that was multiplied (H, or the negated L'): the "-128" bias of the bytes of A
        long long Tsum = 0;
        for (int ci = 0; ci < (int)RD; ci++) {
            const int d = (int)co - ci;
            if (co >= HALF) {
                if (d >= 0) Tsum += F[d];
                if (d <= (int)HALF - 1 && d >= -((int)HALF - 1)) Tsum += F[d + HALF];
            } else {
                if (d >= 0) Tsum -= F[d];
                if (d <= -1) Tsum += F[d + RD];
                if (d <= -(int)HALF - 1) Tsum += F[d + RD + HALF];
            }
        }
        const u32 n = p * RD + co, nt = n >> 4, col = n & 15;
        __int128 tot = 0;
        for (u32 u = 0; u < NL; u++) {
            const u32 m = NL * i + u, mt = m >> 4, r = m & 15, ln = col + 16 * (r >> 2), reg = r & 3;
            const long long c = mt < mth0 ? sum[(((size_t)mt * NT + nt) * 64 + ln) * 4 + reg] : S1[(((size_t)(mt - mth0) * NT + nt) * 64 + ln) * 4 + reg];
            tot += (__int128)(c + 128 * Tsum) << (8 * u);
        }
        if (co < HALF) tot = -tot;
        if (p_small) {
            val = (val + (u64)(((unsigned __int128)g_s128_mod_small(tot, p_small) * pw) % p_small)) % p_small;
            pw = (pw * 128) % p_small;
        } else {
            val = fq_add(val, fq_mul(fq_from_s128((u64)tot, (int64_t)(tot >> 64)), pw));
            pw = fq_mul(pw, 128);
        }
    }
    const size_t e = (size_t)row0 + i;
    if (soa) out[(size_t)co * kappa_total + e] = val;
    else out[e * RD + co] = val;
}

// ---- the digit pass of centred int32 coefficients (a witness handle's planes): 8 columns of one coefficient per thread -> NP words.  coef (c, j) at
// src[c * ld + j]; columns past n hold zero digits.  (SRC 0, canonical u64 coefficients, is the same pass for callers that hold them.)
template <int SRC>
__global__ void __launch_bounds__(256) k_i8g_cut(const void *src, size_t ld, size_t n, u64 p_small, u32 RD, u32 NP, size_t ntiles, ull *pre, size_t ldw) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= ntiles * RD) return;
    const size_t T = gid % ntiles;
    const u32 c = (u32)(gid / ntiles);
    long long x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const size_t j = T * 8 + q;
        if (SRC == 0) {
            const u64 v = j < n ? ((const u64 *)src)[(size_t)c * ld + j] : 0;
            if (p_small) x[q] = v > (p_small - 1) / 2 ? (long long)v - (long long)p_small : (long long)v;
            else x[q] = v > (LF_P - 1) / 2 ? (long long)(v - LF_P) : (long long)v;      // (v - p wraps to the negative two's-complement value)
        } else x[q] = j < n ? (long long)((const int32_t *)src)[(size_t)c * ld + j] : 0;
    }
    for (u32 k = 0; k < NP; k++) {
        ull w = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const long long t = x[q] + 64;
            w |= (ull)(t & 127) << (8 * q);
            x[q] = t >> 7;
        }
        pre[((size_t)k * RD + c) * ldw + T] = w;
    }
}
// The Goldilocks digit pass straight from the NTT form: f [24][ld] -> coefficients (the inverse CRT map through LDS, as k_ajtai_icrt_pack_i8) -> digit
// words.  Block = 32 columns (4 tiles).  The inverse map is data (lf_set_ring_tables): `mat` is its dense 24 x 24 matrix; when every row has at most 8 non-zero
// entries -- the shipped tables: one per slot -- the host also passes the compressed rows sp_val / sp_col [24][8] (column 0xFFFFFFFF = no entry) and an
// output costs 8 lazy 64 x 64 products instead of 24 (the dense pass was 0.5 ms of a 2.0 ms commitment at 2^20 columns: integer-multiplier-bound).
__global__ void __launch_bounds__(256) k_i8g_cut_ntt(const u64 *mat, const u64 *sp_val, const u32 *sp_col, const u64 *ntt, size_t ld, size_t n, u32 NP, size_t ntiles,
                                                     ull *pre, size_t ldw) {
    __shared__ u64 M[24 * 24], X[25][32], Cf[24][33];
    __shared__ u32 MC[24 * 8];
    if (sp_val) {
        if (threadIdx.x < 192) { M[threadIdx.x] = sp_val[threadIdx.x]; MC[threadIdx.x] = sp_col[threadIdx.x] < 24 ? sp_col[threadIdx.x] : 24; }
        if (threadIdx.x < 32) X[24][threadIdx.x] = 0;           // (row 24: the operand of a missing entry)
    } else
        for (int t = threadIdx.x; t < 576; t += 256) M[t] = mat[t];
    const size_t j0 = (size_t)blockIdx.x * 32;
    for (int t = threadIdx.x; t < 768; t += 256) {
        const u32 c = t >> 5, jj = t & 31;
        X[c][jj] = j0 + jj < n ? ntt[(size_t)c * ld + j0 + jj] : 0;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 768; t += 256) {
        const u32 r = t >> 5, jj = t & 31;
        Acc a;
        if (sp_val) {
            acc_set(a, M[r * 8], X[MC[r * 8]][jj]);
#pragma unroll
            for (int q = 1; q < 8; q++) acc_mad(a, M[r * 8 + q], X[MC[r * 8 + q]][jj]);
        } else {
            acc_set(a, M[r * 24], X[0][jj]);
#pragma unroll
            for (int c = 1; c < 24; c++) acc_mad(a, M[r * 24 + c], X[c][jj]);
        }
        Cf[r][jj] = acc_reduce(a);
    }
    __syncthreads();
    if (threadIdx.x >= 96) return;
    const u32 c = threadIdx.x >> 2, tl = threadIdx.x & 3;
    const size_t T = (size_t)blockIdx.x * 4 + tl;
    if (T >= ntiles) return;
    long long x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const u64 v = Cf[c][tl * 8 + q];                         // columns past n hold zeros
        x[q] = v > (LF_P - 1) / 2 ? (long long)(v - LF_P) : (long long)v;
    }
    for (u32 k = 0; k < NP; k++) {
        ull w = 0;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const long long t = x[q] + 64;
            w |= (ull)(t & 127) << (8 * q);
            x[q] = t >> 7;
        }
        pre[((size_t)k * 24 + c) * ldw + T] = w;
    }
}
void launch_i8g_cut_ntt(const u64 *icrt_mat, const u64 *sp_val, const u32 *sp_col, const u64 *ntt, size_t ld, size_t n, u32 NP, unsigned long long *pre, size_t ldw,
                        hipStream_t s) {
    const size_t ntiles = (n + 7) / 8;
    hipLaunchKernelGGL(k_i8g_cut_ntt, dim3((unsigned)cdiv(ntiles, 4)), dim3(256), 0, s, icrt_mat, sp_val, sp_col, ntt, ld, n, NP, ntiles, pre, ldw);
}
void launch_i8g_cut_i32(const int32_t *planes, size_t ld, size_t n, u32 RD, u32 NP, unsigned long long *pre, size_t ldw, hipStream_t s) {
    const size_t ntiles = (n + 7) / 8;
    hipLaunchKernelGGL(k_i8g_cut<1>, dim3((unsigned)cdiv(ntiles * RD, 256)), dim3(256), 0, s, (const void *)planes, ld, n, (u64)0, RD, NP, ntiles, pre, ldw);
}

// planes of a general element / of centred coefficients bounded by 2^31
u32 ajtai_i8g_planes_general(const AjtaiI8Ring &R) { return R.RD == 24 ? 10 : 5; }
u32 ajtai_i8g_planes_i32() { return 5; }

namespace {
struct GPlan {
    u32 halves, m_lo[2], mth[2], nch[2], tpw[2], nsub[2], ft, NT, ndi, mtw;
    size_t per[2], part_words, dsum_words, sum_words;
};
// instantiated shapes: 24-ring: <= 7 row tiles per half, 16 (10 planes) or 8 (5 planes) column tiles; 72-ring: <= 4 row tiles, 24 column tiles (5 planes)
bool g_plan(const AjtaiI8Ring &R, u32 MT, size_t n, u32 NP, u32 nwg, GPlan *g) {
    const u32 ntiles = (u32)((n + 7) / 8);
    if (!ntiles || !NP || !MT || !nwg) return false;
    if (R.RD == 24) {
        if (MT > 14 || NP > 10) return false;
        g->mtw = 7;
        g->NT = NP <= 5 ? 8 : 16;
    } else if (R.RD == 72) {
        if (MT > 8 || NP > 5) return false;
        g->mtw = 4;
        g->NT = 24;
    } else return false;
    g->ndi = g->NT * 16 / R.RD * R.RD;
    g->halves = MT > g->mtw ? 2 : 1;
    if (g->halves == 2 && nwg < 2) nwg = 2;
    if (g->halves == 2) { g->mth[0] = (MT + 1) / 2; g->mth[1] = MT - g->mth[0]; }
    else { g->mth[0] = MT; g->mth[1] = 0; }
    g->m_lo[0] = 0; g->m_lo[1] = g->mth[0];
    // workgroups in proportion to the halves' cost per tile: the multiplier waves' MFMAs and the copy wave's bytes scale with the row tiles, the barrier and the
    // B operands do not -- measured per tile at kappa 26 (7 + 6 row tiles): 1.24 / 1.15 us with 10 planes, 1.02 / 0.92 with 5 (tools/i8g_prof.py): ~ mth + 5
    u32 w0 = g->halves == 2 ? (nwg * (g->mth[0] + 5) + (MT + 10) / 2) / (MT + 10) : nwg;
    if (g->halves == 2) { if (w0 < 1) w0 = 1; if (w0 > nwg - 1) w0 = nwg - 1; }
    const u32 w[2] = {w0, g->halves == 2 ? nwg - w0 : 0};
    g->ft = (u32)(0x7FFFFFFFull / (16384ull * R.RD * 8));
    g->part_words = 0;
    for (int h = 0; h < 2; h++) {
        if (!w[h]) { g->nch[h] = 0; g->tpw[h] = 1; g->nsub[h] = 1; g->per[h] = 0; continue; }
        g->tpw[h] = (ntiles + w[h] - 1) / w[h];
        g->nch[h] = (ntiles + g->tpw[h] - 1) / g->tpw[h];
        g->nsub[h] = (g->tpw[h] + g->ft - 1) / g->ft;
        g->per[h] = (size_t)g->mth[h] * g->NT * 256;
        g->part_words += (size_t)g->nch[h] * g->nsub[h] * g->per[h];
    }
    g->dsum_words = (size_t)g->nch[0] * g->ndi;
    g->sum_words = g->per[0] + g->per[1] + g->ndi;
    return true;
}
}  // namespace
int ajtai_i8g_scratch(const AjtaiI8Ring &R, u32 MT, size_t n, u32 NP, u32 nwg, size_t *part_words, size_t *dsum_words, size_t *sum_words) {
    GPlan g;
    if (!g_plan(R, MT, n, NP, nwg, &g)) return -1;
    *part_words = g.part_words; *dsum_words = g.dsum_words; *sum_words = g.sum_words;
    return 0;
}
int launch_ajtai_i8g(const AjtaiI8Ring &R, const unsigned char *Ab, u32 MT, const unsigned long long *pre, size_t ldw, size_t n, u32 kappa, u32 row0, u32 kappa_total,
                     u32 NP, u32 nwg, int32_t *part, int32_t *dsum, long long *sum, u64 *coef_out, hipStream_t s) {
    GPlan g;
    if (!g_plan(R, MT, n, NP, nwg, &g) || R.NL * kappa > 16 * MT) return -1;
    AjtaiI8GArgs a;
    a.Ab = Ab; a.MT = MT; a.pre = pre; a.ldw = ldw; a.NP = NP; a.ntiles = (u32)((n + 7) / 8); a.ft = g.ft;
    for (int h = 0; h < 2; h++) { a.m_lo[h] = g.m_lo[h]; a.mth[h] = g.mth[h]; a.nch[h] = g.nch[h]; a.tpw[h] = g.tpw[h]; a.nsub[h] = g.nsub[h]; }
    a.part[0] = part;
    a.part[1] = part + (size_t)g.nch[0] * g.nsub[0] * g.per[0];
    a.dsum = dsum;
    const dim3 grid(g.nch[0] + g.nch[1]), block(512);
#define LF_G_LAUNCH(RD_, MTW_, NTW_)                                                                                                                 \
    do {                                                                                                                                             \
        static bool attr_ = false;                                                                                                                   \
        if (!attr_) { (void)hipFuncSetAttribute((const void *)k_ajtai_i8g<RD_, MTW_, NTW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_ = true; } \
        typedef GX<RD_, MTW_, NTW_> GL_;                                                                                                             \
        hipLaunchKernelGGL((k_ajtai_i8g<RD_, MTW_, NTW_>), grid, block, GL_::lds_bytes(), s, a);                                                     \
    } while (0)
    static const bool gprof = getenv("LF_I8G_PROF") != nullptr;
    if (gprof && R.RD == 24) {
        static bool attr_p = false;
        if (!attr_p) {
            (void)hipFuncSetAttribute((const void *)k_ajtai_i8g<24, 7, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)k_ajtai_i8g<24, 7, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_p = true;
        }
        typedef GX<24, 7, 4> G4_;
        typedef GX<24, 7, 2> G2_;
        static const unsigned long long zeros[64] = {0};
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_i8g_prof), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice, s);
        if (g.NT == 16) hipLaunchKernelGGL((k_ajtai_i8g<24, 7, 4, true>), grid, block, G4_::lds_bytes(), s, a);
        else hipLaunchKernelGGL((k_ajtai_i8g<24, 7, 2, true>), grid, block, G2_::lds_bytes(), s, a);
    } else if (R.RD == 24 && g.NT == 16) LF_G_LAUNCH(24, 7, 4);
    else if (R.RD == 24) LF_G_LAUNCH(24, 7, 2);
    else LF_G_LAUNCH(72, 4, 6);
#undef LF_G_LAUNCH
    hipLaunchKernelGGL(k_ajtai_i8g_sum, dim3((unsigned)cdiv(g.sum_words, 256)), dim3(256), 0, s, a.part[0], g.per[0], g.nch[0] * g.nsub[0], a.part[1], g.per[1],
                       g.nch[1] * g.nsub[1], dsum, g.ndi, g.nch[0], sum);
    hipLaunchKernelGGL(k_ajtai_i8g_finish, dim3((unsigned)cdiv((size_t)kappa * R.RD, 256)), dim3(256), 0, s, sum, g.mth[0], g.mth[1], g.NT, g.ndi, NP, kappa, row0,
                       kappa_total, R.RD, R.NL, R.p_small, R.soa_out, coef_out);
    return (int)grid.x;
}
}  // namespace lf

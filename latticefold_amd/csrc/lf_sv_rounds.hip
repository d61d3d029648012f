// lf_sv_rounds.hip -- rounds 1..3 of the folding sumcheck as exact int8 GEMMs (gfx950 v_mfma_i32_16x16x64_i8); see lf_sv_rounds.h for
// the algebra.  Kernels:
//   k_sv_bits      int32 witness planes -> bit planes (one row of magnitude bits per digit plane + one row of sign bits per coefficient);
//   k_sv_pack_eq   eqB[3][2 npairs] -> digit planes EB[48][padded pairs] (balanced base-256 digits of eqB(2p) and eqB(2p+1), in the slot
//                  order of the A operand), MFMA B operand;
//   k_sv_gemm<V,PG> one wave = one (side, coefficient, 16 digit planes) group x one chunk of pairs x one group of PW (sigma, beta) pairs:
//                  per K-step of 64 pairs it expands the 2V signed bits of its 16 pairs from one word of magnitude bits and one of sign
//                  bits per V (shift + mask), builds the +-1 / 0 operand bytes of every (sigma, beta) with byte-wise AND / XOR, and
//                  issues 3 MFMAs per pair (column tiles: the digits of eqB(2p) 0-15, 16-23 | eqB(2p+1) 0-7, 8-23); the waves of a block
//                  share the eqB digits of a super-step (512 positions) through a double-buffered LDS tile;
//   k_sv_sum       element-wise sum of the chunks' partial tiles (int32: |sum| <= 128 * npairs < 2^31 up to 2^23 pairs);
//   k_sv_finish1   per table T: M_pi = sum_u 2^(8u) C_u mod p for eqB(2p) and eqB(2p+1), the table's degree-4 polynomial
//                  sum_pi C_pi(X) (M0 + X (M1 - M0)), times mu_T;
//   k_sv_finish2   sum over the tables of a slot, evaluation at X = 0..4, plus the G part of the message.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <utility>
#include "lf_field.cuh"
#include "lf_kernels.h"
#include "lf_sv_rounds.h"

namespace lf {
static inline size_t cdiv(size_t a, size_t b) { return (a + b - 1) / b; }
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr u32 ONES4 = 0x01010101u;
constexpr int ctz8(unsigned m) { int i = 0; while (i < 8 && !((m >> i) & 1)) i++; return i; }
constexpr int pop8(unsigned m) { int n = 0; for (int i = 0; i < 8; i++) n += (m >> i) & 1; return n; }

// Bit-plane form of the witness planes, the A-operand source of k_sv_gemm: per coefficient c the rows r < 16 ktiles hold bit r of |v| of
// every position (32 positions per word), row 16 ktiles the sign bits; positions padded with zeros to a multiple of 512.
// wave = 512 consecutive positions of one coefficient: 8 ballots per row, lane r keeps row r and writes 64 contiguous bytes.
__device__ __forceinline__ u32 sv_writelane(u32 val, u32 lane, u32 old) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(lane) : "m0");   // one SGPR + m0: the constant-bus limit of gfx9
    return old;
#else
    return old;
#endif
}
__global__ void __launch_bounds__(256) k_sv_bits(const int32_t *planes, size_t ldp, size_t n, size_t npad, u32 rows, u32 *bits, u32 rd) {
    const u32 lane = threadIdx.x & 63;
    const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), per_c = npad / 512;
    if (wid >= rd * per_c) return;
    const u32 c = (u32)(wid / per_c);
    const size_t base = (wid % per_c) * 512;
    u32 klo[8], khi[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const size_t pos = base + (size_t)it * 64 + lane;
        const int32_t v = pos < n ? planes[(size_t)c * ldp + pos] : 0;
        u32 m = v < 0 ? 0u - (u32)v : (u32)v;   // unsigned: v = INT32_MIN is a legal digit (B = 2^32)
        klo[it] = 0; khi[it] = 0;
        // row r of the wave's 64 positions = ballot of bit r; lane r receives it (v_writelane: no compare / select per row)
        const unsigned long long sg = __ballot(v < 0);
        klo[it] = sv_writelane((u32)sg, rows - 1, klo[it]);
        khi[it] = sv_writelane((u32)(sg >> 32), rows - 1, khi[it]);
        for (u32 r = 0; r + 1 < rows; r++) {
            const unsigned long long bal = __ballot((int)(m & 1u));
            m >>= 1;
            klo[it] = sv_writelane((u32)bal, r, klo[it]);
            khi[it] = sv_writelane((u32)(bal >> 32), r, khi[it]);
        }
    }
    if (lane < rows) {
        u32 *o = bits + ((size_t)c * rows + lane) * (npad / 32) + base / 32;
#pragma unroll
        for (int it = 0; it < 8; it += 2) *(uint4 *)(o + 2 * it) = make_uint4(klo[it], khi[it], klo[it + 1], khi[it + 1]);
    }
}

// slot (j, q) of a lane's 16 pairs (operand register j, byte q) holds pair  sv_slot_pair(V, j, q)  of the 16: chosen so that the operand bytes of
// a register are ONE shift + mask of a bit-plane word ((w >> s) & 0x01010101 picks bits s + 8 q)
__host__ __device__ constexpr int sv_slot_pair(int V, int j, int q) { return (j / (4 / V)) * (16 / V) + (j % (4 / V)) + (4 / V) * q; }

// EB[(24 h + 8 q + u)][slot] = balanced base-256 digit u of eqB[q][2 pair + h] in slot order;  thread = (16 pairs, word q, half h);  ld = padded pairs
// single: eq holds ONE value per pair (the split form's E_i[p], see launch_sv_round): rows 8 q + u only (24 digit rows; the caller zeroes rows 24..31)
__global__ void __launch_bounds__(256) k_sv_pack_eq(const u64 *eq, size_t ld, size_t npairs, size_t ldeb, int V, unsigned char *EB, int single) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, groups = ldeb / 16;
    if (gid >= groups * (single ? 3 : 6)) return;
    const u32 qh = (u32)(gid / groups), q = qh % 3, h = qh / 3;
    const size_t p0 = (gid % groups) * 16;
    u64 w[16];
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const size_t pr = p0 + (size_t)sv_slot_pair(V, t >> 2, t & 3);
        // balanced base-256 digits d_u in [-128, 127] of a representative of the word mod p: sum_u d_u 256^u = E or E - p.  With
        // s = E' + 0x80..80 in [0, 2^64) the digits are (byte_u(s) - 128), i.e. byte_u(s) ^ 0x80 as int8 -- no bias column in the GEMM
        const u64 e = pr < npairs ? (single ? eq[(size_t)q * ld + pr] : eq[(size_t)q * ld + 2 * pr + h]) : 0;
        u64 sb = e + 0x8080808080808080ull;
        if (sb < e) sb += 0xFFFFFFFFull;          // wrapped past 2^64: take E - p instead (2^64 - p = 2^32 - 1)
        w[t] = sb ^ 0x8080808080808080ull;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
        u32 o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 16; t++) o[t >> 2] |= (u32)((w[t] >> (8 * u)) & 0xFF) << (8 * (t & 3));
        *(uint4 *)(EB + (size_t)(24 * h + 8 * q + u) * ldeb + p0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// x * 255 for x with one bit per byte, as shift + subtract (the compiler would turn (x << 8) - x into a quarter-rate 32-bit multiply)
__device__ __forceinline__ u32 sv_full(u32 x) {
    u32 t, r;
    asm("v_lshlrev_b32 %0, 8, %1" : "=v"(t) : "v"(x));
    asm("v_sub_u32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(x));
    return r;
}
struct SvGemmArgs {
    const u32 *bits[2];         // k_sv_bits form of the two witnesses
    u32 rows;                   // 16 ktiles + 1
    size_t nw;                  // words per row
    const unsigned char *EB;    // [48][ldeb]
    size_t ldeb;
    u32 ktiles;                 // tiles of 16 digit planes
    u32 nsides;                 // 2 witnesses (round GEMMs) or 1 (launch_sv_vs)
    u32 nsuper, super_per_chunk;   // super-steps of 512 positions (256 / V pairs)
    u32 eb_mod;                 // 0, or: the eqB bytes repeat every eb_mod super-steps (launch_sv_vs_blocks: the weights of the low index bits only)
    u32 rd;                     // coefficient groups per side: 24 (Goldilocks ring) / 72 (BabyBear ring, bb_sv_rounds.hip)
    u32 super0;                 // first super-step of the bit planes (a rank's pair slice of a sharded step; the eqB bytes start at its first pair)
    int32_t *part;              // [chunk][group][pair][3][64][4]
};

// operand register j of pair IDX:  +-1 where every bit of beta is set, sign = product of the signs of sigma
template <int V, int IDX>
__device__ __forceinline__ u32 sv_operand(const u32 (&Xf)[2 * V][4], const u32 (&S)[2 * V][4], const u32 (&D)[2 * V][4], int j) {
    constexpr SvPair pr = sv_pair(V, IDX);
    constexpr int nb = pop8(pr.b);
    if constexpr (nb == 1) {
        return D[ctz8(pr.b)][j];
    } else if constexpr (nb == 2) {
        constexpr int z = ctz8(pr.s), q = ctz8(pr.b & ~pr.s);
        return Xf[q][j] & D[z][j];
    } else {
        constexpr int x = ctz8(pr.b), y = ctz8(pr.b & ~(1u << x)), z = ctz8(pr.b & ~(1u << x) & ~(1u << y));
        return (Xf[x][j] & Xf[y][j] & Xf[z][j]) & ((S[x][j] ^ S[y][j] ^ S[z][j]) | ONES4);
    }
}
// VS (launch_sv_vs*: V = 1, NT = 3): only the two single pairs are read afterwards, pair 0 against the digits of eq(2p) (columns 0..23: tiles 0, 1) and pair 2
// against those of eq(2p+1) (columns 24..47: tiles 1, 2) -- 4 of the 12 MFMAs of a K-step
template <int V, int BASE, int I, int NT, bool VS>
__device__ __forceinline__ void sv_mfma_pair(v4i (&acc)[sv_pairs_per_wave(V, NT)][NT], const v4i (&b)[NT], const u32 (&Xf)[2 * V][4], const u32 (&S)[2 * V][4],
                                             const u32 (&D)[2 * V][4]) {
    if constexpr (VS && (I == 1 || I == 3)) return;
    v4i av;
    av.x = (int)sv_operand<V, BASE + I>(Xf, S, D, 0);
    av.y = (int)sv_operand<V, BASE + I>(Xf, S, D, 1);
    av.z = (int)sv_operand<V, BASE + I>(Xf, S, D, 2);
    av.w = (int)sv_operand<V, BASE + I>(Xf, S, D, 3);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
        if (!VS || (I == 0 && nt < 2) || (I == 2 && nt > 0)) acc[I][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[nt], acc[I][nt], 0, 0, 0);
}
template <int V, int BASE, int NT, bool VS, int... I>
__device__ __forceinline__ void sv_mfma_all(v4i (&acc)[sv_pairs_per_wave(V, NT)][NT], const v4i (&b)[NT], const u32 (&Xf)[2 * V][4], const u32 (&S)[2 * V][4],
                                            const u32 (&D)[2 * V][4], std::integer_sequence<int, I...>) {
    (sv_mfma_pair<V, BASE, I, NT, VS>(acc, b, Xf, S, D), ...);
}

// waves per block = groups that share the eqB bytes of a super-step through LDS (every group needs all of them: read from L2 once per block)
constexpr int sv_waves(int V) { return V <= 2 ? 8 : 4; }
// NT = column tiles: 3 for the 48 digit columns of (eqB(2p), eqB(2p+1)), 2 for the 24 (+ 8 zero) columns of the split form's one value per pair
template <int V, int PG, int NT = 3, bool VS = false>
__global__ void __launch_bounds__(64 * sv_waves(V)) __attribute__((amdgpu_waves_per_eu(V <= 2 ? 2 : 1, V <= 2 ? 2 : 1))) k_sv_gemm(SvGemmArgs a) {
    static_assert(!VS || (V == 1 && NT == 3 && PG == 0), "VS: the two single pairs of the round-1 shape");
    constexpr int NX = 2 * V, PW = sv_pairs_per_wave(V, NT), NPR = sv_num_pairs(V), SS = 4 / V, RPW = 4 / V;   // sub-steps per super-step, registers per word
    constexpr int NW = sv_waves(V), NTH = 64 * NW;
    constexpr int SPAIRS = 256 / V, LROW = SPAIRS + 16;            // pairs (= bytes per eqB row) of a super-step; padded LDS row: conflict-free b128 reads
    constexpr int PIECES = 16 * NT * SPAIRS / 16, PPT = (PIECES + NTH - 1) / NTH;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][16 * NT][LROW];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 15, g = lane >> 4;
    const u32 ngroups = a.rd * a.nsides * a.ktiles, gq = ngroups / NW;
    const u32 grp = (blockIdx.x % gq) * NW + wave, chunk = blockIdx.x / gq;
    const u32 kt = grp % a.ktiles, c = (grp / a.ktiles) % a.rd, side = grp / (a.ktiles * a.rd);
    const u32 *mrow = a.bits[side] + ((size_t)c * a.rows + 16 * kt + row) * a.nw;
    const u32 *srow = a.bits[side] + ((size_t)c * a.rows + a.rows - 1) * a.nw;
    v4i acc[PW][NT];
#pragma unroll
    for (int i = 0; i < PW; i++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[i][nt] = v4i{0, 0, 0, 0};
    const u32 u0 = chunk * a.super_per_chunk, u1 = u0 + a.super_per_chunk < a.nsuper ? u0 + a.super_per_chunk : a.nsuper;
    // staging of the eqB bytes: piece t = 16 bytes of row t / (SPAIRS/16)
    v4i st[PPT];
    auto stage_load = [&](u32 u) {
        const u32 ue = a.eb_mod ? u % a.eb_mod : u;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const u32 t = tid + i * NTH;
            if (PIECES % NTH == 0 || t < PIECES) st[i] = *(const v4i *)(a.EB + (size_t)(t / (SPAIRS / 16)) * a.ldeb + (size_t)ue * SPAIRS + 16 * (t % (SPAIRS / 16)));
        }
    };
    auto stage_store = [&](u32 buf) {
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const u32 t = tid + i * NTH;
            if (PIECES % NTH == 0 || t < PIECES) *(v4i *)&lds[buf][t / (SPAIRS / 16)][16 * (t % (SPAIRS / 16))] = st[i];
        }
    };
    if (u0 < u1) { stage_load(u0); stage_store(0); }
    uint4 wm4 = make_uint4(0, 0, 0, 0), ws4 = wm4;
    mrow += (size_t)a.super0 * 16; srow += (size_t)a.super0 * 16;
    if (u0 < u1) { wm4 = *(const uint4 *)(mrow + ((size_t)u0 * 4 + g) * 4); ws4 = *(const uint4 *)(srow + ((size_t)u0 * 4 + g) * 4); }
    __syncthreads();
    for (u32 u = u0; u < u1; u++) {
        const u32 buf = (u - u0) & 1;
        const u32 wm[4] = {wm4.x, wm4.y, wm4.z, wm4.w}, wsg[4] = {ws4.x, ws4.y, ws4.z, ws4.w};
        const u32 un = u + 1 < u1 ? u + 1 : u;
        stage_load(un);                                  // in flight while this super-step is computed
        wm4 = *(const uint4 *)(mrow + ((size_t)un * 4 + g) * 4);
        ws4 = *(const uint4 *)(srow + ((size_t)un * 4 + g) * 4);
#pragma unroll
        for (int ss = 0; ss < SS; ss++) {
            v4i b[NT];
            const unsigned char *lb = &lds[buf][row][g * (64 / V) + 16 * ss];
#pragma unroll
            for (int nt = 0; nt < NT; nt++) b[nt] = *(const v4i *)(lb + 16 * nt * LROW);
            u32 Xf[NX][4], S[NX][4], D[NX][4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32 wmj = wm[ss * V + j / RPW], wsj = wsg[ss * V + j / RPW];
#pragma unroll
                for (int x = 0; x < NX; x++) {
                    const int sh = NX * (j % RPW) + x;
                    const u32 x1 = (wmj >> sh) & ONES4, s1 = (wsj >> sh) & ONES4;
                    Xf[x][j] = sv_full(x1);   // 0x01 -> 0xFF per byte
                    S[x][j] = sv_full(s1);
                    D[x][j] = Xf[x][j] & (S[x][j] | ONES4);
                }
            }
            sv_mfma_all<V, PG * PW, NT, VS>(acc, b, Xf, S, D, std::make_integer_sequence<int, PW>{});
        }
        stage_store(buf ^ 1);
        __syncthreads();
    }
    int32_t *o = a.part + (((size_t)chunk * ngroups + grp) * NPR + (size_t)PG * PW) * (NT * 256);
#pragma unroll
    for (int i = 0; i < PW; i++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) *(v4i *)(o + ((size_t)i * NT + nt) * 256 + lane * 4) = acc[i][nt];
}

__global__ void __launch_bounds__(256) k_sv_sum(const int32_t *part, size_t words, u32 chunks, int32_t *tot) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= words) return;
    v4i s = v4i{0, 0, 0, 0};
    for (u32 ch = 0; ch < chunks; ch++) s += *(const v4i *)(part + (size_t)ch * words + i);
    *(v4i *)(tot + i) = s;
}

// block = table T = (side, k, c), thread = pair pi
// split (nt == 2): the tiles hold M' = sum_p E[p] (+-1 | 0) and M_h = w_h M' (w_h = c_i eq(beta_i, h): eqB(2p + h) = w_h E[p])
template <bool NU>
__global__ void __launch_bounds__(128) k_sv_finish1(DevCrt t, const int32_t *tot, u32 npr, u32 K, u32 ktiles, const u64 *coef, const Fq3Const *mu_pow, u64 *tp, u32 nt, Fq3Const w0,
                                                    Fq3Const w1) {
    const u32 T = blockIdx.x, c = T % 24, k = (T / 24) % K, side = T / (24 * K);
    const u32 grp = (side * 24 + c) * ktiles + k / 16, krow = k & 15;
    __shared__ u64 sm[15][128];
    const u32 pi = threadIdx.x;
    Fq3 P[5];
#pragma unroll
    for (int e = 0; e < 5; e++) P[e] = fq3_zero();
    if (pi < npr) {
        const int32_t *base = tot + ((size_t)grp * npr + pi) * (nt * 256);
        auto cell = [&](u32 bp) { return (long long)base[(bp >> 4) * 256 + ((bp & 15) + 16 * (krow >> 2)) * 4 + (krow & 3)]; };
        Fq3 M[2];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 3; q++) {
                __int128 v = 0;
                if (nt == 3 || h == 0)
                    for (u32 u = 0; u < 8; u++) v += (__int128)cell(24 * h + 8 * q + u) << (8 * u);
                M[h].c[q] = fq_from_s128((u64)v, (int64_t)(v >> 64));
            }
        if (nt != 3) {
            const Fq3 Mp = M[0];
            M[0] = fq3_mul<NU>(Mp, fq3_make(w0.c[0], w0.c[1], w0.c[2]), t.nu);
            M[1] = fq3_mul<NU>(Mp, fq3_make(w1.c[0], w1.c[1], w1.c[2]), t.nu);
        }
        const Fq3 dM = fq3_sub(M[1], M[0]);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const u64 *cp = coef + ((size_t)pi * 4 + e) * 3;
            const Fq3 C = fq3_make(cp[0], cp[1], cp[2]);
            P[e] = fq3_add(P[e], fq3_mul<NU>(C, M[0], t.nu));
            P[e + 1] = fq3_add(P[e + 1], fq3_mul<NU>(C, dM, t.nu));
        }
    }
#pragma unroll
    for (int e = 0; e < 5; e++)
#pragma unroll
        for (int q = 0; q < 3; q++) sm[3 * e + q][threadIdx.x] = P[e].c[q];
    __syncthreads();
    if (threadIdx.x < 15) {
        u64 s = 0;
        for (u32 i = 0; i < 128; i++) s = fq_add(s, sm[threadIdx.x][i]);
        sm[threadIdx.x][0] = s;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const u32 e = threadIdx.x;
        const Fq3Const mc = mu_pow[(side * K + k) * 3 + c / 8];
        const Fq3 r = fq3_mul<NU>(fq3_make(mc.c[0], mc.c[1], mc.c[2]), fq3_make(sm[3 * e][0], sm[3 * e + 1][0], sm[3 * e + 2][0]), t.nu);
#pragma unroll
        for (int q = 0; q < 3; q++) tp[((size_t)T * 5 + e) * 3 + q] = fq_canon(r.c[q]);
    }
}
// thread = (output i = X*24 + 3*slot + q, part j of 8): out[i] = gpart[i] + sum_{side,k,d} TP[(side,k,8d+slot)](X)
__global__ void __launch_bounds__(1024) k_sv_finish2(const u64 *tp, u32 K, const u64 *gpart, u64 *out) {
    __shared__ u64 sm[8][128];
    const u32 i = threadIdx.x & 127, j = threadIdx.x >> 7;
    const u32 X = i / 24, slot = (i % 24) / 3, q = i % 3;
    u64 v = 0;
    if (i < 120) {
        u64 co[5] = {0, 0, 0, 0, 0};
        for (u32 sk = j; sk < 2 * K; sk += 8)
            for (u32 d = 0; d < 3; d++) {
                const u64 *p = tp + ((size_t)(sk * 24 + 8 * d + slot) * 5) * 3 + q;
#pragma unroll
                for (int e = 0; e < 5; e++) co[e] = fq_add(co[e], p[3 * e]);
            }
        v = co[4];
        for (int e = 3; e >= 0; e--) v = fq_add(fq_mul(v, (u64)X), co[e]);
    }
    sm[j][i] = v;
    __syncthreads();
    if (j == 0 && i < 120) {
        u64 s = fq_canon(gpart[i]);
#pragma unroll
        for (int w = 0; w < 8; w++) s = fq_add(s, sm[w][i]);
        out[i] = fq_canon(s);
    }
}

template <int V, int PG, int NT = 3>
void launch_gemm_pg(const SvGemmArgs &a, u32 grid, hipStream_t s) {
    hipLaunchKernelGGL((k_sv_gemm<V, PG, NT>), dim3(grid), dim3(64 * sv_waves(V)), 0, s, a);
    if constexpr (PG + 1 < sv_num_pairs(V) / sv_pairs_per_wave(V, NT)) launch_gemm_pg<V, PG + 1, NT>(a, grid, s);
}

bool sv_shape_ok(int V, size_t npairs, uint32_t K) {
    return (V == 1 || V == 2 || V == 4) && npairs >= 64 && npairs % 16 == 0 && npairs <= ((size_t)1 << 23) && K >= 1 && K <= 32;
}
static size_t sv_ldeb(size_t npairs) { return (npairs + 255) / 256 * 256; }
size_t sv_eb_bytes(size_t npairs) { return 48 * sv_ldeb(npairs) + 64; }
static size_t sv_npad(size_t n) { return (n + 511) / 512 * 512; }
size_t sv_bits_words(size_t n, uint32_t K, uint32_t rd) { return (size_t)rd * (16 * ((K + 15) / 16) + 1) * (sv_npad(n) / 32); }
void launch_sv_bits(const int32_t *planes, size_t ldp, size_t n, uint32_t K, uint32_t *bits, hipStream_t s, uint32_t rd) {
    const size_t npad = sv_npad(n);
    hipLaunchKernelGGL(k_sv_bits, dim3((unsigned)cdiv(rd * (npad / 512), 4)), dim3(256), 0, s, planes, ldp, n, npad, 16 * ((K + 15) / 16) + 1, bits, rd);
}
// chunks of super-steps (512 positions): the launches of the pair groups run one after the other, a block fills a CU (registers), so one launch
// is one batch of at most 256 blocks
static uint32_t sv_chunks_n(int V, size_t nsuper, uint32_t K, uint32_t nsides, uint32_t rd = 24) {
    const u32 ktiles = (K + 15) / 16, per_chunk = rd * nsides * ktiles / (u32)sv_waves(V);
    if (nsuper == 0) return 1;
    size_t want = 256 / per_chunk;
    if (want > nsuper) want = nsuper;
    if (want < 1) want = 1;
    const size_t spc = cdiv(nsuper, want);
    return (u32)cdiv(nsuper, spc);
}
uint32_t sv_chunks(int V, size_t nsuper, uint32_t K) { return sv_chunks_n(V, nsuper, K, 2); }
size_t sv_tot_words(int V, uint32_t K) { return (size_t)48 * ((K + 15) / 16) * sv_num_pairs(V) * 768; }
// sized for the largest chunk count of the shape: the launch clamps the pairs to the witness length and sv_chunks_n is not monotone in the
// number of super-steps
static uint32_t sv_max_chunks(int V, uint32_t K, uint32_t nsides, uint32_t rd = 24) {
    const u32 per_chunk = rd * nsides * ((K + 15) / 16) / (u32)sv_waves(V);
    return per_chunk >= 256 ? 1 : 256 / per_chunk;
}
size_t sv_part_words(int V, size_t, uint32_t K) { return sv_tot_words(V, K) * sv_max_chunks(V, K, 2); }
size_t sv_tp_words(uint32_t K) { return (size_t)2 * K * 24 * 15; }

// ---- the GEMM stage alone, for the BabyBear backend (bb_sv_rounds.hip): rd coefficient groups per side, three column tiles of digit bytes EB[48][ldeb] (the
// caller packed them in slot order, sv_slot_pair), all sv_num_pairs(V) digit monomials; leaves the summed tiles in tot:
//   tot[((side * rd + c) * ktiles + kt) * npr + pi][tile][64 lanes][4]   (cell (digit plane row, digit column) as k_sv_finish1 reads it)
size_t sv_tot_words_rd(uint32_t rd, int V, uint32_t K) { return (size_t)2 * rd * ((K + 15) / 16) * sv_num_pairs(V) * 768; }
size_t sv_part_words_rd(uint32_t rd, int V, uint32_t K) { return sv_tot_words_rd(rd, V, K) * sv_max_chunks(V, K, 2, rd); }
size_t sv_ldeb_pub(size_t npairs) { return (npairs + 255) / 256 * 256; }
int launch_sv_gemm_tiles(uint32_t rd, int V, const uint32_t *bitsL, const uint32_t *bitsR, size_t nplanes, const unsigned char *EB, size_t ldeb, size_t npairs, uint32_t K,
                         int32_t *part, int32_t *tot, hipStream_t s) {
    if (!sv_shape_ok(V, npairs, K) || (2 * rd * ((K + 15) / 16)) % (u32)sv_waves(V)) return -1;
    const size_t wall = cdiv(nplanes, 2 * (size_t)V), wpairs = wall < npairs ? wall : npairs;
    SvGemmArgs a;
    a.bits[0] = bitsL; a.bits[1] = bitsR;
    a.ktiles = (K + 15) / 16;
    a.nsides = 2;
    a.rd = rd;
    a.rows = 16 * a.ktiles + 1;
    a.nw = sv_npad(nplanes) / 32;
    a.EB = EB; a.ldeb = ldeb;
    a.nsuper = (u32)cdiv(wpairs * V, 256);
    a.super0 = 0;
    a.eb_mod = 0;
    const u32 chunks = sv_chunks_n(V, a.nsuper, K, 2, rd);
    a.super_per_chunk = (u32)cdiv(a.nsuper, chunks);
    a.part = part;
    const u32 grid = 2 * rd * a.ktiles / (u32)sv_waves(V) * chunks;
    if (V == 1) launch_gemm_pg<1, 0>(a, grid, s);
    else if (V == 2) launch_gemm_pg<2, 0>(a, grid, s);
    else launch_gemm_pg<4, 0>(a, grid, s);
    const size_t words = sv_tot_words_rd(rd, V, K);
    hipLaunchKernelGGL(k_sv_sum, dim3((unsigned)cdiv(words / 4, 256)), dim3(256), 0, s, part, words, chunks, tot);
    return 0;
}

int launch_sv_round(const DevCrt &t, int V, const uint32_t *bitsL, const uint32_t *bitsR, size_t nplanes, const uint64_t *eqB, size_t ldeq, size_t pair0, size_t npairs, uint32_t K,
                    const Fq3Const *mu_pow, const uint64_t *coef, unsigned char *EB, int32_t *part, int32_t *tot, uint64_t *tp, const uint64_t *gpart, uint64_t *out,
                    hipStream_t s, hipEvent_t gpart_ready, const uint64_t *E, size_t ldE, const Fq3Const *w01) {
    // pairs of the slice behind which witness positions exist (positions >= nplanes are zero digits: nothing to add)
    const size_t wall = cdiv(nplanes, 2 * (size_t)V);
    const size_t wpairs = pair0 >= wall ? 0 : (wall - pair0 < npairs ? wall - pair0 : npairs);
    if (!sv_shape_ok(V, npairs, K) || (pair0 * V) % 256) return -1;
    const size_t ldeb = sv_ldeb(npairs);
    const bool split = E != nullptr;
    const u32 NT = split ? 2u : 3u;
    if (split) {   // one value per pair: 24 digit rows + 8 zero rows = two column tiles instead of three
        (void)hipMemsetAsync(EB + 24 * ldeb, 0, 8 * ldeb, s);
        hipLaunchKernelGGL(k_sv_pack_eq, dim3((unsigned)cdiv(ldeb / 16 * 3, 256)), dim3(256), 0, s, E + pair0, ldE, npairs, ldeb, V, EB, 1);
    } else
    hipLaunchKernelGGL(k_sv_pack_eq, dim3((unsigned)cdiv(ldeb / 16 * 6, 256)), dim3(256), 0, s, eqB + 2 * pair0, ldeq, npairs, ldeb, V, EB, 0);
    SvGemmArgs a;
    a.bits[0] = bitsL; a.bits[1] = bitsR;
    a.ktiles = (K + 15) / 16;
    a.nsides = 2;
    a.rd = 24;
    a.rows = 16 * a.ktiles + 1;
    a.nw = sv_npad(nplanes) / 32;
    a.EB = EB; a.ldeb = ldeb;
    a.nsuper = (u32)cdiv(wpairs * V, 256);
    a.super0 = (u32)(pair0 * V / 256);
    a.eb_mod = 0;
    const u32 chunks = sv_chunks(V, a.nsuper, K);
    a.super_per_chunk = (u32)cdiv(a.nsuper, chunks);
    a.part = part;
    const u32 grid = 48 * a.ktiles / (u32)sv_waves(V) * chunks;
    if (split) {
        if (V == 1) launch_gemm_pg<1, 0, 2>(a, grid, s);
        else if (V == 2) launch_gemm_pg<2, 0, 2>(a, grid, s);
        else launch_gemm_pg<4, 0, 2>(a, grid, s);
    } else
    if (V == 1) launch_gemm_pg<1, 0>(a, grid, s);
    else if (V == 2) launch_gemm_pg<2, 0>(a, grid, s);
    else launch_gemm_pg<4, 0>(a, grid, s);
    const size_t words = sv_tot_words(V, K) / 3 * NT;
    hipLaunchKernelGGL(k_sv_sum, dim3((unsigned)cdiv(words / 4, 256)), dim3(256), 0, s, part, words, chunks, tot);
    const u32 npr = (u32)sv_num_pairs(V);
    const Fq3Const wz = {{0, 0, 0}}, w0 = split ? w01[0] : wz, w1 = split ? w01[1] : wz;
    if (t.nu2p40) hipLaunchKernelGGL((k_sv_finish1<true>), dim3(2 * K * 24), dim3(128), 0, s, t, tot, npr, K, a.ktiles, coef, mu_pow, tp, NT, w0, w1);
    else hipLaunchKernelGGL((k_sv_finish1<false>), dim3(2 * K * 24), dim3(128), 0, s, t, tot, npr, K, a.ktiles, coef, mu_pow, tp, NT, w0, w1);
    if (gpart_ready) (void)hipStreamWaitEvent(s, gpart_ready, 0);   // the G part was computed on another stream
    hipLaunchKernelGGL(k_sv_finish2, dim3(1), dim3(1024), 0, s, tp, K, gpart, out);
    return 0;
}

// v_s[k][c][q] = sum_i eq[q][i] * digit_k(planes[c][i]) (decomposition.rs:204-211) from the bit-plane form of ONE witness with the round-1
// GEMM: its two single pairs ({x},{x}), x = 0 / 1, against the digits of eq(2p) / eq(2p+1) are the even / odd halves of that sum.
// thread = output (k, c, q): out[(k*24 + c)*3 + q]
__global__ void __launch_bounds__(256) k_sv_vs_finish(const int32_t *tot, u32 K, u32 ktiles, u64 *out) {
    const u32 o = blockIdx.x * 256 + threadIdx.x;
    if (o >= K * 72) return;
    const u32 q = o % 3, c = (o / 3) % 24, k = o / 72;
    const u32 grp = c * ktiles + k / 16, krow = k & 15;
    __int128 v = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int32_t *base = tot + ((size_t)grp * 4 + (h ? 2 : 0)) * 768;   // pair 0 = ({0},{0}), pair 2 = ({1},{1}) in the canonical order of V = 1
        for (u32 u = 0; u < 8; u++) {
            const u32 bp = 24 * h + 8 * q + u;
            v += (__int128)base[(bp >> 4) * 256 + ((bp & 15) + 16 * (krow >> 2)) * 4 + (krow & 3)] << (8 * u);
        }
    }
    out[o] = fq_from_s128((u64)v, (int64_t)(v >> 64));
}
size_t sv_vs_part_words(size_t, uint32_t K) { return sv_tot_words(1, K) / 2 * sv_max_chunks(1, K, 1); }
size_t sv_vs_tot_words(uint32_t K) { return sv_tot_words(1, K) / 2; }
// bits: launch_sv_bits form of the witness (n positions, n even); eq [3][ldeq]; EB / part / tot: scratch (sv_eb_bytes(n / 2), sv_vs_part_words, sv_vs_tot_words).
// Returns 0, or -1 if the shape is not handled.
int launch_sv_vs(const uint32_t *bits, size_t n, const uint64_t *eq, size_t ldeq, uint32_t K, unsigned char *EB, int32_t *part, int32_t *tot, uint64_t *out, hipStream_t s) {
    static_assert(sv_pair(1, 0).s == 1 && sv_pair(1, 0).b == 1 && sv_pair(1, 2).s == 2 && sv_pair(1, 2).b == 2, "single pairs of V = 1");
    const size_t npairs = n / 2;
    if ((n & 1) || !sv_shape_ok(1, npairs, K)) return -1;
    const size_t ldeb = sv_ldeb(npairs);
    hipLaunchKernelGGL(k_sv_pack_eq, dim3((unsigned)cdiv(ldeb / 16 * 6, 256)), dim3(256), 0, s, eq, ldeq, npairs, ldeb, 1, EB, 0);
    SvGemmArgs a;
    a.bits[0] = bits; a.bits[1] = bits;
    a.ktiles = (K + 15) / 16;
    a.nsides = 1;
    a.rd = 24;
    a.rows = 16 * a.ktiles + 1;
    a.nw = sv_npad(n) / 32;
    a.EB = EB; a.ldeb = ldeb;
    a.nsuper = (u32)cdiv(npairs, 256);
    a.super0 = 0;
    a.eb_mod = 0;
    const u32 chunks = sv_chunks_n(1, a.nsuper, K, 1);
    a.super_per_chunk = (u32)cdiv(a.nsuper, chunks);
    a.part = part;
    hipLaunchKernelGGL((k_sv_gemm<1, 0, 3, true>), dim3(24 * a.ktiles / (u32)sv_waves(1) * chunks), dim3(64 * sv_waves(1)), 0, s, a);
    const size_t words = sv_vs_tot_words(K);
    hipLaunchKernelGGL(k_sv_sum, dim3((unsigned)cdiv(words / 4, 256)), dim3(256), 0, s, part, words, chunks, tot);
    hipLaunchKernelGGL(k_sv_vs_finish, dim3((unsigned)cdiv((size_t)K * 72, 256)), dim3(256), 0, s, tot, K, a.ktiles, out);
    return 0;
}
// The same evaluation in two steps, so that its pass over the witness can start BEFORE the evaluation point is complete: with the first J coordinates
// r_1..r_J (low index bits) known, launch_sv_vs_blocks leaves per block b of 2^J positions the partial tiles of  T_b = sum_{i_lo} eq_lo[i_lo] digit_k(f[b 2^J + i_lo])
// (eq_lo = eq((r_1..r_J), .), [3][ldeq], 2^J entries; one chunk of the GEMM per block, the eqB bytes repeat from block to block); once the other
// coordinates are known, launch_sv_vs_combine adds  v_s[k][c] = sum_b w_b T_b  with w_b = eq((r_{J+1}..), b) (F_{p^3} weights, [nblocks][3] device words).
// part: sv_vs_blocks_part_words(nblocks, K) int32 words, nblocks <= sv_vs_max_blocks(K); EB: sv_eb_bytes(2^(J-1)).  Returns 0, or -1 if the shape is not handled.
size_t sv_vs_max_blocks(uint32_t) { return 256; }
size_t sv_vs_blocks_part_words(uint32_t nblocks, uint32_t K) { return sv_vs_tot_words(K) * nblocks; }
int launch_sv_vs_blocks(const uint32_t *bits, size_t n, const uint64_t *eq_lo, size_t ldeq, uint32_t J, uint32_t K, unsigned char *EB, int32_t *part, hipStream_t s) {
    if (J < 10 || J > 24 || (n & (((size_t)1 << J) - 1)) || n == 0) return -1;
    const size_t bs = (size_t)1 << J, nblocks = n / bs, lo_pairs = bs / 2;
    if (nblocks > sv_vs_max_blocks(K) || !sv_shape_ok(1, n / 2, K)) return -1;
    const size_t ldeb = sv_ldeb(lo_pairs);
    hipLaunchKernelGGL(k_sv_pack_eq, dim3((unsigned)cdiv(ldeb / 16 * 6, 256)), dim3(256), 0, s, eq_lo, ldeq, lo_pairs, ldeb, 1, EB, 0);
    SvGemmArgs a;
    a.bits[0] = bits; a.bits[1] = bits;
    a.ktiles = (K + 15) / 16;
    a.nsides = 1;
    a.rd = 24;
    a.rows = 16 * a.ktiles + 1;
    a.nw = sv_npad(n) / 32;
    a.EB = EB; a.ldeb = ldeb;
    a.nsuper = (u32)(n / 512);
    a.super0 = 0;
    a.eb_mod = (u32)(bs / 512);
    a.super_per_chunk = a.eb_mod;
    a.part = part;
    hipLaunchKernelGGL((k_sv_gemm<1, 0, 3, true>), dim3(24 * a.ktiles / (u32)sv_waves(1) * (u32)nblocks), dim3(64 * sv_waves(1)), 0, s, a);
    return 0;
}
// one wave per output (k, c), lane = block b (the blocks are independent until the weighted sum): out[(k*24 + c)*3 + q]
template <bool NU>
__global__ void __launch_bounds__(64) k_sv_vs_combine(DevCrt t, const int32_t *part, u32 nblocks, size_t words, const u64 *wts, u32 K, u32 ktiles, u64 *out) {
    const u32 o = blockIdx.x, lane = threadIdx.x;
    const u32 c = o % 24, k = o / 24;
    const u32 grp = c * ktiles + k / 16, krow = k & 15;
    Fq3 acc = fq3_zero();
    for (u32 b = lane; b < nblocks; b += 64) {
        const int32_t *tot = part + (size_t)b * words;
        u64 tq[3];
#pragma unroll
        for (u32 q = 0; q < 3; q++) {
            __int128 v = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int32_t *base = tot + ((size_t)grp * 4 + (h ? 2 : 0)) * 768;
#pragma unroll
                for (u32 u = 0; u < 8; u++) {
                    const u32 bp = 24 * h + 8 * q + u;
                    v += (__int128)base[(bp >> 4) * 256 + ((bp & 15) + 16 * (krow >> 2)) * 4 + (krow & 3)] << (8 * u);
                }
            }
            tq[q] = fq_from_s128((u64)v, (int64_t)(v >> 64));
        }
        acc = fq3_add(acc, fq3_mul<NU>(fq3_make(tq[0], tq[1], tq[2]), fq3_make(wts[3 * b], wts[3 * b + 1], wts[3 * b + 2]), t.nu));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int q = 0; q < 3; q++) acc.c[q] = fq_add(acc.c[q], __shfl_down(acc.c[q], off, 64));
    if (lane == 0) { out[(size_t)o * 3] = acc.c[0]; out[(size_t)o * 3 + 1] = acc.c[1]; out[(size_t)o * 3 + 2] = acc.c[2]; }
}
void launch_sv_vs_combine(const DevCrt &t, const int32_t *part, uint32_t nblocks, const uint64_t *wts, uint32_t K, uint64_t *out, hipStream_t s) {
    const u32 ktiles = (K + 15) / 16;
    const size_t words = sv_vs_tot_words(K);
    if (t.nu2p40) hipLaunchKernelGGL((k_sv_vs_combine<true>), dim3(K * 24), dim3(64), 0, s, t, part, nblocks, words, wts, K, ktiles, out);
    else hipLaunchKernelGGL((k_sv_vs_combine<false>), dim3(K * 24), dim3(64), 0, s, t, part, nblocks, words, wts, K, ktiles, out);
}
}  // namespace lf

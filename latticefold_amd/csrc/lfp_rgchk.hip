// lfp_rgchk.hip -- gfx950 kernels of the LatticeFold+ monomial set check and range check on the Frog ring (coefficient form):
//   In::set_check   crates/latticefold-plus/src/setchk.rs:65-262     Rg::range_check   src/rgchk.rs:81-186
//
// The coefficient ring's challenges are single F_p words, so the set-check sumcheck runs on tables of F_p words (ev(m, beta) of a unit
// monomial X^e is beta^e: a 16-entry look-up; its square beta^2e another), kept in Montgomery form.  The evaluations of Step 3 -- multilinear
// extensions of columns of monomials at the sumcheck point -- are sums of eq(r, row) binned by exponent: out[t] = sum over the rows with
// exponent t (scalar weights), or sums of negacyclic rotations of ring weights (the M_i f rows: weights M_i^T eq).  Monomial sets cross
// every interface as int8 exponent digits d in (-8, 8) (exp(d) = X^d, X^(16+d) for d < 0; lfplus.h); LFP_ABSENT marks a zero entry.
// HBM-bound integer work; no MFMA.
#include "lfp_kernels.h"
#include "lfp_field.cuh"

namespace lfp {
static inline size_t cdiv(size_t a, size_t b) { return (a + b - 1) / b; }
__device__ __forceinline__ int exp_of(int8_t d) { return d >= 0 ? d : 16 + d; }

// ---- tables of the sumcheck ----------------------------------------------------------------------------------------------------------
// tab[2 col][row] = beta^e, tab[2 col + 1][row] = beta^(2e) (Montgomery), e the exponent of dig[row][col]; 0 for an absent entry
__global__ void __launch_bounds__(256) k_sc_tables(const int8_t *dig, size_t n, u32 ncols, PwTab pw, u64 *tab, size_t ld) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * ncols) return;
    const size_t row = i / ncols;
    const u32 col = (u32)(i % ncols);
    const int8_t d = dig[i];
    u64 m = 0, q = 0;
    if (d != LFP_ABSENT) {
        const int e = exp_of(d);
#pragma unroll
        for (int t = 0; t < 16; t++) { m = e == t ? pw.p[t] : m; q = e == t ? pw.q[t] : q; }
    }
    tab[(size_t)(2 * col) * ld + row] = m;
    tab[(size_t)(2 * col + 1) * ld + row] = q;
}
void launch_sc_tables(const int8_t *dig, size_t n, u32 ncols, const PwTab &pw, u64 *tab, size_t ld, hipStream_t s) {
    hipLaunchKernelGGL(k_sc_tables, dim3((unsigned)cdiv(n * ncols, 256)), dim3(256), 0, s, dig, n, ncols, pw, tab, ld);
}
// eq[i] = prod_j (bit_j(i) ? c_j : 1 - c_j) (Montgomery; bit 0 <-> c[0]: build_eq_x_r)
__global__ void __launch_bounds__(256) k_eq_build(EqPt pt, u32 nv, size_t n, u64 *eq) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 v = pt.one;
    for (u32 j = 0; j < nv; j++) v = mont_mul(v, (i >> j) & 1 ? pt.c[j] : pt.nc[j]);
    eq[i] = v;
}
void launch_eq_build(const EqPt &pt, u32 nv, u64 *eq, hipStream_t s) {
    const size_t n = (size_t)1 << nv;
    hipLaunchKernelGGL(k_eq_build, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, pt, nv, n, eq);
}

// sum over the block of four words per thread -> part[block][4] (mod p)
__device__ __forceinline__ void block_sum4(u64 s[4], u64 *out) {
    __shared__ u64 sm[4][4];
#pragma unroll
    for (int x = 0; x < 4; x++)
        for (int o = 32; o; o >>= 1) s[x] = add_p(s[x], __shfl_xor(s[x], o));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int x = 0; x < 4; x++) sm[wave][x] = s[x];
    __syncthreads();
    if (threadIdx.x < 4) out[threadIdx.x] = add_p(add_p(sm[0][threadIdx.x], sm[1][threadIdx.x]), add_p(sm[2][threadIdx.x], sm[3][threadIdx.x]));
}
// One round of the set-check sumcheck (prove_round, sumcheck/prover.rs:56-162, with the comb_fn of setchk.rs:160-197): for every pair of
// entries the degree-3 round polynomial at X = 0..3,
//     sum over the sets i entering the polynomial of  eq_i(X) * sum_j coef[i][j] (m_ij(X)^2 - m'_ij(X)),
// coef[i][j] = rc^i alpha_i^j (a vector set: rc^i alpha_i).  part[block][4], Montgomery.
// thread = (pair, set, chunk of cc columns): the late rounds have few pairs and 66 columns each -- one thread per pair would walk them all
// (~2000 dependent Montgomery products, 60 us whatever the size); sums mod p are exact, the split changes no word of the message
__global__ void __launch_bounds__(256) k_sc_round(const u64 *tab, size_t ld, size_t half, ScDesc d, const u64 *coef, u64 *part, u32 cc) {
    u64 s[4] = {0, 0, 0, 0};
    const u32 mats_eff = d.nsets_eff < d.nmat ? d.nsets_eff : d.nmat, vec_eff = d.nsets_eff - mats_eff, cpm = (d.ncols + cc - 1) / cc;
    const size_t nchunks = (size_t)mats_eff * cpm + vec_eff, total = half * nchunks;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t b = idx % half;
        const u32 ch = (u32)(idx / half);
        u32 i, j0, j1;
        if (ch < mats_eff * cpm) { i = ch / cpm; j0 = (ch % cpm) * cc; j1 = j0 + cc < d.ncols ? j0 + cc : d.ncols; }
        else { i = d.nmat + (ch - mats_eff * cpm); j0 = 0; j1 = 1; }
        const u32 cols = i < d.nmat ? d.ncols : 1;
        const u32 t0 = i < d.nmat ? i * (2 * d.ncols + 1) : d.nmat * (2 * d.ncols + 1) + 3 * (i - d.nmat);
        u64 in[4] = {0, 0, 0, 0};
        for (u32 j = j0; j < j1; j++) {
            const u64 *tm = tab + (size_t)(t0 + 2 * j) * ld + 2 * b, *tq = tm + ld;
            u64 m = tm[0], q = tq[0];
            const u64 dm = sub_p(tm[1], m), dq = sub_p(tq[1], q), cf = coef[(size_t)i * d.ncols + j];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                in[x] = add_p(in[x], mont_mul(cf, sub_p(mont_mul(m, m), q)));
                m = add_p(m, dm);
                q = add_p(q, dq);
            }
        }
        const u64 *te = tab + (size_t)(t0 + 2 * cols) * ld + 2 * b;
        u64 e = te[0];
        const u64 de = sub_p(te[1], e);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            s[x] = add_p(s[x], mont_mul(e, in[x]));
            e = add_p(e, de);
        }
    }
    block_sum4(s, part + (size_t)blockIdx.x * 4);
}
u32 sc_round_max_blocks() {      // (4 words of partial sums per block for the host: the cap is about occupancy; LFPLUS_SC_BLOCKS moves it)
    constexpr u32 cap = 2048;
    return cap;
}
// returns the number of blocks = rows of part[.][4]
u32 launch_sc_round(const u64 *tab, size_t ld, size_t half, const ScDesc &d, const u64 *coef, u64 *part, hipStream_t s) {
    const u32 mats_eff = d.nsets_eff < d.nmat ? d.nsets_eff : d.nmat, vec_eff = d.nsets_eff - mats_eff;
    // whole sets per thread while that still fills the chip (2 x 256 threads per CU), else chunks of 4 columns, else single columns
    u32 cc = d.ncols;
    if (half * (mats_eff + vec_eff) < ((size_t)1 << 17)) cc = 4;
    if (half * ((size_t)mats_eff * ((d.ncols + 3) / 4) + vec_eff) < ((size_t)1 << 17)) cc = 1;
    if (cc > d.ncols) cc = d.ncols;
    const size_t total = half * ((size_t)mats_eff * ((d.ncols + cc - 1) / cc) + vec_eff);
    size_t nb = cdiv(total, 256);
    if (nb < 1) nb = 1;
    if (nb > sc_round_max_blocks()) nb = sc_round_max_blocks();
    hipLaunchKernelGGL(k_sc_round, dim3((unsigned)nb), dim3(256), 0, s, tab, ld, half, d, coef, part, cc);
    return (u32)nb;
}
// fix_variables of all tables: out[t][b] = in[t][2b] + r (in[t][2b+1] - in[t][2b])
__global__ void __launch_bounds__(256) k_sc_fix(const u64 *in, u64 *out, size_t ld, size_t half, u64 rM) {
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= half) return;
    const u64 *p = in + (size_t)blockIdx.y * ld + 2 * b;
    const u64 lo = p[0], hi = p[1];
    out[(size_t)blockIdx.y * ld + b] = add_p(lo, mont_mul(rM, sub_p(hi, lo)));
}
void launch_sc_fix(const u64 *in, u64 *out, size_t ld, u32 ntab, size_t half, u64 rM, hipStream_t s) {
    hipLaunchKernelGGL(k_sc_fix, dim3((unsigned)cdiv(half, 256), ntab), dim3(256), 0, s, in, out, ld, half, rM);
}

// ---- the first two rounds straight from the exponent digits ------------------------------------------------------------------------------
// A set's 2 ncols scalar tables beta^e, beta^2e are look-ups of one byte per entry: materialising them (k_sc_tables: 33 tables of n words per matrix set,
// 3.3 GB at n = 2^20 with 12 sets) and streaming them through round 0, the first fix and round 1 was 10 GB of traffic for 0.2 GB of digits.  Round 0 and the
// fused "fix at r_0 + round 1" read the digits instead (17-entry power table in LDS, entry 16 = the absent monomial); the first tables that exist are the
// n/2-entry ones the fused pass writes.  Same field elements: every sum is exact mod p.
struct DigPow {
    u64 p[17], q[17];
};
__device__ __forceinline__ void digpow_load(const PwTab &pw, DigPow *sm) {
    if (threadIdx.x < 16) { sm->p[threadIdx.x] = pw.p[threadIdx.x]; sm->q[threadIdx.x] = pw.q[threadIdx.x]; }
    if (threadIdx.x == 16) { sm->p[16] = 0; sm->q[16] = 0; }
    __syncthreads();
}
__device__ __forceinline__ u32 dig_index(int8_t d) { return d == LFP_ABSENT ? 16u : (u32)exp_of(d); }
// the bytes of row `row`, columns j0 .. j0 + 15 (fewer at the end of a narrow set), as exponent indices
template <int NC>
__device__ __forceinline__ void dig_row(const int8_t *dig, size_t row, u32 (&e)[NC]) {
    if (NC == 16) {
        const uint4 w = *(const uint4 *)(dig + row * 16);
        const u32 ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 16; j++) e[j] = dig_index((int8_t)(ww[j >> 2] >> (8 * (j & 3))));
    } else {
#pragma unroll
        for (int j = 0; j < NC; j++) e[j] = dig_index(dig[row * NC + j]);
    }
}
// one set's share of round 0: part[block][4] = sum over the block's pairs of eq(X) sum_j coef[j] (m_j(X)^2 - q_j(X)), X = 0..3
template <int NC>
__global__ void __launch_bounds__(256) k_sc_round0_dig(const int8_t *dig, size_t half, PwTab pw, const u64 *eq, const u64 *coef, u64 *part) {
    __shared__ DigPow T;
    __shared__ u64 cf[NC];
    if (threadIdx.x < NC) cf[threadIdx.x] = coef[threadIdx.x];
    digpow_load(pw, &T);
    u64 s[4] = {0, 0, 0, 0};
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < half; b += (size_t)gridDim.x * 256) {
        u32 e0[NC], e1[NC];
        dig_row<NC>(dig, 2 * b, e0);
        dig_row<NC>(dig, 2 * b + 1, e1);
        u64 in[4] = {0, 0, 0, 0};
#pragma unroll 4
        for (int j = 0; j < NC; j++) {
            u64 m = T.p[e0[j]], q = T.q[e0[j]];
            const u64 dm = sub_p(T.p[e1[j]], m), dq = sub_p(T.q[e1[j]], q), c = cf[j];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                in[x] = add_p(in[x], mont_mul(c, sub_p(mont_mul(m, m), q)));
                m = add_p(m, dm);
                q = add_p(q, dq);
            }
        }
        const ulonglong2 ee = *(const ulonglong2 *)(eq + 2 * b);
        u64 ev = ee.x;
        const u64 de = sub_p(ee.y, ev);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            s[x] = add_p(s[x], mont_mul(ev, in[x]));
            ev = add_p(ev, de);
        }
    }
    block_sum4(s, part + (size_t)blockIdx.x * 4);
}
u32 launch_sc_round0_dig(const int8_t *dig, size_t n, u32 ncols, const PwTab &pw, const u64 *eq, const u64 *coef, u64 *part, hipStream_t s) {
    const size_t half = n / 2;
    size_t nb = cdiv(half, 256);
    if (nb < 1) nb = 1;
    if (nb > sc_round_max_blocks()) nb = sc_round_max_blocks();
    if (ncols == 16) hipLaunchKernelGGL((k_sc_round0_dig<16>), dim3((unsigned)nb), dim3(256), 0, s, dig, half, pw, eq, coef, part);
    else if (ncols == 1) hipLaunchKernelGGL((k_sc_round0_dig<1>), dim3((unsigned)nb), dim3(256), 0, s, dig, half, pw, eq, coef, part);
    else return 0;
    return (u32)nb;
}
// fix_variables at r of the set's (virtual) n-entry tables -> its 2 NC + 1 tables of n / 2 entries (written: tab[t][.], stride ld), and the set's share of the
// NEXT round from those values: thread = four consecutive rows = one pair of the fixed tables
template <int NC>
__global__ void __launch_bounds__(256) k_sc_fix_round_dig(const int8_t *dig, size_t quarter, PwTab pw, const u64 *eq, u64 rM, u64 *tab, size_t ld, const u64 *coef,
                                                          u64 *part) {
    __shared__ DigPow T;
    __shared__ u64 cf[NC];
    if (threadIdx.x < NC) cf[threadIdx.x] = coef[threadIdx.x];
    digpow_load(pw, &T);
    u64 s[4] = {0, 0, 0, 0};
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < quarter; c += (size_t)gridDim.x * 256) {
        u32 e0[NC], e1[NC], e2[NC], e3[NC];
        dig_row<NC>(dig, 4 * c, e0);
        dig_row<NC>(dig, 4 * c + 1, e1);
        dig_row<NC>(dig, 4 * c + 2, e2);
        dig_row<NC>(dig, 4 * c + 3, e3);
        u64 in[4] = {0, 0, 0, 0};
#pragma unroll 2
        for (int j = 0; j < NC; j++) {
            const u64 ma = T.p[e0[j]], mc = T.p[e2[j]], qa = T.q[e0[j]], qc = T.q[e2[j]];
            const u64 m0 = add_p(ma, mont_mul(rM, sub_p(T.p[e1[j]], ma))), m1 = add_p(mc, mont_mul(rM, sub_p(T.p[e3[j]], mc)));
            const u64 q0 = add_p(qa, mont_mul(rM, sub_p(T.q[e1[j]], qa))), q1 = add_p(qc, mont_mul(rM, sub_p(T.q[e3[j]], qc)));
            *(ulonglong2 *)(tab + (size_t)(2 * j) * ld + 2 * c) = ulonglong2{m0, m1};
            *(ulonglong2 *)(tab + (size_t)(2 * j + 1) * ld + 2 * c) = ulonglong2{q0, q1};
            u64 m = m0, q = q0;
            const u64 dm = sub_p(m1, m0), dq = sub_p(q1, q0), cc = cf[j];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                in[x] = add_p(in[x], mont_mul(cc, sub_p(mont_mul(m, m), q)));
                m = add_p(m, dm);
                q = add_p(q, dq);
            }
        }
        const ulonglong2 ea = *(const ulonglong2 *)(eq + 4 * c), eb = *(const ulonglong2 *)(eq + 4 * c + 2);
        const u64 f0 = add_p(ea.x, mont_mul(rM, sub_p(ea.y, ea.x))), f1 = add_p(eb.x, mont_mul(rM, sub_p(eb.y, eb.x)));
        *(ulonglong2 *)(tab + (size_t)(2 * NC) * ld + 2 * c) = ulonglong2{f0, f1};
        u64 ev = f0;
        const u64 de = sub_p(f1, f0);
#pragma unroll
        for (int x = 0; x < 4; x++) {
            s[x] = add_p(s[x], mont_mul(ev, in[x]));
            ev = add_p(ev, de);
        }
    }
    block_sum4(s, part + (size_t)blockIdx.x * 4);
}
u32 launch_sc_fix_round_dig(const int8_t *dig, size_t n, u32 ncols, const PwTab &pw, const u64 *eq, u64 rM, u64 *tab, size_t ld, const u64 *coef, u64 *part,
                            hipStream_t s) {
    const size_t quarter = n / 4;
    size_t nb = cdiv(quarter, 256);
    if (nb < 1) nb = 1;
    if (nb > sc_round_max_blocks()) nb = sc_round_max_blocks();
    if (ncols == 16) hipLaunchKernelGGL((k_sc_fix_round_dig<16>), dim3((unsigned)nb), dim3(256), 0, s, dig, quarter, pw, eq, rM, tab, ld, coef, part);
    else if (ncols == 1) hipLaunchKernelGGL((k_sc_fix_round_dig<1>), dim3((unsigned)nb), dim3(256), 0, s, dig, quarter, pw, eq, rM, tab, ld, coef, part);
    else return 0;
    return (u32)nb;
}

// ---- evaluations ---------------------------------------------------------------------------------------------------------------------
// part[chunk][col][16] = sum over the chunk's rows of w[row] * X^e(dig[row][col]).  wstride 1: scalar weights (constant polynomials);
// 16: ring weights -- coefficient t of w X^e is w[t - e] (t >= e), -w[t - e + 16] (t < e).
// 16 lanes per row (lane t = coefficient t), ALL NC columns of the row per pass: the weight row is loaded once, coalesced (lane t reads word t), and the
// rotation by the exponent is a 16-lane shuffle; a lane keeps one accumulator per column.  (First version: thread = row, grid.y = column -- every thread
// gathered its 16 weight words with a data-dependent index from 64 different cache lines per instruction, and the 128 MB weight table of a 2^20-row
// instance was read once per column: 39 ms of the 167 ms prove were the 48 evaluation passes of the set check.)
// (sums are kept as lazy 96-bit integers: a subtraction adds p - v, one reduction per accumulator at the end -- add_p AND sub_p per item, 13 of its 17 integer
// instructions, were what the pass spent its time on; at most 2^20 rows per lane: the sums stay below 2^84)
struct Acc96 { u32 a0, a1, a2; };
__device__ __forceinline__ void acc96_add(Acc96 &a, u64 v) {
    const u32 v0 = (u32)v, v1 = (u32)(v >> 32);
    asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc"
        : "+v"(a.a0), "+v"(a.a1), "+v"(a.a2)
        : "v"(v0), "v"(v1)
        : "vcc");
}
__device__ __forceinline__ u64 acc96_red(const Acc96 &a) {   // (a2 2^64 + lo) mod p
    u64 lo = ((u64)a.a1 << 32) | a.a0;
    if (lo >= P) lo -= P;
    return add_p(mont_mul((u64)a.a2, R2), lo);
}
// (the rotation as a barrel shifter of four DPP row rotations instead of the 16-lane shuffle: correct, and 16.2 instead of 10.2 ms for the 48 passes at 2^20 rows)
template <int NC>
__global__ void __launch_bounds__(256) k_wmono(const int8_t *dig, size_t dstride, size_t n, const u64 *w, u32 wstride, u64 *part, u32 pcols, u32 pc0) {
    const int t = threadIdx.x & 15, r = threadIdx.x >> 4;
    Acc96 acc[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] = Acc96{0, 0, 0};
    for (size_t row = (size_t)blockIdx.x * 16 + r; row < n; row += (size_t)gridDim.x * 16) {
        const u64 wv = wstride == 1 ? w[row] : w[row * 16 + t];
        int8_t d[NC];
        if (NC == 16) {
            const uint4 dv = *(const uint4 *)(dig + row * 16);
            const u32 dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int c = 0; c < NC; c++) d[c] = (int8_t)(dw[c >> 2] >> (8 * (c & 3)));
        } else {
#pragma unroll
            for (int c = 0; c < NC; c++) d[c] = dig[row * dstride + c];
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int e = exp_of(d[c]);
            if (wstride == 1) {
                acc96_add(acc[c], (d[c] != LFP_ABSENT && e == t) ? wv : 0);
            } else {
                const u64 v = __shfl(wv, (t - e) & 15, 16);
                const u64 vs = t >= e ? v : (v ? P - v : 0);      // - v mod p
                acc96_add(acc[c], d[c] != LFP_ABSENT ? vs : 0);
            }
        }
    }
    // the 16 row groups of the block -> one sum per (column, coefficient)
    __shared__ u64 sm[16][NC][16];
#pragma unroll
    for (int c = 0; c < NC; c++) sm[r][c][t] = acc96_red(acc[c]);
    __syncthreads();
    for (int o = threadIdx.x; o < NC * 16; o += 256) {
        const int c = o >> 4, tt = o & 15;
        u64 sacc = 0;
#pragma unroll
        for (int g = 0; g < 16; g++) sacc = add_p(sacc, sm[g][c][tt]);
        part[((size_t)blockIdx.x * pcols + pc0 + c) * 16 + tt] = sacc;
    }
}
// part[chunk][16] = sum over the chunk's rows of w[row] * f[row] (f: n ring elements, canonical).  wstride 1: scalar weights in Montgomery
// form (eq tables); 16: ring weights, canonical (negacyclic products).  thread = (row lane, coefficient): a wave reads 512 contiguous bytes of f per step; the
// terms are lazy 160-bit sums over ALL the rows of the thread (acc160_*), one reduction at the end.  Ring weights: coefficient t of w f = sum_j w[j] f[t - j]
// (j <= t), - w[j] f[t - j + 16] (j > t): signed terms in two's complement, made non-negative by the multiple rows * 16 p 2^64 of p 2^64 added at the end; both
// operands canonical, so the reduced sum (x 2^-64) goes back through to_mont.  (First version: thread = row, 16 x 16 mont_mul + add_p / sub_p per row: 0.25 ms per
// pass at 2^20 rows.)
template <bool RINGW>
__global__ void __launch_bounds__(256) k_wring(const u64 *f, size_t n, const u64 *w, u64 *part) {
    const u32 c = threadIdx.x & 15, rl = threadIdx.x >> 4;
    __shared__ u64 lw[16][16], lf[16][16];
    Acc160 acc;
    acc160_zero(acc);
    u64 rows = 0;
    for (size_t base = (size_t)blockIdx.x * 16; base < n; base += (size_t)gridDim.x * 16) {
        const size_t row = base + rl;
        const bool ok = row < n;
        if (!RINGW) {
            if (ok) acc160_mad(acc, w[row], f[row * 16 + c]);
        } else {
            lw[rl][c] = ok ? w[row * 16 + c] : 0;
            lf[rl][c] = ok ? f[row * 16 + c] : 0;
            __syncthreads();
#pragma unroll
            for (u32 j = 0; j < 16; j++) acc160_mad_signed(acc, lw[rl][j], lf[rl][(c - j) & 15], j > c);
            rows++;
            __syncthreads();
        }
    }
    u64 v;
    if (RINGW) {      // + rows * 16 p * 2^64, then (sum 2^-64) 2^64
        const u64 x = rows << 4, lo = x * P, hi = __umul64hi(x, P);
        asm("v_add_co_u32 %0, vcc, %0, %3\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\tv_addc_co_u32 %2, vcc, %2, %5, vcc"
            : "+v"(acc.a[2]), "+v"(acc.a[3]), "+v"(acc.a[4])
            : "v"((u32)lo), "v"((u32)(lo >> 32)), "v"((u32)hi)
            : "vcc");
        v = to_mont(acc160_red(acc));
    } else
        v = acc160_red(acc);
    __shared__ u64 sm[16][16];
    sm[rl][c] = v;
    __syncthreads();
    if (threadIdx.x < 16) {
        u64 t = 0;
        for (int p = 0; p < 16; p++) t = add_p(t, sm[p][threadIdx.x]);
        part[(size_t)blockIdx.x * 16 + threadIdx.x] = t;
    }
}
// part[chunk] = sum of x[i * xstride] * y[i]; x Montgomery (x_mont) or canonical, y canonical: canonical sum
__global__ void __launch_bounds__(256) k_wdot(const u64 *x, u32 xstride, int x_mont, const u64 *y, size_t n, u64 *part) {
    u64 s[4] = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u64 xv = x[i * xstride];
        s[0] = add_p(s[0], mont_mul(x_mont ? xv : to_mont(xv), y[i]));
    }
    block_sum4(s, part + (size_t)blockIdx.x * 4);
}
// out[o] = sum over chunks of part[chunk * stride + o] (o < nout), optionally taken out of Montgomery form.
// block = 32 outputs x 32 chunk lanes (the first version walked the chunks with one thread per output: 256 - 1024 dependent loads, 0.25 ms per call and a
// fifth of a 2^20-row prove in 130 calls)
__global__ void __launch_bounds__(1024) k_sum_parts(const u64 *part, u32 chunks, size_t stride, u32 nout, int out_of_mont, u64 *out) {
    __shared__ u64 sm[32][33];
    const u32 ol = threadIdx.x & 31, cl = threadIdx.x >> 5, o = blockIdx.x * 32 + ol;
    u64 s = 0;
    if (o < nout)
        for (u32 ch = cl; ch < chunks; ch += 32) s = add_p(s, part[(size_t)ch * stride + o]);
    sm[cl][ol] = s;
    __syncthreads();
    if (cl == 0 && o < nout) {
#pragma unroll
        for (int g = 1; g < 32; g++) s = add_p(s, sm[g][ol]);
        out[o] = out_of_mont ? from_mont(s) : s;
    }
}
// Scalar weights, 16 columns, up to four weight tables at once: sum_row w_q[row] X^e(dig[row][col]) is a HISTOGRAM over the exponent -- bucket (col, e) collects the
// weights of the rows whose entry is X^e.  thread = (row, column); a bucket is two 64-bit LDS counters (sums of the low and of the high 32-bit halves of the
// weights: exact for 2^32 rows) fed by ds_add_u64; the digits are decoded once for all the weight tables.  The lane-per-coefficient form above (k_wmono, wstride 1)
// has 16 lanes per row of which one adds a non-zero word per column: 0.14 ms per pass and weight table at 2^20 rows, 48 passes per set check.
// part[q][chunk][col][16], canonical residues of the (Montgomery-form) sums; LDS layout [q][e][col] with 17 columns: the 16 columns of a row fall into distinct banks,
// the four rows of a wave are shifted against each other
struct WPtrs { const u64 *w[4]; };
__global__ void __launch_bounds__(256) k_whist16(const int8_t *dig, size_t n, WPtrs wp, u32 nw, u64 *part, size_t pstride) {
    __shared__ unsigned long long h[4][16][17][2];
    for (u32 i = threadIdx.x; i < 4 * 16 * 17 * 2; i += 256) (&h[0][0][0][0])[i] = 0;
    __syncthreads();
    const u32 col = threadIdx.x & 15, rl = threadIdx.x >> 4;
#pragma unroll 2
    for (size_t row = (size_t)blockIdx.x * 16 + rl; row < n; row += (size_t)gridDim.x * 16) {
        const int8_t d = dig[row * 16 + col];
        if (d == LFP_ABSENT) continue;
        const int e = exp_of(d);
        for (u32 q = 0; q < nw; q++) {
            const u64 v = wp.w[q][row];
            (void)__hip_atomic_fetch_add(&h[q][e][col][0], (unsigned long long)(u32)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            (void)__hip_atomic_fetch_add(&h[q][e][col][1], (unsigned long long)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const u32 t = threadIdx.x & 15, c = threadIdx.x >> 4;
    for (u32 q = 0; q < nw; q++) {
        const unsigned __int128 T = (unsigned __int128)h[q][t][c][0] + ((unsigned __int128)h[q][t][c][1] << 32);
        u64 lo = (u64)T;
        if (lo >= P) lo -= P;
        part[q * pstride + ((size_t)blockIdx.x * 16 + c) * 16 + t] = add_p(mont_mul((u64)(T >> 64), R2), lo);
    }
}
// out[q * ostride + o] = sum over chunks of part[q * pstride + chunk * nout + o], taken out of Montgomery form (k_sum_parts with a weight-table index in grid.y)
__global__ void __launch_bounds__(1024) k_sum_parts_q(const u64 *part, u32 chunks, size_t pstride, u32 nout, u64 *out, size_t ostride) {
    __shared__ u64 sm[32][33];
    const u32 ol = threadIdx.x & 31, cl = threadIdx.x >> 5, o = blockIdx.x * 32 + ol;
    part += (size_t)blockIdx.y * pstride;
    u64 s = 0;
    if (o < nout)
        for (u32 ch = cl; ch < chunks; ch += 32) s = add_p(s, part[(size_t)ch * nout + o]);
    sm[cl][ol] = s;
    __syncthreads();
    if (cl == 0 && o < nout) {
#pragma unroll
        for (int g = 1; g < 32; g++) s = add_p(s, sm[g][ol]);
        out[(size_t)blockIdx.y * ostride + o] = from_mont(s);
    }
}
// (256 blocks = one wave per SIMD: the passes were latency-bound; LFPLUS_EVAL_CHUNKS moves the cap)
u32 eval_chunks(size_t n) {
    constexpr size_t cap = 1024;
    size_t b = cdiv(n, 256);
    return (u32)(b < 1 ? 1 : (b > cap ? cap : b));
}
// out[o] = sum over `chunks` rows of part[chunk * stride + o], o < nout (words kept in the form they have)
void launch_sum_parts(const u64 *part, u32 chunks, size_t stride, u32 nout, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)cdiv((size_t)nout, 32)), dim3(1024), 0, s, part, chunks, stride, nout, 0, out);
}
void launch_wmono(const int8_t *dig, size_t n, u32 ncols, const u64 *w, u32 wstride, u64 *part, u64 *out, hipStream_t s) {
    const u32 ch = eval_chunks(n);
    if (ncols == 16) hipLaunchKernelGGL((k_wmono<16>), dim3(ch), dim3(256), 0, s, dig, (size_t)16, n, w, wstride, part, 16u, 0u);
    else {   // other widths: one column at a time (the range check's vector sets have one)
        for (u32 c0 = 0; c0 < ncols; c0++) hipLaunchKernelGGL((k_wmono<1>), dim3(ch), dim3(256), 0, s, dig + c0, (size_t)ncols, n, w, wstride, part, ncols, c0);
    }
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)cdiv((size_t)ncols * 16, 32)), dim3(1024), 0, s, part, ch, (size_t)ncols * 16, ncols * 16, wstride == 1, out);
}
// the 16 columns of a monomial matrix against nw <= 4 scalar weight tables (Montgomery): out[q * ostride + col * 16 + t]; part: nw * eval_chunks(n) * 256 words
void launch_whist16(const int8_t *dig, size_t n, const u64 *const *w, u32 nw, u64 *part, u64 *out, size_t ostride, hipStream_t s) {
    const u32 ch = eval_chunks(n);
    WPtrs wp = {{nullptr, nullptr, nullptr, nullptr}};
    for (u32 q = 0; q < nw; q++) wp.w[q] = w[q];
    hipLaunchKernelGGL(k_whist16, dim3(ch), dim3(256), 0, s, dig, n, wp, nw, part, (size_t)ch * 256);
    hipLaunchKernelGGL(k_sum_parts_q, dim3(8, nw), dim3(1024), 0, s, part, ch, (size_t)ch * 256, 256u, out, ostride);
}
void launch_wring(const u64 *f, size_t n, const u64 *w, u32 wstride, u64 *part, u64 *out, hipStream_t s) {
    const u32 ch = eval_chunks(n);
    if (wstride == 1) hipLaunchKernelGGL((k_wring<false>), dim3(ch), dim3(256), 0, s, f, n, w, part);
    else hipLaunchKernelGGL((k_wring<true>), dim3(ch), dim3(256), 0, s, f, n, w, part);
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1024), 0, s, part, ch, (size_t)16, 16u, 0, out);
}
void launch_wdot(const u64 *x, u32 xstride, int x_mont, const u64 *y, size_t n, u64 *part, u64 *out, hipStream_t s) {
    const u32 ch = eval_chunks(n);
    hipLaunchKernelGGL(k_wdot, dim3(ch), dim3(256), 0, s, x, xstride, x_mont, y, n, part);
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1024), 0, s, part, ch, (size_t)4, 1u, 0, out);
}
// w[c] = sum over the non-zeros (row, c) of M of M[row][c] * eq[row]: the transposed matrix in CSR form (colptr over c, rowidx, values
// canonical), eq in Montgomery form -> w canonical ring elements.  thread = (c, coefficient)
__global__ void __launch_bounds__(256) k_spmvT_eq(const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t n, u64 *w) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 16) return;
    const size_t c = i >> 4;
    const u32 t = (u32)(i & 15);
    u64 s = 0;
    for (u32 k = colptr[c]; k < colptr[c + 1]; k++) s = add_p(s, mont_mul(eq[rowidx[k]], val[(size_t)k * 16 + t]));
    w[i] = s;
}
// constant-coefficient matrices: w[c] is a constant polynomial -- its coefficient 0 as ONE scalar, in Montgomery form like eq itself
__global__ void __launch_bounds__(256) k_spmvT_eq_const(const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t n, u64 *w) {
    const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    u64 s = 0;
    for (u32 k = colptr[c]; k < colptr[c + 1]; k++) s = add_p(s, mont_mul(eq[rowidx[k]], val[k]));      // val: one word per non-zero (LfpMatrix::valTc)
    w[c] = to_mont(s);
}
void launch_spmvT_eq_const(const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t n, u64 *w, hipStream_t s) {
    hipLaunchKernelGGL(k_spmvT_eq_const, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, colptr, rowidx, val, eq, n, w);
}
void launch_spmvT_eq(const u32 *colptr, const u32 *rowidx, const u64 *val, const u64 *eq, size_t n, u64 *w, hipStream_t s) {
    hipLaunchKernelGGL(k_spmvT_eq, dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, s, colptr, rowidx, val, eq, n, w);
}
// tau (n canonical words) as n ring constants is never materialised: the M_i tau row needs sum_c w[c][0] tau[c] only (the constant term)
}  // namespace lfp

// ---- Cm::prove (cm.rs:56-347) -------------------------------------------------------------------------------------------------------------
namespace lfp {
// ring tables from the compact forms: out[row] = X^e(dig[row]) (mono != 0) or the constant tau[row]
__global__ void __launch_bounds__(256) k_cm_materialize(const int8_t *dig, const u64 *tau, size_t n, u64 *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 16) return;
    const size_t row = i >> 4;
    const int t = (int)(i & 15);
    out[i] = dig ? (u64)(exp_of(dig[row]) == t) : (t == 0 ? tau[row] : 0);
}
void launch_cm_materialize(const int8_t *dig, const u64 *tau, size_t n, u64 *out, hipStream_t s) {
    hipLaunchKernelGGL(k_cm_materialize, dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, s, dig, tau, n, out);
}
__global__ void __launch_bounds__(256) k_to_mont(const u64 *in, size_t n, u64 *out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = to_mont(in[i]);
}
void launch_to_mont(const u64 *in, size_t n, u64 *out, hipStream_t s) { hipLaunchKernelGGL(k_to_mont, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, in, n, out); }
// h[row] = sum_{ki < k} sum_{j < 16} X^e(Df[ki][row][j]) * s'[ki][j]  (cm.rs:82-103): rotations of the short challenges (|coefficients| <= 128), exact
// in int32; thread = (row, coefficient).  The LDS copy of s'[ki][j] is the 32-entry negacyclic extension ext[m] = -v[m] (m < 16), v[m - 16] (m >= 16): coefficient
// t of X^e v is ext[16 + t - e], one read without the wrap test; a row's 16 digits come in one 16-byte load and exp(d) = d & 15 for the digits in (-8, 8)
// (first version: 16 byte loads per row and ki, index mask, compare and negate per term -- 0.33 ms per instance at 2^20 rows and k = 4)
__global__ void __launch_bounds__(256) k_cm_h(const int8_t *Df, size_t n, u32 k, const int32_t *sp /* [k][16][16] */, u64 *h) {
    __shared__ int32_t ext[16 * 16 * 32];
    for (u32 i = threadIdx.x; i < k * 512; i += 256) {
        const u32 kj = i >> 5, m = i & 31;
        const int32_t v = sp[kj * 16 + (m & 15)];
        ext[i] = m < 16 ? -v : v;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 16) return;
    const size_t row = i >> 4;
    const int t = (int)(i & 15);
    int acc = 0;
    for (u32 ki = 0; ki < k; ki++) {
        const uint4 dv = *(const uint4 *)(Df + ((size_t)ki * n + row) * 16);
        const u32 dw[4] = {dv.x, dv.y, dv.z, dv.w};
        const int32_t *ek = ext + ki * 512 + 16 + t;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int e = (int)((dw[j >> 2] >> (8 * (j & 3))) & 15u);
            acc += ek[j * 32 - e];
        }
    }
    h[i] = acc >= 0 ? (u64)acc : P - (u64)(-acc);
}
void launch_cm_h(const int8_t *Df, size_t n, u32 k, const int32_t *sp, u64 *h, hipStream_t s) {
    hipLaunchKernelGGL(k_cm_h, dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, s, Df, n, k, sp, h);
}
// g[row] = s0 tau[row] + s1 X^e(mtau[row]) + s2 f[row] + h[row]  (cm.rs:164-181); s: three short challenges as int32 [3][16].
// The challenges are small signed integers: |s| * f < 2^95, so the 17 products of a coefficient (s0[t] tau and the 16 terms of the negacyclic s2 f) are summed as
// plain 128-bit integers, positive and negative terms apart, and reduced once each ((hi 2^64 + lo) mod p = hi 2^64 + lo: mont_mul(hi, 2^128) + lo).
// (First version: mul_p = two Montgomery products per term, 34 per coefficient: 0.59 ms per instance at 2^20 rows against 0.1 ms of traffic.)
__device__ __forceinline__ u64 red128(unsigned __int128 x) {      // x < 2^100
    u64 lo = (u64)x;
    if (lo >= P) lo -= P;
    return add_p(mont_mul((u64)(x >> 64), R2), lo);
}
__global__ void __launch_bounds__(256) k_cm_g(const u64 *tau, const int8_t *mtau, const u64 *f, const u64 *h, size_t n, CmShort s, u64 *g) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 16) return;
    const size_t row = i >> 4;
    const int t = (int)(i & 15);
    auto fe = [](int v) { return v >= 0 ? (u64)v : P - (u64)(-v); };
    unsigned __int128 pos = 0, neg = 0;
    {
        const int s0 = s.v[0][t];
        const unsigned __int128 pr = (unsigned __int128)(u64)(s0 < 0 ? -(long long)s0 : (long long)s0) * tau[row];
        pos += s0 < 0 ? 0 : pr;
        neg += s0 < 0 ? pr : 0;
    }
    const u64 *fr = f + row * 16;
#pragma unroll
    for (int j = 0; j < 16; j++) {          // s2[j] X^j * f: coefficient t gets s2[j] f[t - j] (t >= j), - s2[j] f[t - j + 16]
        const int sv = s.v[2][j];
        const bool ng = (sv < 0) != (t < j);
        const unsigned __int128 pr = (unsigned __int128)(u64)(sv < 0 ? -(long long)sv : (long long)sv) * fr[(t - j) & 15];
        pos += ng ? 0 : pr;
        neg += ng ? pr : 0;
    }
    u64 acc = add_p(h[i], sub_p(red128(pos), red128(neg)));
    const int e = exp_of(mtau[row]), r1 = s.v[1][(t - e) & 15];
    acc = add_p(acc, fe(t >= e ? r1 : -r1));
    g[i] = acc;
}
void launch_cm_g(const u64 *tau, const int8_t *mtau, const u64 *f, const u64 *h, size_t n, const CmShort &s, u64 *g, hipStream_t st) {
    hipLaunchKernelGGL(k_cm_g, dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, st, tau, mtau, f, h, n, s, g);
}
// One round of a sumchecker of Cm::prove (cm.rs:201-347; comb_fn :287-311), degree 2.  Scalar tables (Montgomery): S[0] = eq(r, .), S[1 + l] = tau_l;
// ring tables (canonical, [entry][16]): R[l * (per - 1) + j - 1] = table j >= 1 of instance l in the reference's order (m_tau, f, h, then per matrix
// M tau, M m_tau, M f, M h), R[L (per - 1)] = t0, R[L (per - 1) + 1] = t1.  rcp[i] = rc^i (Montgomery).  thread = (pair, coefficient);
// part[block][3][16] canonical.
__global__ void __launch_bounds__(256) k_cm_round(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, CmDesc d, const u64 *rcp, u64 *part) {
    const u32 c = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const u32 per = 4 + 4 * d.nM, nring = d.L * (per - 1);
    u64 s[3] = {0, 0, 0};
    for (size_t b = (size_t)blockIdx.x * 16 + pl; b < half; b += (size_t)gridDim.x * 16) {
        const u64 e0 = S[2 * b], e1 = S[2 * b + 1], e2 = add_p(e1, sub_p(e1, e0));
        const u64 *t0p = R + ((size_t)nring * ldr + 2 * b) * 16 + c, *t1p = R + ((size_t)(nring + 1) * ldr + 2 * b) * 16 + c;
        const u64 a0 = t0p[0], a1 = t0p[16], b0 = t1p[0], b1 = t1p[16];
        const u64 rz = rcp[d.L * per], rz1 = rcp[d.L * per + 1];
        // rz t0 + rz1 t1 at X = 0, 1, 2
        const u64 z0 = add_p(mont_mul(rz, a0), mont_mul(rz1, b0)), z1 = add_p(mont_mul(rz, a1), mont_mul(rz1, b1)), z2 = add_p(z1, sub_p(z1, z0));
        for (u32 l = 0; l < d.L; l++) {
            const u64 *tp = S + (size_t)(1 + l) * lds + 2 * b;
            const u64 m0 = tp[0], m1 = tp[1], m2 = add_p(m1, sub_p(m1, m0));     // tau (Montgomery) at X = 0, 1, 2
            u64 in0 = 0, in1 = 0, in2 = 0;
            if (c == 0) {                                                        // the constant tau as table 0: rcp[l per] tau
                const u64 r0 = rcp[l * per];
                in0 = mont_mul(r0, from_mont(m0)); in1 = mont_mul(r0, from_mont(m1)); in2 = mont_mul(r0, from_mont(m2));
            }
            for (u32 j = 1; j < per; j++) {
                const u64 *rp = R + ((size_t)(l * (per - 1) + j - 1) * ldr + 2 * b) * 16 + c;
                const u64 v0 = rp[0], v1 = rp[16], v2 = add_p(v1, sub_p(v1, v0)), rj = rcp[l * per + j];
                in0 = add_p(in0, mont_mul(rj, v0)); in1 = add_p(in1, mont_mul(rj, v1)); in2 = add_p(in2, mont_mul(rj, v2));
            }
            s[0] = add_p(s[0], add_p(mont_mul(e0, in0), mont_mul(m0, z0)));
            s[1] = add_p(s[1], add_p(mont_mul(e1, in1), mont_mul(m1, z1)));
            s[2] = add_p(s[2], add_p(mont_mul(e2, in2), mont_mul(m2, z2)));
        }
    }
    __shared__ u64 sm[3][16][16];
    for (int x = 0; x < 3; x++) sm[x][pl][c] = s[x];
    __syncthreads();
    if (threadIdx.x < 48) {
        const u32 x = threadIdx.x >> 4, cc = threadIdx.x & 15;
        u64 t = 0;
        for (int p = 0; p < 16; p++) t = add_p(t, sm[x][p][cc]);
        part[(size_t)blockIdx.x * 48 + threadIdx.x] = t;
    }
}
// The same round with fix_variables of the PREVIOUS round fused in: S / R are the previous tables (2 * half pairs of entries: 4 per new pair), every value is
// fixed with rM (Montgomery) on the way -- f = lo + r (hi - lo) -- and stored to So / Ro (ld_o entries per table) for the next round.  A table entry is used by
// exactly one thread of the round kernel, so the separate k_cm_fix pass (read T, write T/2) and the round's own read (T/2) become one read of T and one write of
// T/2: a fifth of a sumchecker's traffic, and one launch per round instead of three.
__global__ void __launch_bounds__(256) k_cm_round_fused(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, CmDesc d, const u64 *rcp, u64 rM, u64 *So, u64 *Ro,
                                                        size_t ld_o, u64 *part) {
    const u32 c = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const u32 per = 4 + 4 * d.nM, nring = d.L * (per - 1);
    u64 s[3] = {0, 0, 0};
    auto fixw = [&](u64 lo, u64 hi) { return add_p(lo, mont_mul(rM, sub_p(hi, lo))); };
    // ring table `tb` at the new pair b: the two fixed entries (stored), coefficient c
    auto ringpair = [&](u32 tb, size_t b, u64 &v0, u64 &v1) {
        const u64 *rp = R + ((size_t)tb * ldr + 4 * b) * 16 + c;
        v0 = fixw(rp[0], rp[16]); v1 = fixw(rp[32], rp[48]);
        u64 *op = Ro + ((size_t)tb * ld_o + 2 * b) * 16 + c;
        op[0] = v0; op[16] = v1;
    };
    auto scalpair = [&](u32 tb, size_t b, u64 &v0, u64 &v1) {
        const u64 *sp = S + (size_t)tb * lds + 4 * b;
        v0 = fixw(sp[0], sp[1]); v1 = fixw(sp[2], sp[3]);
        if (c == 0) { So[(size_t)tb * ld_o + 2 * b] = v0; So[(size_t)tb * ld_o + 2 * b + 1] = v1; }
    };
    for (size_t b = (size_t)blockIdx.x * 16 + pl; b < half; b += (size_t)gridDim.x * 16) {
        u64 e0, e1, a0, a1, b0, b1;
        scalpair(0, b, e0, e1);
        const u64 e2 = add_p(e1, sub_p(e1, e0));
        ringpair(nring, b, a0, a1); ringpair(nring + 1, b, b0, b1);
        const u64 rz = rcp[d.L * per], rz1 = rcp[d.L * per + 1];
        const u64 z0 = add_p(mont_mul(rz, a0), mont_mul(rz1, b0)), z1 = add_p(mont_mul(rz, a1), mont_mul(rz1, b1)), z2 = add_p(z1, sub_p(z1, z0));
        for (u32 l = 0; l < d.L; l++) {
            u64 m0, m1;
            scalpair(1 + l, b, m0, m1);
            const u64 m2 = add_p(m1, sub_p(m1, m0));
            u64 in0 = 0, in1 = 0, in2 = 0;
            if (c == 0) {
                const u64 r0 = rcp[l * per];
                in0 = mont_mul(r0, from_mont(m0)); in1 = mont_mul(r0, from_mont(m1)); in2 = mont_mul(r0, from_mont(m2));
            }
            for (u32 j = 1; j < per; j++) {
                u64 v0, v1;
                ringpair(l * (per - 1) + j - 1, b, v0, v1);
                const u64 v2 = add_p(v1, sub_p(v1, v0)), rj = rcp[l * per + j];
                in0 = add_p(in0, mont_mul(rj, v0)); in1 = add_p(in1, mont_mul(rj, v1)); in2 = add_p(in2, mont_mul(rj, v2));
            }
            s[0] = add_p(s[0], add_p(mont_mul(e0, in0), mont_mul(m0, z0)));
            s[1] = add_p(s[1], add_p(mont_mul(e1, in1), mont_mul(m1, z1)));
            s[2] = add_p(s[2], add_p(mont_mul(e2, in2), mont_mul(m2, z2)));
        }
    }
    __shared__ u64 sm[3][16][16];
    for (int x = 0; x < 3; x++) sm[x][pl][c] = s[x];
    __syncthreads();
    if (threadIdx.x < 48) {
        const u32 x = threadIdx.x >> 4, cc = threadIdx.x & 15;
        u64 t = 0;
        for (int p = 0; p < 16; p++) t = add_p(t, sm[x][p][cc]);
        part[(size_t)blockIdx.x * 48 + threadIdx.x] = t;
    }
}
void launch_cm_round_fused(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, const CmDesc &d, const u64 *rcp, u64 rM, u64 *So, u64 *Ro, size_t ld_o, u64 *part,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_cm_round_fused, dim3(cm_round_blocks(half)), dim3(256), 0, s, S, lds, R, ldr, half, d, rcp, rM, So, Ro, ld_o, part);
}
// (the host adds the block partials -- 48 / 64 words each, read from mapped memory -- before it can run the transcript: one block per CU at most)
// workgroups of a sumcheck round (16 pairs per workgroup and pass; up to 2048: one per CU leaves a memory-bound round at 1/3 of the HBM rate).  Above 256 the
// driver adds the block partials on the device (launch_reduce) instead of on the host.  LFPLUS_ROUND_BLOCKS moves the cap
u32 cm_round_blocks(size_t half) {
    constexpr size_t cap = 2048;
    size_t b = cdiv(half, 16);
    return (u32)(b < 1 ? 1 : (b > cap ? cap : b));
}
void launch_cm_round(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, const CmDesc &d, const u64 *rcp, u64 *part, hipStream_t s) {
    hipLaunchKernelGGL(k_cm_round, dim3(cm_round_blocks(half)), dim3(256), 0, s, S, lds, R, ldr, half, d, rcp, part);
}
// fix_variables of tables of `w` words per entry: out[t][b][x] = in[t][2b][x] + r (in[t][2b+1][x] - in[t][2b][x]); r in Montgomery form (the words keep
// whatever form they have)
__global__ void __launch_bounds__(256) k_cm_fix(const u64 *in, size_t ld_in, u64 *out, size_t ld_out, u32 w, size_t half, u64 rM) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= half * w) return;
    const size_t b = i / w, x = i % w;
    const u64 *p = in + ((size_t)blockIdx.y * ld_in + 2 * b) * w + x;
    const u64 lo = p[0], hi = p[w];
    out[((size_t)blockIdx.y * ld_out + b) * w + x] = add_p(lo, mont_mul(rM, sub_p(hi, lo)));
}
void launch_cm_fix(const u64 *in, size_t ld_in, u64 *out, size_t ld_out, u32 w, u32 ntab, size_t half, u64 rM, hipStream_t s) {
    hipLaunchKernelGGL(k_cm_fix, dim3((unsigned)cdiv(half * w, 256), ntab), dim3(256), 0, s, in, ld_in, out, ld_out, w, half, rM);
}
// ---- the sumcheckers through BATCHED tables.  The combination function (cm.rs:287-311) is linear in the instance tables:
//     eq(b) sum_l sum_j rc^(l per + j) T_lj(b)  +  sum_l tau_l(b) (rc^(L per) t0(b) + rc^(L per + 1) t1(b))   =   eq(b) U(b) + V(b) Z(b)
// with U = sum_lj rc^(l per + j) T_lj (ring), V = sum_l tau_l (scalar), Z = rc^(L per) t0 + rc^(L per + 1) t1 (ring), and fix_variables is linear as well: the round
// messages of the sumcheck over (eq, V | U, Z) ARE the messages of the sumcheck over the 1 + L scalar and L (per - 1) + 2 ring tables, word for word (exact field
// arithmetic), at two ring tables per round instead of 47 (L = 3, three matrices).  The evaluations of the instance tables at the final point, which the
// reference reads off its fully fixed tables (cm.rs:313-331), are eq(ro, .)-weighted sums over the ORIGINAL tables: one more pass over them (k_cm_evals).
// Per sumchecker at 2^20 rows: one read of the tables to combine + one to evaluate (2 x 6.3 GB) instead of ~3 x 6.3 GB spread over 20 rounds of 47-table kernels.
// thread = (row, coefficient).  S2 = eq | V (ld2 entries each), R2 = U | Z
__global__ void __launch_bounds__(256) k_cm_combine(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t n, CmDesc d, const u64 *rcp, u64 *S2, u64 *R2, size_t ld2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 16) return;
    const size_t row = i >> 4;
    const u32 c = (u32)(i & 15);
    const u32 per = 4 + 4 * d.nM, nring = d.L * (per - 1);
    u64 u = 0, v = 0;
    Acc160 ua;                // the L (per - 1) products rc^i T_i(row) as one lazy sum
    acc160_zero(ua);
    for (u32 l = 0; l < d.L; l++) {
        const u64 m = S[(size_t)(1 + l) * lds + row];
        v = add_p(v, m);
        if (c == 0) u = add_p(u, mont_mul(rcp[l * per], from_mont(m)));          // the constant tau_l as table 0 of the instance
        const u64 *rp = R + ((size_t)(l * (per - 1)) * ldr + row) * 16 + c;
#pragma unroll 5
        for (u32 j = 1; j < per; j++) acc160_mad(ua, rcp[l * per + j], rp[(size_t)(j - 1) * ldr * 16]);
    }
    const u64 t0 = R[((size_t)nring * ldr + row) * 16 + c], t1 = R[((size_t)(nring + 1) * ldr + row) * 16 + c];
    R2[row * 16 + c] = add_p(u, acc160_red(ua));
    R2[(ld2 + row) * 16 + c] = add_p(mont_mul(rcp[d.L * per], t0), mont_mul(rcp[d.L * per + 1], t1));
    if (c == 0) { S2[row] = S[row]; S2[ld2 + row] = v; }
}
void launch_cm_combine(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t n, const CmDesc &d, const u64 *rcp, u64 *S2, u64 *R2, size_t ld2, hipStream_t s) {
    hipLaunchKernelGGL(k_cm_combine, dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, s, S, lds, R, ldr, n, d, rcp, S2, R2, ld2);
}
// One round over the batched tables: sum over the pairs of eq U + V Z at X = 0, 1, 2.  FUSED: S / R are the previous round's tables (4 entries per new pair), fixed
// with rM on the way and stored to So / Ro (as k_cm_round_fused).  thread = (pair, coefficient); part[block][3][16] canonical
template <bool FUSED>
__global__ void __launch_bounds__(256) k_cm2_round(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, u64 rM, u64 *So, u64 *Ro, size_t ld_o, u64 *part) {
    const u32 c = threadIdx.x & 15, pl = threadIdx.x >> 4;
    u64 s[3] = {0, 0, 0};
    auto fixw = [&](u64 lo, u64 hi) { return add_p(lo, mont_mul(rM, sub_p(hi, lo))); };
    for (size_t b = (size_t)blockIdx.x * 16 + pl; b < half; b += (size_t)gridDim.x * 16) {
        u64 sv[2][2], rv[2][2];
#pragma unroll
        for (u32 tb = 0; tb < 2; tb++) {
            if (FUSED) {
                const u64 *sp = S + (size_t)tb * lds + 4 * b, *rp = R + ((size_t)tb * ldr + 4 * b) * 16 + c;
                sv[tb][0] = fixw(sp[0], sp[1]); sv[tb][1] = fixw(sp[2], sp[3]);
                rv[tb][0] = fixw(rp[0], rp[16]); rv[tb][1] = fixw(rp[32], rp[48]);
                if (c == 0) { So[(size_t)tb * ld_o + 2 * b] = sv[tb][0]; So[(size_t)tb * ld_o + 2 * b + 1] = sv[tb][1]; }
                u64 *op = Ro + ((size_t)tb * ld_o + 2 * b) * 16 + c;
                op[0] = rv[tb][0]; op[16] = rv[tb][1];
            } else {
                const u64 *sp = S + (size_t)tb * lds + 2 * b, *rp = R + ((size_t)tb * ldr + 2 * b) * 16 + c;
                sv[tb][0] = sp[0]; sv[tb][1] = sp[1];
                rv[tb][0] = rp[0]; rv[tb][1] = rp[16];
            }
        }
        const u64 e2 = add_p(sv[0][1], sub_p(sv[0][1], sv[0][0])), v2 = add_p(sv[1][1], sub_p(sv[1][1], sv[1][0]));
        const u64 u2 = add_p(rv[0][1], sub_p(rv[0][1], rv[0][0])), z2 = add_p(rv[1][1], sub_p(rv[1][1], rv[1][0]));
        s[0] = add_p(s[0], add_p(mont_mul(sv[0][0], rv[0][0]), mont_mul(sv[1][0], rv[1][0])));
        s[1] = add_p(s[1], add_p(mont_mul(sv[0][1], rv[0][1]), mont_mul(sv[1][1], rv[1][1])));
        s[2] = add_p(s[2], add_p(mont_mul(e2, u2), mont_mul(v2, z2)));
    }
    __shared__ u64 sm[3][16][16];
    for (int x = 0; x < 3; x++) sm[x][pl][c] = s[x];
    __syncthreads();
    if (threadIdx.x < 48) {
        const u32 x = threadIdx.x >> 4, cc = threadIdx.x & 15;
        u64 t = 0;
        for (int p = 0; p < 16; p++) t = add_p(t, sm[x][p][cc]);
        part[(size_t)blockIdx.x * 48 + threadIdx.x] = t;
    }
}
void launch_cm2_round(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, u64 *part, hipStream_t s) {
    hipLaunchKernelGGL((k_cm2_round<false>), dim3(cm_round_blocks(half)), dim3(256), 0, s, S, lds, R, ldr, half, (u64)0, (u64 *)nullptr, (u64 *)nullptr, (size_t)0, part);
}
void launch_cm2_round_fused(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t half, u64 rM, u64 *So, u64 *Ro, size_t ld_o, u64 *part, hipStream_t s) {
    hipLaunchKernelGGL((k_cm2_round<true>), dim3(cm_round_blocks(half)), dim3(256), 0, s, S, lds, R, ldr, half, rM, So, Ro, ld_o, part);
}
// part[chunk][table][16] = sum over the chunk's rows of eq[row] * T_table[row] (eq Montgomery, tables canonical): the evaluations of `ntab` ring tables at the
// point eq was built from.  grid (chunks, tables); thread = (row lane, coefficient): a wave reads 512 contiguous bytes of the table per step
__global__ void __launch_bounds__(256) k_cm_evals(const u64 *R, size_t ldr, size_t n, const u64 *eq, u32 ntab, u64 *part) {
    const u32 c = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const u64 *T = R + (size_t)blockIdx.y * ldr * 16;
    Acc160 acc;
    acc160_zero(acc);
#pragma unroll 4
    for (size_t row = (size_t)blockIdx.x * 16 + rl; row < n; row += (size_t)gridDim.x * 16) acc160_mad(acc, eq[row], T[row * 16 + c]);
    __shared__ u64 sm[16][16];
    sm[rl][c] = acc160_red(acc);
    __syncthreads();
    if (threadIdx.x < 16) {
        u64 t = 0;
        for (int p = 0; p < 16; p++) t = add_p(t, sm[p][threadIdx.x]);
        part[((size_t)blockIdx.x * ntab + blockIdx.y) * 16 + threadIdx.x] = t;
    }
}
u32 cm_eval_chunks(size_t n) {
    const size_t b = cdiv(n, 16 * 64);      // 64 rows per thread and chunk at least
    return (u32)(b < 1 ? 1 : (b > 64 ? 64 : b));
}
// ---- compact instance tables (round 5) ----------------------------------------------------------------------------------------------------
// Per instance, table 0 (m_tau) is a column of unit monomials and tables 3 + 4q (M_q tau, M_q with constant coefficients) are columns of scalars: 128-byte ring
// elements whose content is one exponent byte / one word.  The two streaming passes of a sumchecker read them in that form (12 of 47 tables at L = 3, nM = 3: a
// quarter of the bytes); the arithmetic is the dense kernels' -- the same lazy products, against 0 / 1 / the scalar -- so every word of the proof is unchanged.
__global__ void __launch_bounds__(256) k_cm_combine_c(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t n, CmDesc d, const u64 *rcp, CmCompact cc, u64 *S2, u64 *R2,
                                                      size_t ld2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 16) return;
    const size_t row = i >> 4;
    const u32 c = (u32)(i & 15);
    const u32 per = 4 + 4 * d.nM, nring = d.L * (per - 1);
    u64 u = 0, v = 0;
    Acc160 ua;
    acc160_zero(ua);
    for (u32 l = 0; l < d.L; l++) {
        const u64 m = S[(size_t)(1 + l) * lds + row];
        v = add_p(v, m);
        if (c == 0) u = add_p(u, mont_mul(rcp[l * per], from_mont(m)));
        const u64 *rp = R + ((size_t)(l * (per - 1)) * ldr + row) * 16 + c;
        acc160_mad(ua, rcp[l * per + 1], (u64)((u32)exp_of(cc.mtau[l][row]) == c));                       // m_tau: X^e
        acc160_mad(ua, rcp[l * per + 2], rp[(size_t)1 * ldr * 16]);                                      // f
        acc160_mad(ua, rcp[l * per + 3], rp[(size_t)2 * ldr * 16]);                                      // h
        for (u32 q = 0; q < d.nM; q++) {
            const u32 j = 4 + 4 * q;
            acc160_mad(ua, rcp[l * per + j], c == 0 ? cc.mts[((size_t)l * d.nM + q) * cc.ldm + row] : 0);   // M_q tau: a scalar
#pragma unroll
            for (u32 x = 1; x < 4; x++) acc160_mad(ua, rcp[l * per + j + x], rp[(size_t)(j + x - 1) * ldr * 16]);
        }
    }
    const u64 t0 = R[((size_t)nring * ldr + row) * 16 + c], t1 = R[((size_t)(nring + 1) * ldr + row) * 16 + c];
    R2[row * 16 + c] = add_p(u, acc160_red(ua));
    R2[(ld2 + row) * 16 + c] = add_p(mont_mul(rcp[d.L * per], t0), mont_mul(rcp[d.L * per + 1], t1));
    if (c == 0) { S2[row] = S[row]; S2[ld2 + row] = v; }
}
void launch_cm_combine_c(const u64 *S, size_t lds, const u64 *R, size_t ldr, size_t n, const CmDesc &d, const u64 *rcp, const CmCompact &cc, u64 *S2, u64 *R2, size_t ld2,
                         hipStream_t s) {
    hipLaunchKernelGGL(k_cm_combine_c, dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, s, S, lds, R, ldr, n, d, rcp, cc, S2, R2, ld2);
}
// evaluations of the DENSE tables named in `list` only (part / out keep the layout of launch_cm_evals: slot = table index; the other slots are the caller's)
__global__ void __launch_bounds__(256) k_cm_evals_list(const u64 *R, size_t ldr, size_t n, const u64 *eq, u32 ntab, CmTabList list, u64 *part) {
    const u32 c = threadIdx.x & 15, rl = threadIdx.x >> 4, tab = list.idx[blockIdx.y];
    const u64 *T = R + (size_t)tab * ldr * 16;
    Acc160 acc;
    acc160_zero(acc);
#pragma unroll 4
    for (size_t row = (size_t)blockIdx.x * 16 + rl; row < n; row += (size_t)gridDim.x * 16) acc160_mad(acc, eq[row], T[row * 16 + c]);
    __shared__ u64 sm[16][16];
    sm[rl][c] = acc160_red(acc);
    __syncthreads();
    if (threadIdx.x < 16) {
        u64 t = 0;
        for (int p = 0; p < 16; p++) t = add_p(t, sm[p][threadIdx.x]);
        part[((size_t)blockIdx.x * ntab + tab) * 16 + threadIdx.x] = t;
    }
}
// the scalar tables: part[blk][y] = sum over the block's rows of eq[row] mts[y][row]; finish: out[tab(y)][0] = the sum, coefficients 1..15 = 0
__global__ void __launch_bounds__(256) k_cm_evals_scalar(const u64 *mts, size_t ldm, size_t n, const u64 *eq, u32 nt, u64 *part) {
    u64 s[4] = {0, 0, 0, 0};
    const u64 *y = mts + (size_t)blockIdx.y * ldm;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s[0] = add_p(s[0], mont_mul(eq[i], y[i]));
    block_sum4(s, part + ((size_t)blockIdx.x * nt + blockIdx.y) * 4);
}
__global__ void __launch_bounds__(256) k_cm_evals_scalar_fin(const u64 *part, u32 chunks, u32 nt, CmTabList tabs, u64 *out) {
    const u32 y = blockIdx.x, c = threadIdx.x & 15;
    if (threadIdx.x >= 16) return;
    u64 s = 0;
    if (c == 0)
        for (u32 ch = 0; ch < chunks; ch++) s = add_p(s, part[((size_t)ch * nt + y) * 4]);
    out[(size_t)tabs.idx[y] * 16 + c] = s;
}
void launch_cm_evals_c(const u64 *R, size_t ldr, size_t n, const u64 *eq, u32 ntab, const CmTabList &dense, u32 ndense, const CmCompact &cc, u32 L, u32 nM, u32 per,
                       u64 *part, u64 *out, hipStream_t s) {
    const u32 ch = cm_eval_chunks(n);
    hipLaunchKernelGGL(k_cm_evals_list, dim3(ch, ndense), dim3(256), 0, s, R, ldr, n, eq, ntab, dense, part);
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)cdiv((size_t)ntab * 16, 32)), dim3(1024), 0, s, part, ch, (size_t)ntab * 16, ntab * 16, 0, out);   // (the compact slots: overwritten below)
    for (u32 l = 0; l < L; l++) launch_wmono(cc.mtau[l], n, 1, eq, 1, part, out + (size_t)l * (per - 1) * 16, s);          // sum_row eq[row] X^e(row): the exponent histogram
    if (nM) {
        CmTabList tabs;
        const u32 nt = L * nM, chs = eval_chunks(n);
        for (u32 l = 0; l < L; l++) for (u32 q = 0; q < nM; q++) tabs.idx[l * nM + q] = (uint16_t)(l * (per - 1) + 3 + 4 * q);
        hipLaunchKernelGGL(k_cm_evals_scalar, dim3(chs, nt), dim3(256), 0, s, cc.mts, cc.ldm, n, eq, nt, part);
        hipLaunchKernelGGL(k_cm_evals_scalar_fin, dim3(nt), dim3(256), 0, s, part, chs, nt, tabs, out);
    }
}
void launch_cm_evals(const u64 *R, size_t ldr, size_t n, const u64 *eq, u32 ntab, u64 *part, u64 *out, hipStream_t s) {
    const u32 ch = cm_eval_chunks(n);
    hipLaunchKernelGGL(k_cm_evals, dim3(ch, ntab), dim3(256), 0, s, R, ldr, n, eq, ntab, part);
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)cdiv((size_t)ntab * 16, 32)), dim3(1024), 0, s, part, ch, (size_t)ntab * 16, ntab * 16, 0, out);
}
}  // namespace lfp

// ---- ComR1CS::linearize (r1cs.rs:76-139) ------------------------------------------------------------------------------------------------------
namespace lfp {
// One round of the degree-3 sumcheck of eq (ga gb - gc) (comb_fn r1cs.rs:95; ring products).  E: eq(r, .) Montgomery scalars; G: ga | gb | gc, canonical
// ring tables [table][ld][16].  thread = (pair, coefficient); part[block][4][16] canonical.
// The pair's four evaluations.  ga(X) gb(X) is quadratic in X: the three negacyclic products P(0) = a0 b0, P(1) = a1 b1, P(inf) = (a1 - a0)(b1 - b0) give
// P(2) = 2 P(1) - P(0) + 2 P(inf) and P(3) = 3 P(1) - 2 P(0) + 6 P(inf) (exact mod p: the words of the message are those of four products); the 16 terms of a
// product coefficient are one lazy signed sum (acc160_mad_signed) with a single reduction.  First version: four products of 16 mont_mul + add_p / sub_p each --
// 0.77 ms for the first round at 2^20 rows against 0.06 ms of table traffic.
__device__ __forceinline__ u64 r1cs_negacyclic(const u64 (*la)[16], const u64 (*lb)[16], u32 pl, u32 c) {
    Acc160 n;
    acc160_bias16(n);
#pragma unroll
    for (u32 j = 0; j < 16; j++) acc160_mad_signed(n, la[pl][j], lb[pl][(c - j) & 15], j > c);
    return acc160_red(n);
}
__device__ __forceinline__ void r1cs_pair_eval(u64 (*la)[16][16], u64 (*lb)[16][16], u32 pl, u32 c, u64 e0, u64 de, u64 a0, u64 a1, u64 b0, u64 b1, u64 c0, u64 dc, u64 s[4]) {
    la[0][pl][c] = to_mont(a0); la[1][pl][c] = to_mont(a1); la[2][pl][c] = to_mont(sub_p(a1, a0));
    lb[0][pl][c] = b0; lb[1][pl][c] = b1; lb[2][pl][c] = sub_p(b1, b0);
    __syncthreads();
    const u64 p0 = r1cs_negacyclic(la[0], lb[0], pl, c), p1 = r1cs_negacyclic(la[1], lb[1], pl, c), pi = r1cs_negacyclic(la[2], lb[2], pl, c);
    __syncthreads();
    const u64 d = add_p(p1, pi), t2 = sub_p(add_p(d, d), p0), pi2 = add_p(pi, pi), t3 = add_p(add_p(t2, sub_p(p1, p0)), add_p(pi2, pi2));
    const u64 e1 = add_p(e0, de), e2 = add_p(e1, de), e3 = add_p(e2, de), c1 = add_p(c0, dc), c2 = add_p(c1, dc), c3 = add_p(c2, dc);
    s[0] = add_p(s[0], mont_mul(e0, sub_p(p0, c0)));
    s[1] = add_p(s[1], mont_mul(e1, sub_p(p1, c1)));
    s[2] = add_p(s[2], mont_mul(e2, sub_p(t2, c2)));
    s[3] = add_p(s[3], mont_mul(e3, sub_p(t3, c3)));
}
__device__ __forceinline__ void r1cs_block_store(u64 s[4], u32 pl, u32 c, u64 *part) {
    __shared__ u64 sm[4][16][16];
    for (int x = 0; x < 4; x++) sm[x][pl][c] = s[x];
    __syncthreads();
    if (threadIdx.x < 64) {
        const u32 x = threadIdx.x >> 4, cc = threadIdx.x & 15;
        u64 t = 0;
        for (int p = 0; p < 16; p++) t = add_p(t, sm[x][p][cc]);
        part[(size_t)blockIdx.x * 64 + threadIdx.x] = t;
    }
}
__global__ void __launch_bounds__(256) k_r1cs_round(const u64 *E, const u64 *G, size_t ld, size_t half, u64 *part) {
    const u32 c = threadIdx.x & 15, pl = threadIdx.x >> 4;
    __shared__ u64 la[3][16][16], lb[3][16][16];
    u64 s[4] = {0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * 16; base < half; base += (size_t)gridDim.x * 16) {
        const size_t b = base + pl;
        u64 e0 = 0, de = 0, a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0, dc = 0;
        if (b < half) {
            const u64 *ga = G + (2 * b) * 16 + c, *gb = ga + ld * 16, *gc = gb + ld * 16;
            e0 = E[2 * b]; de = sub_p(E[2 * b + 1], e0);
            a0 = ga[0]; a1 = ga[16];
            b0 = gb[0]; b1 = gb[16];
            c0 = gc[0]; dc = sub_p(gc[16], c0);
        }
        r1cs_pair_eval(la, lb, pl, c, e0, de, a0, a1, b0, b1, c0, dc, s);
    }
    r1cs_block_store(s, pl, c, part);
}
// the same round with fix_variables of the previous round fused in (as k_cm_round_fused): E / G are the previous tables (4 entries per new pair), fixed with rM on
// the way and stored to Eo / Go (ld_o entries per table)
__global__ void __launch_bounds__(256) k_r1cs_round_fused(const u64 *E, const u64 *G, size_t ld, size_t half, u64 rM, u64 *Eo, u64 *Go, size_t ld_o, u64 *part) {
    const u32 c = threadIdx.x & 15, pl = threadIdx.x >> 4;
    __shared__ u64 la[3][16][16], lb[3][16][16];
    u64 s[4] = {0, 0, 0, 0};
    auto fixw = [&](u64 lo, u64 hi) { return add_p(lo, mont_mul(rM, sub_p(hi, lo))); };
    for (size_t base = (size_t)blockIdx.x * 16; base < half; base += (size_t)gridDim.x * 16) {
        const size_t b = base + pl;
        u64 e0 = 0, de = 0, c0 = 0, dc = 0;
        u64 v0[3] = {0, 0, 0}, v1[3] = {0, 0, 0};
        if (b < half) {
            e0 = fixw(E[4 * b], E[4 * b + 1]);
            const u64 e1 = fixw(E[4 * b + 2], E[4 * b + 3]);
            if (c == 0) { Eo[2 * b] = e0; Eo[2 * b + 1] = e1; }
            de = sub_p(e1, e0);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const u64 *gp = G + ((size_t)q * ld + 4 * b) * 16 + c;
                v0[q] = fixw(gp[0], gp[16]); v1[q] = fixw(gp[32], gp[48]);
                u64 *op = Go + ((size_t)q * ld_o + 2 * b) * 16 + c;
                op[0] = v0[q]; op[16] = v1[q];
            }
            c0 = v0[2]; dc = sub_p(v1[2], c0);
        }
        r1cs_pair_eval(la, lb, pl, c, e0, de, v0[0], v1[0], v0[1], v1[1], c0, dc, s);
    }
    r1cs_block_store(s, pl, c, part);
}
void launch_r1cs_round_fused(const u64 *E, const u64 *G, size_t ld, size_t half, u64 rM, u64 *Eo, u64 *Go, size_t ld_o, u64 *part, hipStream_t s) {
    hipLaunchKernelGGL(k_r1cs_round_fused, dim3(cm_round_blocks(half)), dim3(256), 0, s, E, G, ld, half, rM, Eo, Go, ld_o, part);
}
void launch_r1cs_round(const u64 *E, const u64 *G, size_t ld, size_t half, u64 *part, hipStream_t s) {
    hipLaunchKernelGGL(k_r1cs_round, dim3(cm_round_blocks(half)), dim3(256), 0, s, E, G, ld, half, part);
}
// acc[i] = acc[i] + x[i] mod p
__global__ void __launch_bounds__(256) k_vec_add(u64 *acc, const u64 *x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] = add_p(acc[i], x[i]);
}
void launch_vec_add(u64 *acc, const u64 *x, size_t n, hipStream_t s) { hipLaunchKernelGGL(k_vec_add, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, acc, x, n); }
// *flag |= bit if any of the n words is not a canonical residue (an uploaded vector is checked where it lands: a host scan of a 2^20-row witness costs more than its upload)
__global__ void __launch_bounds__(256) k_check_canonical(const u64 *x, size_t n, u32 *flag, u32 bit) {
    bool bad = false;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (size_t)gridDim.x * 512) {
        if (i + 1 < n) { const ulonglong2 v = *(const ulonglong2 *)(x + i); bad |= v.x >= P || v.y >= P; }
        else bad |= x[i] >= P;
    }
    if (bad) atomicOr(flag, bit);
}
void launch_check_canonical(const u64 *x, size_t n, u32 *flag, u32 bit, hipStream_t s) {
    size_t nb = cdiv(n, 512 * 8);
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_check_canonical, dim3((unsigned)nb), dim3(256), 0, s, x, n, flag, bit);
}
}  // namespace lfp

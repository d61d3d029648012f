// lf_sv_rounds.h -- the first rounds of the folding sumcheck as exact int8 GEMMs on the matrix cores (lf_sv_rounds.hip).
//
// Before round i of the folding sumcheck (nifs/folding/utils.rs:273-325, b = 2) an f-hat entry is  sum_b W_b y_b  with V = 2^(i-1) weights
// W_b = eq((r_1..r_{i-1}), b) and ternary digits y_b; a pair of entries therefore depends on 2V "signed bits" y_x = s_x b_x (x < V: the even
// entry, x >= V: the odd one) of 2V consecutive witness positions, and its cubic
//     (h^3 - h)(X),   h(X) = sum_x w_x(X) y_x,   w_x = W_x (1 - X)  (x < V),   W_{x-V} X  (x >= V)
// is a sum over "pairs" pi = (sigma, beta): a product of an odd number of signs times an AND of bits (y^2 = b, y^3 = y), with a polynomial
// coefficient C_pi(X) that depends on the challenges only.  The round message needs
//     sum_p eqB(p)(X) sum_tables mu_T sum_pi C_pi(X) sigma_pi(T,p) beta_pi(T,p)
// and the inner sums over p,  M_pi[T] = sum_p eqB(2p | 2p+1) * (+-1 | 0),  are an int8 GEMM: rows = the 16 digit planes of a
// (side, coefficient) group, inner dimension = pairs p, columns = the 24 + 24 balanced base-256 digits of eqB(2p), eqB(2p+1).
// No modular multiplication touches the 2^20-entry tables any more; the field work is 2 M_pi per (table, pair).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
namespace lf {
struct SvPair {
    unsigned char s, b;   // subset of the 2V signed bits whose signs are multiplied / whose magnitude bits are ANDed
};
constexpr int sv_num_pairs(int V) { return 2 * V + 2 * V * (2 * V - 1) + 2 * V * (2 * V - 1) * (2 * V - 2) / 6; }
// canonical order: for every z: ({z},{z}), then ({z},{q,z}) for q != z (all pairs with one sign first, grouped by that sign), then the triples
constexpr SvPair sv_pair(int V, int idx) {
    const int NX = 2 * V;
    int i = 0;
    for (int z = 0; z < NX; z++) {
        if (i == idx) return SvPair{(unsigned char)(1u << z), (unsigned char)(1u << z)};
        i++;
        for (int q = 0; q < NX; q++) {
            if (q == z) continue;
            if (i == idx) return SvPair{(unsigned char)(1u << z), (unsigned char)((1u << z) | (1u << q))};
            i++;
        }
    }
    for (int x = 0; x < NX; x++)
        for (int y = x + 1; y < NX; y++)
            for (int z = y + 1; z < NX; z++) {
                if (i == idx) {
                    const unsigned char m = (unsigned char)((1u << x) | (1u << y) | (1u << z));
                    return SvPair{m, m};
                }
                i++;
            }
    return SvPair{0, 0};
}
// digit monomials per wave and launch: bounded by the accumulator registers (4 per monomial and column tile); with two column tiles (split form) twice as many fit
constexpr int sv_pairs_per_wave(int V, int NT = 3) { return V == 1 ? 4 : (V == 2 ? (NT == 2 ? 20 : 10) : (NT == 2 ? 40 : 20)); }

bool sv_shape_ok(int V, size_t npairs, uint32_t K);
size_t sv_eb_bytes(size_t npairs);                    // packed eqB pair bytes [48][padded pairs]
size_t sv_part_words(int V, size_t npairs, uint32_t K);   // int32 words of the per-chunk partial tiles
size_t sv_tot_words(int V, uint32_t K);               // int32 words of their sums
size_t sv_tp_words(uint32_t K);                       // u64 words of the per-table polynomials
// bit-plane form of a witness ([24][ldp] int32 planes, n positions, |v| < 2^K): rows of magnitude bits + one row of signs per coefficient
size_t sv_bits_words(size_t n, uint32_t K, uint32_t rd = 24);   // rd: coefficients per ring element (24 / 72)
void launch_sv_bits(const int32_t *planes, size_t ldp, size_t n, uint32_t K, uint32_t *bits, hipStream_t s, uint32_t rd = 24);
// The GEMM stage alone, for the BabyBear backend (bb_sv_rounds.hip): rd coefficient groups per side, three column tiles of digit bytes EB[48][ldeb] in slot
// order, all sv_num_pairs(V) digit monomials; the summed int32 tiles are left in tot (layout: lf_sv_rounds.hip).  0, or -1 (shape).
size_t sv_tot_words_rd(uint32_t rd, int V, uint32_t K);
size_t sv_part_words_rd(uint32_t rd, int V, uint32_t K);
size_t sv_ldeb_pub(size_t npairs);
__host__ __device__ constexpr int sv_slot_pair_pub(int V, int j, int q) { return (j / (4 / V)) * (16 / V) + (j % (4 / V)) + (4 / V) * q; }
int launch_sv_gemm_tiles(uint32_t rd, int V, const uint32_t *bitsL, const uint32_t *bitsR, size_t nplanes, const unsigned char *EB, size_t ldeb, size_t npairs, uint32_t K,
                         int32_t *part, int32_t *tot, hipStream_t s);
// norm part of round log2(V)+1:  out[X*24 + 3*slot + q] = gpart[...] + sum_tables mu_T sum_p eqB(p)(X) (h_T^3 - h_T)(p)(X),  X = 0..4.
// bitsL / bitsR: launch_sv_bits forms of the two witnesses (nplanes positions); eqB: [3][ldeq]; the launch covers the pairs [pair0, pair0 + npairs)
// (a rank's slice of a sharded step: the partial messages of the ranks add up; gpart = the same slice's G part);
// coef: [sv_num_pairs(V)][4][3] words C_pi (device); mu_pow: [2K*3].  Returns 0, or -1 if the shape is not handled.
struct DevCrt;
struct Fq3Const;
int launch_sv_round(const DevCrt &t, int V, const uint32_t *bitsL, const uint32_t *bitsR, size_t nplanes, const uint64_t *eqB, size_t ldeq, size_t pair0, size_t npairs, uint32_t K,
                    const Fq3Const *mu_pow, const uint64_t *coef, unsigned char *EB, int32_t *part, int32_t *tot, uint64_t *tp, const uint64_t *gpart, uint64_t *out,
                    hipStream_t s, hipEvent_t gpart_ready = nullptr,
                    // split form (optional): eqB(2p + h) = w01[h] * E[p] with ONE value E[p] per pair (E: [3][ldE], indexed by the global pair) -- the GEMM then runs
                    // against the 24 digit columns of E (two column tiles instead of three) and the finish multiplies by w01[0], w01[1] (host pointers)
                    const uint64_t *E = nullptr, size_t ldE = 0, const Fq3Const *w01 = nullptr);
// v_s[k][c][q] = sum_i eq[q][i] * digit_k(planes[c][i]) of ONE witness from its bit-plane form (the two single pairs of the round-1 GEMM);
// out[(k*24 + c)*3 + q] canonical.  Scratch: EB sv_eb_bytes(n / 2), part sv_vs_part_words(n, K), tot sv_vs_tot_words(K).  0, or -1 (shape).
size_t sv_vs_part_words(size_t n, uint32_t K);
size_t sv_vs_tot_words(uint32_t K);
int launch_sv_vs(const uint32_t *bits, size_t n, const uint64_t *eq, size_t ldeq, uint32_t K, unsigned char *EB, int32_t *part, int32_t *tot, uint64_t *out, hipStream_t s);
// the same in two steps (the pass over the witness starts before the point is complete): see lf_sv_rounds.hip
size_t sv_vs_max_blocks(uint32_t K);
size_t sv_vs_blocks_part_words(uint32_t nblocks, uint32_t K);
int launch_sv_vs_blocks(const uint32_t *bits, size_t n, const uint64_t *eq_lo, size_t ldeq, uint32_t J, uint32_t K, unsigned char *EB, int32_t *part, hipStream_t s);
void launch_sv_vs_combine(const DevCrt &t, const int32_t *part, uint32_t nblocks, const uint64_t *wts, uint32_t K, uint64_t *out, hipStream_t s);
}  // namespace lf

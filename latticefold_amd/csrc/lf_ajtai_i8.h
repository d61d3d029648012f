// lf_ajtai_i8.h -- digit-plane Ajtai commitments on the int8 matrix cores (lf_ajtai_i8.hip), shared by the two ring backends.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
namespace lf {
struct AjtaiI8Ring {
    uint32_t RD, NL;      // ring degree (24 / 72), bytes per canonical coefficient (8 / 4)
    uint64_t p_small;     // modulus when it is below 2^32, 0 = Goldilocks
    int soa_out;          // coefficient-form results as SoA [RD][elements] (1) or AoS [elements][RD] (0); canonical u64 either way
};
inline AjtaiI8Ring ajtai_i8_goldilocks() { return AjtaiI8Ring{24, 8, 0, 1}; }
inline AjtaiI8Ring ajtai_i8_babybear() { return AjtaiI8Ring{72, 4, 2013265921ull, 0}; }
// A repacked once per matrix: row i of a row chunk (canonical coefficients, element (c, j) at coef[c*cs + j*js]) -> bytes in MFMA operand
// order, MT = ajtai_i8_row_tiles(rows of the chunk)
// the same from the NTT form of a Goldilocks row in one pass (dense inverse map + packing)
void launch_ajtai_icrt_pack_i8(const uint64_t *icrt_mat, const uint64_t *ntt, size_t n, uint32_t i, uint32_t MT, unsigned char *Ab, hipStream_t s);
void launch_ajtai_pack_i8(const uint64_t *coef, size_t cs, size_t js, size_t n, uint32_t i, uint32_t MT, uint32_t RD, uint32_t NL, unsigned char *Ab, hipStream_t s);
uint32_t ajtai_i8_row_tiles(const AjtaiI8Ring &R, uint32_t kappa);
uint32_t ajtai_i8_col_tiles(const AjtaiI8Ring &R, uint32_t NP);
uint32_t ajtai_i8_max_rows(const AjtaiI8Ring &R);
uint32_t ajtai_i8_max_planes(const AjtaiI8Ring &R);
uint32_t ajtai_i8_max_planes_mt(const AjtaiI8Ring &R, uint32_t MT);   // ... for this row-tile count (the specialised kernels take two plane groups of 8 per launch)
size_t ajtai_i8_slack_bytes();   // readable bytes required behind the packed matrix (the tile copy of the kernel is unconditional)
size_t ajtai_i8_part_words(uint32_t nwg, uint32_t MT, uint32_t NT);
size_t ajtai_i8_sum_words(const AjtaiI8Ring &R, uint32_t MT, uint32_t NT, uint32_t NP);
// commitments of the digit planes k0 .. k0+NP-1 of `planes` ([RD][ld] int32) under rows [row0, row0+kappa) of A (one packed row chunk with MT
// row tiles): coefficient-form results into coef_out (element plane*kappa_total + row).  Returns the grid size or -1.
int launch_ajtai_i8(const AjtaiI8Ring &R, const unsigned char *Ab, uint32_t MT, const int32_t *planes, size_t ld, size_t n, uint32_t kappa, uint32_t row0,
                    uint32_t kappa_total, uint32_t k0, uint32_t NP, uint32_t nwg, int32_t *part, int32_t *dsum, long long *sum, uint64_t *coef_out, hipStream_t s,
                    const uint32_t *bits = nullptr, size_t bits_nw = 0, uint32_t bits_rows = 0);
// bits (optional): the bit-plane form of `planes` (lf_sv_rounds.h launch_sv_bits over the same columns: [RD][bits_rows][bits_nw] words); the
// 24-ring / 13-row-tile kernel then cuts its digits from two words per (plane, coefficient) and tile instead of eight int32 values
// ---- general commitments on the same byte planes of A (lf_ajtai_i8g.hip): AjtaiCommitmentScheme::commit_ntt (commitment_scheme.rs:37-54,75-77),
// Witness::commit (arith.rs:357-362).  f arrives as balanced base-128 digit words (launch_i8g_cut_*: pre [NP][RD][ldw], ldw >= ceil(n / 8)),
// NP = ajtai_i8g_planes_general (an arbitrary element) or ajtai_i8g_planes_i32 (centred coefficients that fit an int32).
uint32_t ajtai_i8g_planes_general(const AjtaiI8Ring &R);
uint32_t ajtai_i8g_planes_i32();
// ... straight from the NTT form of a Goldilocks vector f [24][ld] (dense inverse map icrt_mat [24][24] on the device)
// sp_val / sp_col (optional, [24][8]): the rows of the same map in compressed form when none has more than 8 non-zero entries (column 0xFFFFFFFF = no entry)
void launch_i8g_cut_ntt(const uint64_t *icrt_mat, const uint64_t *sp_val, const uint32_t *sp_col, const uint64_t *ntt, size_t ld, size_t n, uint32_t NP,
                        unsigned long long *pre, size_t ldw, hipStream_t s);
void launch_i8g_cut_i32(const int32_t *planes /* [RD][ld] centred */, size_t ld, size_t n, uint32_t RD, uint32_t NP, unsigned long long *pre, size_t ldw, hipStream_t s);
// scratch sizes (int32 / int32 / int64 words) for a launch with these parameters; -1: shape not handled
int ajtai_i8g_scratch(const AjtaiI8Ring &R, uint32_t MT, size_t n, uint32_t NP, uint32_t nwg, size_t *part_words, size_t *dsum_words, size_t *sum_words);
// rows [row0, row0 + kappa) of the commitment (one packed row chunk of A with MT row tiles) in coefficient form, canonical, into coef_out
// (element row0 + i of kappa_total; SoA [RD][kappa_total] or AoS per R.soa_out).  Returns the grid size or -1.
int launch_ajtai_i8g(const AjtaiI8Ring &R, const unsigned char *Ab, uint32_t MT, const unsigned long long *pre, size_t ldw, size_t n, uint32_t kappa, uint32_t row0,
                     uint32_t kappa_total, uint32_t NP, uint32_t nwg, int32_t *part, int32_t *dsum, long long *sum, uint64_t *coef_out, hipStream_t s);
int ajtai_i8g_read_prof(unsigned long long *out64);
int ajtai_i8g_read_wg(unsigned int *out512);   // (LF_I8G_PROF) loop duration of every workgroup of the last general commit, 100 MHz ticks   // ... of the last general-commit launch made with LF_I8G_PROF set
// measurement: per-phase shader-clock totals of the last launch made with LF_I8_PROF set (out64[8 waves][8]: 7 phases + tile count of workgroup 0)
int ajtai_i8_read_prof(unsigned long long *out64);
// v[k][c][q] = sum_j eq[q][j] * digit_k(planes[c][j]) on the matrix cores (Goldilocks; see lf_ajtai_i8.hip).  mode_bits: K binary digit planes,
// out[(k*24+c)*3+q]; mode 0: the coefficients themselves (|.| <= bound), out[c*3+q].  Returns 0, or -1 if the shape is not handled.
size_t coef_eval_i8_eb_bytes(size_t n);
size_t coef_eval_i8_part_words(uint32_t nwg);
int launch_coef_eval_i8(const int32_t *planes, size_t ldp, size_t n, const uint64_t *eq, size_t ldeq, uint32_t K, int mode_bits, uint64_t bound, unsigned char *EB,
                        uint32_t nwg, int32_t *part, long long *sum /* 24*2*256 words */, uint64_t *out, hipStream_t s);
}  // namespace lf

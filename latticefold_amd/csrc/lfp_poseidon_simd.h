// lfp_poseidon_simd.h -- AVX-512 IFMA lanes for the host side of the LatticeFold+ Fiat-Shamir transcript (PoseidonTranscript<RqPoly> on the Frog ring,
// crates/latticefold-plus/src/transcript.rs:20-78; width 24, 8 full + 22 partial rounds, x^7) over a GENERIC 64-bit prime: Montgomery words with
// R = 2^104, 64x64-bit products as 52-bit limb products (vpmadd52luq / vpmadd52huq) accumulated without carries, one two-step radix-2^52 Montgomery
// reduction per output word; the partial rounds collapsed into two mat-vecs and a scalar chain as in lf_poseidon_simd.h.  A prove at 2^20 rows makes
// ~2 000 permutations on the critical path of the host (set-check challenges, 120 sumcheck rounds, the evaluations' absorb): the scalar form
// (FastPerm, lfp_protocol.cpp) costs 7-12 us each.  Same output as FastPerm::run / permute_plain (tests/test_oracle_lfp_protocol.py, test_abi_cpu.py).
// Selected at run time (cpuid); LFPLUS_POSEIDON_SCALAR=1 forces the scalar path.
#pragma once
#include <stdint.h>

namespace lfp_psimd {

bool supported();   // avx512f + avx512ifma + avx512dq on this CPU
// p and the tables of the sparse-factorised permutation as PLAIN canonical words: ark[(RF+RP)*24], mds[24*24] row-major, cst[RP*24], e00[RP],
// row[RP*23], col[RP*23], post[23*23] row-major
void build(uint64_t p, const uint64_t *ark, const uint64_t *mds, const uint64_t *cst, const uint64_t *e00, const uint64_t *row, const uint64_t *col,
           const uint64_t *post);
void permute(uint64_t st[24]);   // canonical words in and out

}  // namespace lfp_psimd

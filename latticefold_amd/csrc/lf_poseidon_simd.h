// lf_poseidon_simd.h -- AVX-512 IFMA lanes for the host side of the Goldilocks Fiat-Shamir transcript (Poseidon, width 24).
// The transcript is the sequential host part of NIFSProver::prove (transcript/poseidon.rs:29-75); at 2^16..2^20 rows its
// permutations are the exposed host time between GPU phases, so the permutation itself is vectorised: eight state words per
// zmm register, 64x64-bit products as 52-bit limb products (vpmadd52luq/vpmadd52huq) accumulated without carries, one
// reduction mod p = 2^64 - 2^32 + 1 per output word.  The 22 partial rounds are collapsed by linearity into one 22 x 24
// mat-vec, a scalar chain over word 0 (S-box + one multiply per round, lazy 192-bit sums for the cross terms) and one
// closing 24 x 46 mat-vec.  Same output as Transcript::permute_scalar / permute_plain (tested).
// Selected at run time (cpuid); LF_POSEIDON_SCALAR=1 forces the scalar path.
#pragma once
#include <stdint.h>

namespace lf {
namespace psimd {

bool supported();   // avx512f + avx512ifma + avx512dq on this CPU
// tables of the sparse-factorised permutation (lf_host.cpp): ark[(RF+RP)*24], mds[24*24] row-major, cst[RP*24], e00[RP],
// row[RP*23], col[RP*23], post[23*23] row-major
void build(const uint64_t *ark, const uint64_t *mds, const uint64_t *cst, const uint64_t *e00, const uint64_t *row, const uint64_t *col,
           const uint64_t *post);
void permute(uint64_t st[24]);

}  // namespace psimd
}  // namespace lf

// lf_host.h -- host-side pieces of the MI355X LatticeFold prover: ring tables (data-driven CRT), small
// ring operations on a handful of elements, Poseidon/Grain, Fiat-Shamir transcript.
// Everything bulk runs on the GPU (lf_kernels.hip); this file only handles O(proof size) data.
#pragma once
#include <stddef.h>
#include <string>
#include <vector>

#include "lf_field.cuh"

namespace lf {

constexpr int D = 24;      // ring degree, Phi_72 = X^24 - X^12 + 1 (cyclotomic-rings/src/rings/goldilocks.rs:19)
constexpr int SLOTS = 8;   // NTT slots
constexpr int TAU = 3;     // extension degree of each slot
constexpr int RE = 24;     // u64 words per ring element

// Butterfly data for the structured CRT, derived from the (data) cube roots y_k; see lf_ring.cpp.
struct CrtTables {
    u64 nu;           // F_{p^3} non-residue
    int nu_is_2p40;   // fast path flag
    Fq3 y[8];         // image of X in slot k
    // forward butterflies relative to omega = y_0^3
    u64 w4, w2, w10, w1, w7, w5, w11;        // omega^4, omega^2, omega^10, omega, omega^7, omega^5, omega^11
    // inverse butterflies
    u64 inv_1m2w4;                            // 1/(1 - 2 omega^4)
    u64 i2w2, i2w10, i2w1, i2w7, i2w5, i2w11; // 1/(2 omega^e)
    u64 inv2;                                 // 1/2
    // natural butterfly position p -> slot, and per natural position the monomial twist
    int slot_of_pos[8];
    int pos1[8], pos2[8];    // F_{p^3} coordinate receiving A_1 / A_2
    u64 tw1[8], tw2[8];      // multipliers for A_1 / A_2
    u64 itw1[8], itw2[8];    // their inverses
    // dense form for the host (and the generic validation path): slot_k = sum_c a_c * ypow[k][c]
    Fq3 ypow[8][24];
    u64 icrt[24][24];
};

// default tables (NU = 2^40, ascending exponents; see DESIGN.md "CRT map is data")
void default_ring(u64 *nonres, u64 y[24]);
// returns 0 or <0 (LF_ERR_BAD_TABLES)
int build_crt_tables(u64 nonres, const u64 y[24], CrtTables &out);

// ---- host ring helpers on canonical AoS elements (24 words) ------------------------------------------------
struct HostRing {
    CrtTables T;
    Fq3 mul3(Fq3 a, Fq3 b) const { return T.nu_is_2p40 ? fq3_mul<true>(a, b, T.nu) : fq3_mul<false>(a, b, T.nu); }
    Fq3 inv3(Fq3 a) const { return T.nu_is_2p40 ? fq3_inv<true>(a, T.nu) : fq3_inv<false>(a, T.nu); }
    void crt(const u64 *coef, u64 *ntt) const;
    void icrt(const u64 *ntt, u64 *coef) const;
    void mul_ntt(const u64 *a, const u64 *b, u64 *out) const;      // slot-wise
    void mul_fq3(const u64 *a, Fq3 s, u64 *out) const;             // times diagonal embedding
    static void add(const u64 *a, const u64 *b, u64 *out);
    static void sub(const u64 *a, const u64 *b, u64 *out);
    static void from_u64(u64 v, u64 *out);                          // R::from(u128)
    static void from_fq3(Fq3 s, u64 *out);                          // R::from(BaseRing)
    static bool is_diag(const u64 *e, Fq3 *out);
};

// balanced digits of one canonical coefficient (stark_rings::balanced_decomposition)
void balanced_digits(u64 v, u64 base, unsigned digits, int64_t *out, int mode = 0);   // mode: see lf_set_digit_mode

// ---- Poseidon + transcript (crates/latticefold/src/transcript/poseidon.rs:29-75) ------------------------------
class Transcript {
  public:
    Transcript();
    void absorb_fq(const u64 *x, size_t n);
    void absorb_ring(const u64 *elems, size_t count);     // Transcript::absorb / absorb_slice
    void absorb_label(const char *ascii);                 // absorb_field_element(from_be_bytes_mod_order(label))
    void absorb_fq3_as_ring(Fq3 c);                       // absorb(R::from(c))
    void absorb_u64_as_ring(u64 v);                       // absorb(R::from(v as u128))
    Fq3 get_challenge();                                  // squeeze tau words, absorb them back
    void get_short_challenge(u64 coeff_out[24]);          // TranscriptWithShortChallenges (Goldilocks set)
    static void permute(u64 st[24]);        // sparse-factorised partial rounds (same output); AVX-512 IFMA lanes when available
    static void permute_scalar(u64 st[24]); // the same factorisation, scalar (reference for the SIMD path)
    static void permute_plain(u64 st[24]);  // textbook definition, for the self-test
    static void params(const u64 **ark, const u64 **mds);
    // sponge state hand-over to / from the device sponge (lf_kernels.hip): 24 state words, rate index, mode (1 = squeezing)
    void get_state(u64 out[26]) const { for (int i = 0; i < 24; i++) out[i] = st_[i]; out[24] = (u64)idx_; out[25] = squeezing_ ? 1 : 0; }
    // External-basis hook (lf_set_ext_basis): while set, ring elements / challenges handed to absorb_ring / absorb_fq3_as_ring /
    // absorb_u64_as_ring are INTERNAL-basis words and are converted to the external basis before the sponge sees them, and get_challenge
    // returns the internal coordinates of the squeezed (external) challenge.  T, Ti: 3x3 row-major, ext = T int; nullptr = off.
    void set_basis(const u64 *T, const u64 *Ti) { bT_ = T; bTi_ = Ti; }
    void set_state(const u64 in[26]) { for (int i = 0; i < 24; i++) st_[i] = in[i]; idx_ = (int)in[24]; squeezing_ = in[25] != 0; }

    void squeeze(u64 *out, size_t n);   // raw field elements of the sponge (lf_transcript_squeeze_bytes)

  private:
    u64 st_[24];
    bool squeezing_;
    int idx_;
    const u64 *bT_ = nullptr, *bTi_ = nullptr;
};

}  // namespace lf

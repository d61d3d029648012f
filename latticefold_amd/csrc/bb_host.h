// bb_host.h -- host-side pieces of the BabyBearRingNTT backend: ring tables (data-driven CRT), ring operations on a
// handful of elements (canonical u64 words, plain % arithmetic: O(proof size) only), Poseidon + Fiat-Shamir transcript.
// Reference anchors: cyclotomic-rings/src/rings/babybear.rs:1-68 (ring aliases, challenge set),
// rings/poseidon/babybear.rs:7-1425 (Poseidon parameters), latticefold/src/transcript/poseidon.rs:29-75.
#pragma once
#include <stddef.h>
#include <vector>

#include "bb_field.cuh"

namespace lfbb {

struct H9 { u64 c[TAU]; };   // canonical F_{p^9} element on the host

// Tables for the structured CRT.  a(X) = sum_{r<9} X^r A_r(X^9); A_r is evaluated at the 8 primitive 24th roots by the
// same three radix-2 layers as the Goldilocks ring (U^8 - U^4 + 1), then X^r -> tw[r][p] * Y^pos[r][p] in slot_of_pos[p].
struct BbTables {
    u64 nu;                       // F_{p^9} = F_p[Y]/(Y^9 - nu), canonical
    H9 y[8];                      // image of X in slot k
    u64 w4, w2, w10, w1, w7, w5, w11;
    int slot_of_pos[8];
    int pos[TAU][8];              // pos[0][p] = 0
    u64 tw[TAU][8];               // tw[0][p] = 1
    H9 ypow[8][D];                // dense CRT: slot_k = sum_c a_c * ypow[k][c]
    u64 icrt[D][D];               // dense inverse
};
void bb_default_ring(u64 *nonres, u64 y[8 * TAU]);
int bb_build_tables(u64 nonres, const u64 *y, BbTables &out);   // 0 or <0

inline u64 hmul(u64 a, u64 b) { return a * b % BB_P; }
inline u64 hadd(u64 a, u64 b) { u64 r = a + b; return r >= BB_P ? r - BB_P : r; }
inline u64 hsub(u64 a, u64 b) { return a >= b ? a - b : a + BB_P - b; }
u64 hpow(u64 a, u64 e);
inline u64 hinv(u64 a) { return hpow(a, BB_P - 2); }
inline u64 hfrom_i64(int64_t v) { int64_t r = v % (int64_t)BB_P; return (u64)(r < 0 ? r + (int64_t)BB_P : r); }

struct BbHostRing {
    BbTables T;
    H9 mul9(const H9 &a, const H9 &b) const;
    void crt(const u64 *coef, u64 *ntt) const;
    void icrt(const u64 *ntt, u64 *coef) const;
    void mul_ntt(const u64 *a, const u64 *b, u64 *out) const;
    void mul_h9(const u64 *a, const H9 &s, u64 *out) const;
    static void add(const u64 *a, const u64 *b, u64 *out);
    static void sub(const u64 *a, const u64 *b, u64 *out);
    static void from_u64(u64 v, u64 *out);
    static void from_h9(const H9 &s, u64 *out);
};
void bb_balanced_digits(u64 v, u64 base, unsigned digits, int64_t *out, int mode = 0);

class BbTranscript {
  public:
    BbTranscript();
    void absorb_fq(const u64 *x, size_t n);
    void absorb_ring(const u64 *elems, size_t count);
    void absorb_label(const char *ascii);
    void absorb_h9_as_ring(const H9 &c);
    void absorb_u64_as_ring(u64 v);
    H9 get_challenge();                            // squeeze tau words, absorb them back
    void get_short_challenge(u64 coeff_out[D]);    // 18 bytes -> 24 coefficients in [-32,32), zero-padded to degree 72
    static void permute(u64 st[24]);               // sparse-factorised partial rounds; AVX2 Montgomery lanes when available
    static void permute_scalar(u64 st[24]);        // same factorisation, scalar (reference for the SIMD path)
    static void permute_plain(u64 st[24]);
    static void params(const u64 **ark, const u64 **mds);
    // external-basis hook, as lf::Transcript::set_basis (T, Ti: 9x9 row-major, ext = T int; nullptr = off)
    void set_basis(const u64 *T, const u64 *Ti) { bT_ = T; bTi_ = Ti; }

    void squeeze(u64 *out, size_t n);   // raw field elements of the sponge (lf_transcript_squeeze_bytes)

  private:
    u64 st_[24];
    bool squeezing_;
    int idx_;
    const u64 *bT_ = nullptr, *bTi_ = nullptr;
};

}  // namespace lfbb

// lf_wire.cpp -- byte layout of an LFProof for interchange with the reference's verifier ("wire format").
//
// The reference derives CanonicalSerialize for the whole proof tree (nifs.rs:28-34, linearization/structs.rs:14-37,
// decomposition/structs.rs:18-50, folding/structs.rs:17-40, utils/sumcheck.rs:41-42, sumcheck/prover.rs:13-17,
// commitment/homomorphic_commitment.rs:11-14) and round-trips it with Compress::Yes (folding/tests/mod.rs:656-680).  With the
// ark-serialize 0.4 rules that derive means: struct = fields in declaration order; Vec<T> = u64 little-endian length, then the
// elements; a prime-field element = its canonical value in ceil(bits/8) little-endian bytes (compression only affects curve
// points, so Compress::Yes and ::No coincide here).
//
// LAYOUT ASSUMPTION (parity unpinned, like the CRT tables -- DESIGN.md 6): the serializer of a ring element lives in the
// un-vendored stark-rings crate.  It is taken to be the element's d base-field words in the order the transcript absorbs them
// (slot-major NTT coefficients, the flat order of this ABI) with no length prefix (a fixed-size array), 8 bytes per word for
// both rings (ark Fp64).  tools/probe_stark_rings.rs prints one serialized element on a machine with the Rust toolchain.
#include <string.h>

#include "../../include/lfhip.h"
#include "bb_field.cuh"
#include "lf_host.h"

namespace {
typedef uint64_t u64;

struct Shape {
    size_t s, d, t, K, l, kappa, b, tau, re;
    u64 mod;
};
bool shape_of(const lf_params *p, int ring, Shape &sh) {
    if (!p || (ring != LF_RING_GOLDILOCKS && ring != LF_RING_BABYBEAR)) return false;
    sh.s = p->s; sh.d = p->d; sh.t = p->t; sh.K = p->K; sh.l = p->l; sh.kappa = p->kappa; sh.b = p->b;
    sh.tau = ring == LF_RING_BABYBEAR ? lfbb::TAU : lf::TAU;
    sh.re = ring == LF_RING_BABYBEAR ? lfbb::RE : lf::RE;
    sh.mod = ring == LF_RING_BABYBEAR ? (u64)lfbb::BB_P : LF_P;
    return true;
}

// One pass over the proof tree; `Io` either counts, writes or reads.  The flat proof (include/lfhip.h, lf_fold_step) already has
// the reference's field order, so the walk only inserts / checks the Vec length prefixes.
template <class Io>
bool walk(const Shape &sh, Io &io) {
    auto vec = [&](size_t n) { return io.len(n); };
    auto elems = [&](size_t n) { return io.elems(n * sh.re); };
    auto vecvec = [&](size_t outer, size_t inner) {
        if (!vec(outer)) return false;
        for (size_t i = 0; i < outer; i++)
            if (!vec(inner) || !elems(inner)) return false;
        return true;
    };
    // LinearizationProof { linearization_sumcheck: Proof(Vec<ProverMsg { evaluations: Vec<NTT> }>), v: Vec<NTT>, u: Vec<NTT> }
    if (!vecvec(sh.s, sh.d + 2)) return false;
    if (!vec(sh.tau) || !elems(sh.tau) || !vec(sh.t) || !elems(sh.t)) return false;
    // DecompositionProof { u_s: Vec<Vec<NTT>>, v_s: Vec<Vec<NTT>>, x_s: Vec<Vec<NTT>>, y_s: Vec<Commitment { val: Vec<NTT> }> }, left then right
    for (int side = 0; side < 2; side++)
        if (!vecvec(sh.K, sh.t) || !vecvec(sh.K, sh.tau) || !vecvec(sh.K, sh.l + 1) || !vecvec(sh.K, sh.kappa)) return false;
    // FoldingProof { pointshift_sumcheck_proof: Proof, theta_s: Vec<Vec<NTT>>, eta_s: Vec<Vec<NTT>> }
    return vecvec(sh.s, 2 * sh.b + 1) && vecvec(2 * sh.K, sh.tau) && vecvec(2 * sh.K, sh.t);
}

struct Counter {
    size_t bytes = 0;
    bool len(size_t) { bytes += 8; return true; }
    bool elems(size_t words) { bytes += 8 * words; return true; }
};
struct Writer {
    const u64 *src;
    uint8_t *dst;
    u64 mod;
    static void put(uint8_t *&d, u64 v) { for (int i = 0; i < 8; i++) *d++ = (uint8_t)(v >> (8 * i)); }
    bool len(size_t n) { put(dst, (u64)n); return true; }
    bool elems(size_t words) {
        for (size_t i = 0; i < words; i++) {
            if (src[i] >= mod) return false;            // not a canonical residue
            put(dst, src[i]);
        }
        src += words;
        return true;
    }
};
struct Reader {
    const uint8_t *src, *end;
    u64 *dst;
    u64 mod;
    bool get(u64 &v) {
        if (end - src < 8) return false;
        v = 0;
        for (int i = 0; i < 8; i++) v |= (u64)src[i] << (8 * i);
        src += 8;
        return true;
    }
    bool len(size_t n) { u64 v; return get(v) && v == (u64)n; }     // Vec lengths are fixed by the parameters
    bool elems(size_t words) {
        for (size_t i = 0; i < words; i++) {
            u64 v;
            if (!get(v) || v >= mod) return false;                    // Validate::Yes: canonical field elements only
            *dst++ = v;
        }
        return true;
    }
};
}  // namespace

extern "C" {
size_t lf_proof_wire_size(const lf_params *p, int ring) {
    Shape sh;
    if (!shape_of(p, ring, sh)) return 0;
    Counter c;
    walk(sh, c);
    return c.bytes;
}
int lf_proof_serialize(const lf_params *p, int ring, const uint64_t *proof, uint8_t *out, size_t cap) {
    Shape sh;
    if (!proof || !out || !shape_of(p, ring, sh)) return LF_ERR_INVALID;
    if (cap < lf_proof_wire_size(p, ring)) return LF_ERR_INVALID;
    Writer w{proof, out, sh.mod};
    return walk(sh, w) ? LF_OK : LF_ERR_INVALID;
}
int lf_proof_deserialize(const lf_params *p, int ring, const uint8_t *in, size_t len, uint64_t *proof) {
    Shape sh;
    if (!in || !proof || !shape_of(p, ring, sh)) return LF_ERR_INVALID;
    Reader r{in, in + len, proof, sh.mod};
    if (!walk(sh, r)) return LF_ERR_INVALID;
    return r.src == r.end ? LF_OK : LF_ERR_INVALID;                   // trailing bytes are an error
}
}

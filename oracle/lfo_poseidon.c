/* lfo_poseidon.c -- ORACLE (test infrastructure only): Poseidon parameters (Grain LFSR),
 * permutation, arkworks-0.4 duplex sponge and the LatticeFold PoseidonTranscript.
 *
 * Reference: crates/latticefold/src/transcript/poseidon.rs:29-75 (transcript),
 * crates/cyclotomic-rings/src/rings/poseidon/goldilocks.rs:7-1425 (constants: width 24 =
 * rate 20 + capacity 4, 8 full + 22 partial rounds, alpha 7).  The constants are NOT copied:
 * they are regenerated with the published Poseidon Grain-LFSR procedure
 * (ark-crypto-primitives 0.4.0 `find_poseidon_ark_and_mds(64, 23, 8, 22, 0)`) and checked
 * against spot values + checksums of the reference table in tests/golden/kats.json.
 * Sponge model: ark-crypto-primitives 0.4.0 `PoseidonSponge` (absent dependency), KAT-pinned
 * by transcript/poseidon.rs:85-142. */
#include "lfo_field.h"
#include <stdlib.h>

#define W 24
#define RATE 20
#define CAP 4
#define RF 8
#define RP 22

static u64 g_ark[(RF + RP) * W];
static u64 g_mds[W * W];
static int g_pinit = 0;

/* ---- Grain LFSR (Poseidon paper, Appendix; arkworks grain_lfsr.rs) ---------------------- */
typedef struct { unsigned char st[80]; int head; } grain;
static int grain_update(grain *g) {
    int h = g->head;
    unsigned char nb = g->st[(h + 62) % 80] ^ g->st[(h + 51) % 80] ^ g->st[(h + 38) % 80] ^
                       g->st[(h + 23) % 80] ^ g->st[(h + 13) % 80] ^ g->st[h];
    g->st[h] = nb;
    g->head = (h + 1) % 80;
    return nb;
}
static void grain_put(grain *g, int lo, int hi, u64 v) {
    for (int i = hi; i >= lo; i--) { g->st[i] = v & 1; v >>= 1; }
}
static void grain_init(grain *g, u64 prime_bits, u64 width, u64 rf, u64 rp) {
    memset(g, 0, sizeof(*g));
    g->st[1] = 1;             /* field = prime field */
    /* bits 2..5: s-box = x^alpha (not inverse) -> 0 */
    grain_put(g, 6, 17, prime_bits);
    grain_put(g, 18, 29, width);
    grain_put(g, 30, 39, rf);
    grain_put(g, 40, 49, rp);
    for (int i = 50; i < 80; i++) g->st[i] = 1;
    for (int i = 0; i < 160; i++) grain_update(g);
}
static u64 grain_bits64(grain *g) { /* 64 filtered bits, most significant first */
    u64 v = 0;
    for (int i = 0; i < 64; i++) {
        int nb = grain_update(g);
        while (!nb) { grain_update(g); nb = grain_update(g); }
        v = (v << 1) | (u64)grain_update(g);
    }
    return v;
}

static void poseidon_init(void) {
    if (g_pinit) return;
    grain g;
    grain_init(&g, 64, W, RF, RP);
    /* The reference's BabyBear table (rings/poseidon/babybear.rs) holds the SAME 64-bit constants as the
     * Goldilocks one (generated for a 64-bit prime), embedded with Fq::from(i128), i.e. reduced mod p_BB:
     * so generate over Goldilocks and reduce. */
    const u64 PG = 0xFFFFFFFF00000001ULL;
    for (int i = 0; i < (RF + RP) * W; i++) { /* rejection sampling */
        u64 v;
        do v = grain_bits64(&g); while (v >= PG);
        g_ark[i] = v % LFO_P;
    }
    u64 xs[W], ys[W];
    for (int i = 0; i < W; i++) xs[i] = grain_bits64(&g) % PG;
    for (int i = 0; i < W; i++) ys[i] = grain_bits64(&g) % PG;
    for (int i = 0; i < W; i++)
        for (int j = 0; j < W; j++) { /* Cauchy over Goldilocks */
            u128 sm = ((u128)xs[i] + ys[j]) % PG, r = 1, bs = sm;
            for (u64 e = PG - 2; e; e >>= 1) { if (e & 1) r = (u128)(((u128)(u64)r * (u64)bs) % PG); bs = (u128)(((u128)(u64)bs * (u64)bs) % PG); }
            g_mds[i * W + j] = (u64)r % LFO_P;
        }
    g_pinit = 1;
}

void lfo_poseidon_params(u64 *ark, u64 *mds) {
    poseidon_init();
    memcpy(ark, g_ark, sizeof(g_ark));
    memcpy(mds, g_mds, sizeof(g_mds));
}

static inline u64 pow7(u64 x) {
    u64 x2 = fq_mul(x, x), x4 = fq_mul(x2, x2);
    return fq_mul(fq_mul(x4, x2), x);
}

void lfo_poseidon_permute(u64 *st) {
    poseidon_init();
    u64 nw[W];
    for (int r = 0; r < RF + RP; r++) {
        for (int i = 0; i < W; i++) st[i] = fq_add(st[i], g_ark[r * W + i]);
        if (r < RF / 2 || r >= RF / 2 + RP)
            for (int i = 0; i < W; i++) st[i] = pow7(st[i]);
        else
            st[0] = pow7(st[0]);
        for (int i = 0; i < W; i++) {
            u128 acc = 0; /* sum of 24 products < 2^133: reduce every term instead */
            u64 a = 0;
            (void)acc;
            for (int j = 0; j < W; j++) a = fq_add(a, fq_mul(st[j], g_mds[i * W + j]));
            nw[i] = a;
        }
        memcpy(st, nw, sizeof(nw));
    }
}

/* ---- duplex sponge ------------------------------------------------------------------------ */
struct lfo_transcript {
    u64 st[W];
    int squeezing; /* mode */
    int idx;       /* next_absorb_index / next_squeeze_index */
};

lfo_transcript *lfo_transcript_new(void) {
    poseidon_init();
    lfo_transcript *t = (lfo_transcript *)calloc(1, sizeof(*t));
    return t;
}
void lfo_transcript_free(lfo_transcript *t) { free(t); }

void lfo_transcript_absorb_fq(lfo_transcript *t, const u64 *x, size_t n) {
    if (n == 0) return;
    int idx;
    if (!t->squeezing) {
        idx = t->idx;
        if (idx == RATE) { lfo_poseidon_permute(t->st); idx = 0; }
    } else {
        lfo_poseidon_permute(t->st);
        idx = 0;
    }
    for (;;) {
        if ((size_t)idx + n <= RATE) {
            for (size_t i = 0; i < n; i++) t->st[CAP + idx + i] = fq_add(t->st[CAP + idx + i], x[i]);
            t->squeezing = 0;
            t->idx = idx + (int)n;
            return;
        }
        size_t take = RATE - idx;
        for (size_t i = 0; i < take; i++) t->st[CAP + idx + i] = fq_add(t->st[CAP + idx + i], x[i]);
        lfo_poseidon_permute(t->st);
        x += take;
        n -= take;
        idx = 0;
    }
}

static void squeeze_fq(lfo_transcript *t, u64 *out, size_t n) {
    int idx;
    if (!t->squeezing) {
        lfo_poseidon_permute(t->st);
        idx = 0;
    } else {
        idx = t->idx;
        if (idx == RATE) { lfo_poseidon_permute(t->st); idx = 0; }
    }
    for (;;) {
        if ((size_t)idx + n <= RATE) {
            memcpy(out, t->st + CAP + idx, n * sizeof(u64));
            t->squeezing = 1;
            t->idx = idx + (int)n;
            return;
        }
        size_t take = RATE - idx;
        memcpy(out, t->st + CAP + idx, take * sizeof(u64));
        if (n != RATE) lfo_poseidon_permute(t->st); /* arkworks: "unless we are done ... permute" */
        out += take;
        n -= take;
        idx = 0;
    }
}

/* Transcript::absorb(R): the 24 base-field words of each element, slot-major */
void lfo_transcript_absorb_ring(lfo_transcript *t, const u64 *e, size_t count) {
    for (size_t i = 0; i < count; i++) lfo_transcript_absorb_fq(t, e + (size_t)RE * i, RE);
}

/* get_challenge: squeeze tau words, absorb them back (poseidon.rs:49-57) */
void lfo_transcript_get_challenge(lfo_transcript *t, u64 *out) {
    squeeze_fq(t, out, TAU);
    lfo_transcript_absorb_fq(t, out, TAU);
}

/* squeeze_bytes(18) (ark-crypto-primitives 0.4.0 PoseidonSponge): usable = (MODULUS_BIT_SIZE-1)/8 low LE
 * bytes per element (7 Goldilocks, 3 BabyBear), ceil(18/usable) elements; then the challenge-set decoder */
void lfo_transcript_get_short_challenge(lfo_transcript *t, u64 *coeff_out) {
    enum { UB = (LFO_MOD_BITS - 1) / 8, NE = (18 + UB - 1) / UB };
    u64 e[NE];
    uint8_t bs[NE * UB];
    squeeze_fq(t, e, NE);
    for (int i = 0; i < NE; i++)
        for (int j = 0; j < UB; j++) bs[UB * i + j] = (uint8_t)(e[i] >> (8 * j));
    lfo_short_challenge_from_bytes(bs, 18, coeff_out);
}

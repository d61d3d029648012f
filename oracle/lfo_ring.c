/* lfo_ring.c -- ORACLE (test infrastructure only): ring tables, CRT/ICRT, balanced
 * decomposition, RotSum, short challenges.  See lfo.h for the parity statement. */
#include "lfo_field.h"
#include <stdlib.h>

u64 lfo_NONRES = 0; /* set by default_ring(): a primitive 24th root of unity (Goldilocks: 2^40) */

int lfo_ring_degree(void) { return LFO_D; }
int lfo_ring_tau(void) { return LFO_TAU; }
u64 lfo_modulus(void) { return LFO_P; }

static int g_digit_mode = 0;
const u64 *lfo_TENSOR = NULL;
static u64 g_tensor[TAU * TAU * TAU];
static int g_general = 0;        /* CRT given as a dense d x d matrix (lfo_set_ring_general) */
static u64 g_crt[RE][RE];
static int g_init = 0;
static fqe g_y[8];         /* image of X in slot k */
static fqe g_ypow[8][RE];  /* y_k^i */
static u64 g_icrt[RE][RE]; /* inverse of the d x d F_p matrix of CRT */

static fqe fqe_pow_small(fqe a, unsigned e) {
    fqe r = fqe_one();
    while (e--) r = fqe_mul(r, a);
    return r;
}

static int build_tables(void) {
    /* CRT as an F_p-linear map: out[3k+c] = sum_i a_i * (y_k^i).c   */
    static u64 M[RE][2 * RE];
    if (g_general) {
        for (int r = 0; r < RE; r++)
            for (int i = 0; i < RE; i++) M[r][i] = g_crt[r][i];
    } else
    for (int k = 0; k < 8; k++) {
        fqe p = fqe_one();
        for (int i = 0; i < RE; i++) {
            g_ypow[k][i] = p;
            for (int c = 0; c < TAU; c++) M[TAU * k + c][i] = p.c[c];
            p = fqe_mul(p, g_y[k]);
        }
    }
    for (int r = 0; r < RE; r++)
        for (int c = 0; c < RE; c++) M[r][RE + c] = (r == c);
    /* Gauss-Jordan over F_p */
    for (int col = 0; col < RE; col++) {
        int piv = -1;
        for (int r = col; r < RE; r++)
            if (M[r][col]) { piv = r; break; }
        if (piv < 0) return -1; /* not an isomorphism */
        if (piv != col)
            for (int c = 0; c < 2 * RE; c++) { u64 t = M[piv][c]; M[piv][c] = M[col][c]; M[col][c] = t; }
        u64 inv = fq_inv(M[col][col]);
        for (int c = 0; c < 2 * RE; c++) M[col][c] = fq_mul(M[col][c], inv);
        for (int r = 0; r < RE; r++) {
            if (r == col || !M[r][col]) continue;
            u64 f = M[r][col];
            for (int c = 0; c < 2 * RE; c++) M[r][c] = fq_sub(M[r][c], fq_mul(f, M[col][c]));
        }
    }
    for (int r = 0; r < RE; r++)
        for (int c = 0; c < RE; c++) g_icrt[r][c] = M[r][RE + c];
    return 0;
}

/* smallest primitive 24th root of unity found from generator candidates (BabyBear); Goldilocks uses 2^40 */
static u64 default_zeta(void) {
#ifdef LFO_RING_BABYBEAR
    for (u64 g = 2;; g++) {
        u64 z = fq_pow(g, (LFO_P - 1) / 24);
        if (fq_pow(z, 12) != 1 && fq_pow(z, 8) != 1) return z; /* order exactly 24 */
    }
#else
    return 1ULL << 40;
#endif
}

static void default_ring(void) {
    /* Phi(X) = prod_{e in (Z/24)^*} (X^tau - zeta^e), zeta of order 24, tau in {3, 9}; slots in ascending e.
     * Goldilocks: F_{p^3} = F_p[Y]/(Y^3 - zeta), zeta = 2^40; slot e = 1 mod 3 uses X -> zeta^a * Y, e = 2 mod 3 uses
     * X -> zeta^a * Y^2 with 3a = e - g (the map used since round 1).
     * BabyBear: F_{p^9} = F_p[Y]/(Y^9 - 2) (2 is a non-cube mod p, so the binomial is irreducible); slot e maps X -> c * Y^g
     * with g in {1,2} the class for which zeta^e / 2^g is a cube and c its 9th root inside the cube subgroup
     * (order (p-1)/3 = 5*2^27, coprime to 9). */
    static const int E[8] = {1, 5, 7, 11, 13, 17, 19, 23};
    u64 zeta = default_zeta();
#ifdef LFO_RING_BABYBEAR
    lfo_NONRES = 2;
    const u64 sub = (LFO_P - 1) / 3;
    u64 e9 = 0;
    for (u64 k = 1; k < 9; k++)
        if ((k * sub + 1) % 9 == 0) { e9 = (k * sub + 1) / 9; break; }
    for (int k = 0; k < 8; k++) {
        u64 ze = fq_pow(zeta, (u64)E[k]);
        fqe y = fqe_zero();
        for (int g = 1; g <= 2; g++) {
            u64 w = fq_mul(ze, fq_inv(fq_pow(2, (u64)g)));
            if (fq_pow(w, sub) != 1) continue;
            y.c[g] = fq_pow(w, e9);
            break;
        }
        g_y[k] = y;
    }
#else
    lfo_NONRES = zeta;
    for (int k = 0; k < 8; k++) {
        int e = E[k], g = e % 3, a = -1;
        for (int t = 0; t < 24; t++)
            if ((TAU * t) % 24 == (e - g) % 24) { a = t; break; }
        fqe y = fqe_zero();
        y.c[g] = fq_pow(zeta, (u64)a);
        g_y[k] = y;
    }
#endif
}

/* runs when the library is loaded: lfo_NONRES and the ring tables are valid before any entry point is used
 * (the protocol functions multiply in F_{p^tau} before they ever reach lfo_crt) */
static void ensure_init(void) __attribute__((constructor));
static void ensure_init(void) {
    if (g_init) return;
    default_ring();
    if (build_tables() != 0) abort();
    g_init = 1;
}

/* SURVEY 8(c) "conventions are data", the fully general form: CRT as a dense d x d matrix over F_p (out[r] = sum_c crt[r*d + c] coef[c],
 * rows in slot-major coordinate order) and the structure constants of F_{p^tau} in the caller's basis (e_0 = 1).  Checked: e_0 is the
 * unit, the map is invertible and multiplicative on X * X^j (ring homomorphism on the monomials). */
int lfo_set_ring_general(const u64 *crt, const u64 *tensor) {
    ensure_init();
    for (int j = 0; j < TAU; j++)
        for (int k = 0; k < TAU; k++)
            if (tensor[(size_t)(0 * TAU + j) * TAU + k] % LFO_P != (u64)(j == k) || tensor[(size_t)(j * TAU + 0) * TAU + k] % LFO_P != (u64)(j == k)) return -1;
    for (int i = 0; i < TAU * TAU * TAU; i++) g_tensor[i] = tensor[i] % LFO_P;
    for (int r = 0; r < RE; r++)
        for (int c = 0; c < RE; c++) g_crt[r][c] = crt[(size_t)r * RE + c] % LFO_P;
    lfo_TENSOR = g_tensor;
    g_general = 1;
    if (build_tables() != 0) { lfo_TENSOR = NULL; g_general = 0; build_tables(); return -1; }
    /* CRT(X^i) (.) CRT(X) == CRT(X^(i+1)) for i + 1 < d */
    for (int i = 0; i + 1 < RE; i++)
        for (int k = 0; k < 8; k++) {
            fqe a, x, w;
            for (int c = 0; c < TAU; c++) { a.c[c] = g_crt[TAU * k + c][i]; x.c[c] = g_crt[TAU * k + c][1]; w.c[c] = g_crt[TAU * k + c][i + 1]; }
            fqe pr = fqe_mul(a, x);
            if (memcmp(pr.c, w.c, sizeof(pr.c)) != 0) { lfo_TENSOR = NULL; g_general = 0; build_tables(); return -2; }
        }
    return 0;
}

int lfo_set_ring(u64 nonres, const u64 *y) {
    ensure_init();
    lfo_TENSOR = NULL;
    g_general = 0;
    u64 old_nr = lfo_NONRES;
    fqe old_y[8];
    memcpy(old_y, g_y, sizeof(old_y));
    lfo_NONRES = nonres % LFO_P;
    int ok = 1;
    u64 roots[8];
    for (int k = 0; k < 8 && ok; k++) {
        fqe v;
        for (int c = 0; c < TAU; c++) v.c[c] = y[TAU * k + c] % LFO_P;
        g_y[k] = v;
        fqe cube = fqe_pow_small(v, TAU);
        for (int c = 1; c < TAU; c++) if (cube.c[c]) ok = 0;
        u64 z = cube.c[0];
        roots[k] = z;
        /* z must be a root of Phi_24(Y) = Y^8 - Y^4 + 1 */
        u64 z4 = fq_pow(z, 4), z8 = fq_mul(z4, z4);
        if (fq_add(fq_sub(z8, z4), 1) != 0) ok = 0;
        for (int j = 0; j < k; j++)
            if (roots[j] == z) ok = 0;
    }
    if (ok && build_tables() != 0) ok = 0;
    if (!ok) {
        lfo_NONRES = old_nr;
        memcpy(g_y, old_y, sizeof(old_y));
        build_tables();
        return -1;
    }
    return 0;
}

void lfo_get_ring(u64 *nonres, u64 *y) {
    ensure_init();
    *nonres = lfo_NONRES;
    for (int k = 0; k < 8; k++)
        for (int c = 0; c < TAU; c++) y[TAU * k + c] = g_y[k].c[c];
}

void lfo_set_digit_mode(int mode) { g_digit_mode = mode; }

void lfo_fq3_mul(const u64 *a, const u64 *b, u64 *out) { /* F_{p^tau} product (name kept from the Goldilocks build) */
    ensure_init();
    fqe_store(out, fqe_mul(fqe_load(a), fqe_load(b)));
}

/* CRT: slot_k = a(y_k) evaluated in F_{p^3} */
void lfo_crt(const u64 *in, u64 *out, size_t count) {
    ensure_init();
#pragma omp parallel for schedule(static) if (count >= 4096)
    for (size_t e = 0; e < count; e++) {
        const u64 *a = in + RE * e;
        u64 res[RE];
        if (g_general) {
            for (int r = 0; r < RE; r++) {
                u64 acc = 0;
                for (int i = 0; i < RE; i++)
                    if (a[i]) acc = fq_add(acc, fq_mul(g_crt[r][i], a[i]));
                res[r] = acc;
            }
        } else
        for (int k = 0; k < 8; k++) {
            fqe acc = fqe_zero();
            for (int i = 0; i < RE; i++)
                if (a[i]) acc = fqe_add(acc, fqe_mul_fq(g_ypow[k][i], a[i]));
            rq_set_slot(res, k, acc);
        }
        memcpy(out + RE * e, res, sizeof(res));
    }
}

void lfo_icrt(const u64 *in, u64 *out, size_t count) {
    ensure_init();
#pragma omp parallel for schedule(static) if (count >= 4096)
    for (size_t e = 0; e < count; e++) {
        const u64 *x = in + RE * e;
        u64 res[RE];
        for (int i = 0; i < RE; i++) {
            u64 acc = 0;
            for (int j = 0; j < RE; j++)
                if (x[j]) acc = fq_add(acc, fq_mul(g_icrt[i][j], x[j]));
            res[i] = acc;
        }
        memcpy(out + RE * e, res, sizeof(res));
    }
}

void lfo_ring_mul_ntt(const u64 *a, const u64 *b, u64 *out, size_t count) {
    ensure_init();
    for (size_t e = 0; e < count; e++) rq_mul(out + RE * e, a + RE * e, b + RE * e);
}

/* multiply by X in Z_p[X]/(X^d - X^(d/2) + 1) */
static void rot_x(u64 *a) {
    u64 top = a[RE - 1];
    for (int i = RE - 1; i > 0; i--) a[i] = a[i - 1];
    a[0] = fq_neg(top);
    a[RE / 2] = fq_add(a[RE / 2], top);
}

void lfo_ring_mul_coeff(const u64 *a, const u64 *b, u64 *out) {
    u64 rot[RE], acc[RE] = {0};
    memcpy(rot, a, sizeof(rot));
    for (int i = 0; i < RE; i++) {
        for (int j = 0; j < RE; j++) acc[j] = fq_add(acc[j], fq_mul(rot[j], b[i]));
        rot_x(rot);
    }
    memcpy(out, acc, sizeof(acc));
}

/* ---- balanced decomposition ------------------------------------------------------------
 * stark_rings::balanced_decomposition (source absent).  mode 0 restates the published
 * lattirust/stark-rings algorithm as recollected: lift to the signed representative in
 * [-(p-1)/2, (p-1)/2]; repeatedly rem = curr % b (C/Rust truncating remainder); if
 * |rem| <= b/2 keep it and curr /= b, else digit = rem -+ b and curr = curr/b +- 1; pad with
 * zeros to `digits`.  mode 1: floor division, digits in [-b/2, b/2).  UNPINNED (lfo.h). */
static void decompose_coeff(u64 v, u64 base, u32 digits, int64_t *out) {
    __int128 b = (__int128)base;
    __int128 half = b / 2;
    __int128 curr = v <= (LFO_P - 1) / 2 ? (__int128)v : (__int128)v - (__int128)LFO_P;
    for (u32 k = 0; k < digits; k++) {
        __int128 rem, q;
        if (g_digit_mode == 0 || b == 2) { /* base 2: sign and bits of the magnitude under either mode (the floor rule does not terminate) */
            rem = curr % b;
            q = curr / b;
            __int128 arem = rem < 0 ? -rem : rem;
            if (arem > half) {
                if (rem < 0) { rem += b; q -= 1; }
                else { rem -= b; q += 1; }
            }
        } else {
            rem = curr % b;
            if (rem < 0) rem += b;
            if (rem >= half) rem -= b;
            q = (curr - rem) / b;
        }
        out[k] = (int64_t)rem;
        curr = q;
    }
}

void lfo_decompose(const u64 *in, size_t count, u64 base, u32 digits, int layout, u64 *out) {
#pragma omp parallel for schedule(static) if (count >= 4096)
    for (size_t e = 0; e < count; e++) {
        int64_t dg[64];
        for (int c = 0; c < RE; c++) {
            decompose_coeff(in[RE * e + c], base, digits, dg);
            for (u32 k = 0; k < digits; k++) {
                size_t idx = layout == 0 ? e * digits + k : (size_t)k * count + e;
                out[RE * idx + c] = fq_from_i64(dg[k]);
            }
        }
    }
}

void lfo_recompose(const u64 *in, size_t count_out, u64 base, u32 digits, u64 *out) {
#pragma omp parallel for schedule(static) if (count_out >= 4096)
    for (size_t e = 0; e < count_out; e++) {
        u64 acc[RE] = {0};
        u64 pw = 1;
        for (u32 j = 0; j < digits; j++) {
            const u64 *x = in + RE * (e * digits + j);
            for (int c = 0; c < RE; c++) acc[c] = fq_add(acc[c], fq_mul(x[c], pw));
            pw = fq_mul(pw, base % LFO_P);
        }
        memcpy(out + RE * e, acc, sizeof(acc));
    }
}

/* rot_sum / rot_lin_combination, cyclotomic-rings/src/rotation.rs:45-104.
 * theta: n x tau_elems NTT-form ring elements; flatten_to_coeffs = concatenation of slots
 * (KAT-verified); result: tau_elems NTT-form elements. */
void lfo_rot_lin_combination(const u64 *rho_coeff, const u64 *theta, u32 n, u32 tau_elems, u64 *out) {
    ensure_init();
    u32 flat = tau_elems * 8; /* number of F_{p^tau} entries; must equal the ring degree */
    static fqe res[RE];
    for (int j = 0; j < RE; j++) res[j] = fqe_zero();
    for (u32 i = 0; i < n; i++) {
        u64 rot[RE];
        memcpy(rot, rho_coeff + (size_t)RE * i, sizeof(rot));
        const u64 *th = theta + (size_t)RE * tau_elems * i;
        for (u32 bi = 0; bi < flat && bi < RE; bi++) {
            fqe b = fqe_load(th + TAU * bi);
            for (int j = 0; j < RE; j++) res[j] = fqe_add(res[j], fqe_mul_fq(b, rot[j]));
            rot_x(rot);
        }
    }
    for (int j = 0; j < RE; j++) fqe_store(out + TAU * j, res[j]);
}

/* {Goldilocks,BabyBear}ChallengeSet::short_challenge_from_random_bytes, rings/goldilocks.rs:36-68,
 * rings/babybear.rs:36-68: 24 six-bit fields, LSB-first within each 3-byte group, each minus 32;
 * BabyBear builds the degree-72 polynomial from these 24 coefficients (higher ones zero). */
int lfo_short_challenge_from_bytes(const uint8_t *bs, size_t n, u64 *coeff_out) {
    if (n != 18) return -1;
    memset(coeff_out, 0, RE * sizeof(u64));
    for (int g = 0; g < 6; g++) {
        u32 w = (u32)bs[3 * g] | ((u32)bs[3 * g + 1] << 8) | ((u32)bs[3 * g + 2] << 16);
        for (int j = 0; j < 4; j++) coeff_out[4 * g + j] = fq_from_i64((int64_t)((w >> (6 * j)) & 63) - 32);
    }
    return 0;
}

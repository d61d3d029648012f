/* lfo_ring.c -- ORACLE (test infrastructure only): ring tables, CRT/ICRT, balanced
 * decomposition, RotSum, short challenges.  See lfo.h for the parity statement. */
#include "lfo_field.h"
#include <stdlib.h>

u64 lfo_NONRES = 1ULL << 40; /* default: 2^40, a primitive 24th root of unity (order checked in init) */

static int g_digit_mode = 0;
static int g_init = 0;
static fq3 g_y[8];        /* image of X in slot k */
static fq3 g_ypow[8][24]; /* y_k^i */
static u64 g_icrt[24][24]; /* inverse of the 24x24 F_p matrix of CRT */

static fq3 fq3_pow_small(fq3 a, unsigned e) {
    fq3 r = fq3_one();
    while (e--) r = fq3_mul(r, a);
    return r;
}

static int build_tables(void) {
    /* CRT as an F_p-linear map: out[3k+c] = sum_i a_i * (y_k^i).c   */
    static u64 M[24][48];
    for (int k = 0; k < 8; k++) {
        fq3 p = fq3_one();
        for (int i = 0; i < 24; i++) {
            g_ypow[k][i] = p;
            for (int c = 0; c < 3; c++) M[3 * k + c][i] = p.c[c];
            p = fq3_mul(p, g_y[k]);
        }
    }
    for (int r = 0; r < 24; r++)
        for (int c = 0; c < 24; c++) M[r][24 + c] = (r == c);
    /* Gauss-Jordan over F_p */
    for (int col = 0; col < 24; col++) {
        int piv = -1;
        for (int r = col; r < 24; r++)
            if (M[r][col]) { piv = r; break; }
        if (piv < 0) return -1; /* not an isomorphism */
        if (piv != col)
            for (int c = 0; c < 48; c++) { u64 t = M[piv][c]; M[piv][c] = M[col][c]; M[col][c] = t; }
        u64 inv = fq_inv(M[col][col]);
        for (int c = 0; c < 48; c++) M[col][c] = fq_mul(M[col][c], inv);
        for (int r = 0; r < 24; r++) {
            if (r == col || !M[r][col]) continue;
            u64 f = M[r][col];
            for (int c = 0; c < 48; c++) M[r][c] = fq_sub(M[r][c], fq_mul(f, M[col][c]));
        }
    }
    for (int r = 0; r < 24; r++)
        for (int c = 0; c < 24; c++) g_icrt[r][c] = M[r][24 + c];
    return 0;
}

static void default_ring(void) {
    /* Phi_72(X) = prod_{e in (Z/24)^*} (X^3 - zeta^e), zeta = 2^40 (order 24).  With
     * F_{p^3} = F_p[Y]/(Y^3 - zeta):  slot for e = 1 mod 3 uses X -> zeta^((e-1)/3) * Y,
     * slot for e = 2 mod 3 uses X -> zeta^((e-2)/3) * Y^2.  Slots in ascending e. */
    static const int E[8] = {1, 5, 7, 11, 13, 17, 19, 23};
    u64 zeta = 1ULL << 40;
    lfo_NONRES = zeta;
    for (int k = 0; k < 8; k++) {
        int e = E[k];
        fq3 y = fq3_zero();
        if (e % 3 == 1) y.c[1] = fq_pow(zeta, (e - 1) / 3);
        else y.c[2] = fq_pow(zeta, (e - 2) / 3);
        g_y[k] = y;
    }
}

static void ensure_init(void) {
    if (g_init) return;
    default_ring();
    if (build_tables() != 0) abort();
    g_init = 1;
}

int lfo_set_ring(u64 nonres, const u64 *y) {
    ensure_init();
    u64 old_nr = lfo_NONRES;
    fq3 old_y[8];
    memcpy(old_y, g_y, sizeof(old_y));
    lfo_NONRES = nonres % LFO_P;
    int ok = 1;
    u64 roots[8];
    for (int k = 0; k < 8 && ok; k++) {
        fq3 v = {{y[3 * k] % LFO_P, y[3 * k + 1] % LFO_P, y[3 * k + 2] % LFO_P}};
        g_y[k] = v;
        fq3 cube = fq3_pow_small(v, 3);
        if (cube.c[1] || cube.c[2]) ok = 0;
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
        for (int c = 0; c < 3; c++) y[3 * k + c] = g_y[k].c[c];
}

void lfo_set_digit_mode(int mode) { g_digit_mode = mode; }

void lfo_fq3_mul(const u64 *a, const u64 *b, u64 *out) {
    ensure_init();
    fq3 x = {{a[0], a[1], a[2]}}, y = {{b[0], b[1], b[2]}};
    fq3 r = fq3_mul(x, y);
    out[0] = r.c[0]; out[1] = r.c[1]; out[2] = r.c[2];
}

/* CRT: slot_k = a(y_k) evaluated in F_{p^3} */
void lfo_crt(const u64 *in, u64 *out, size_t count) {
    ensure_init();
#pragma omp parallel for schedule(static) if (count >= 4096)
    for (size_t e = 0; e < count; e++) {
        const u64 *a = in + 24 * e;
        u64 res[24];
        for (int k = 0; k < 8; k++) {
            fq3 acc = fq3_zero();
            for (int i = 0; i < 24; i++)
                if (a[i]) acc = fq3_add(acc, fq3_mul_fq(g_ypow[k][i], a[i]));
            rq_set_slot(res, k, acc);
        }
        memcpy(out + 24 * e, res, sizeof(res));
    }
}

void lfo_icrt(const u64 *in, u64 *out, size_t count) {
    ensure_init();
#pragma omp parallel for schedule(static) if (count >= 4096)
    for (size_t e = 0; e < count; e++) {
        const u64 *x = in + 24 * e;
        u64 res[24];
        for (int i = 0; i < 24; i++) {
            u64 acc = 0;
            for (int j = 0; j < 24; j++)
                if (x[j]) acc = fq_add(acc, fq_mul(g_icrt[i][j], x[j]));
            res[i] = acc;
        }
        memcpy(out + 24 * e, res, sizeof(res));
    }
}

void lfo_ring_mul_ntt(const u64 *a, const u64 *b, u64 *out, size_t count) {
    ensure_init();
    for (size_t e = 0; e < count; e++) rq_mul(out + 24 * e, a + 24 * e, b + 24 * e);
}

/* multiply by X in Z_p[X]/(X^24 - X^12 + 1) */
static void rot_x(u64 *a) {
    u64 top = a[23];
    for (int i = 23; i > 0; i--) a[i] = a[i - 1];
    a[0] = fq_neg(top);
    a[12] = fq_add(a[12], top);
}

void lfo_ring_mul_coeff(const u64 *a, const u64 *b, u64 *out) {
    u64 rot[24], acc[24] = {0};
    memcpy(rot, a, sizeof(rot));
    for (int i = 0; i < 24; i++) {
        for (int j = 0; j < 24; j++) acc[j] = fq_add(acc[j], fq_mul(rot[j], b[i]));
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
        if (g_digit_mode == 0) {
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
        for (int c = 0; c < 24; c++) {
            decompose_coeff(in[24 * e + c], base, digits, dg);
            for (u32 k = 0; k < digits; k++) {
                size_t idx = layout == 0 ? e * digits + k : (size_t)k * count + e;
                out[24 * idx + c] = fq_from_i64(dg[k]);
            }
        }
    }
}

void lfo_recompose(const u64 *in, size_t count_out, u64 base, u32 digits, u64 *out) {
#pragma omp parallel for schedule(static) if (count_out >= 4096)
    for (size_t e = 0; e < count_out; e++) {
        u64 acc[24] = {0};
        u64 pw = 1;
        for (u32 j = 0; j < digits; j++) {
            const u64 *x = in + 24 * (e * digits + j);
            for (int c = 0; c < 24; c++) acc[c] = fq_add(acc[c], fq_mul(x[c], pw));
            pw = fq_mul(pw, base % LFO_P);
        }
        memcpy(out + 24 * e, acc, sizeof(acc));
    }
}

/* rot_sum / rot_lin_combination, cyclotomic-rings/src/rotation.rs:45-104.
 * theta: n x tau_elems NTT-form ring elements; flatten_to_coeffs = concatenation of slots
 * (KAT-verified); result: tau_elems NTT-form elements. */
void lfo_rot_lin_combination(const u64 *rho_coeff, const u64 *theta, u32 n, u32 tau_elems, u64 *out) {
    ensure_init();
    u32 flat = tau_elems * 8; /* number of F_{p^3} entries; must equal ring degree 24 */
    fq3 res[24];
    for (int j = 0; j < 24; j++) res[j] = fq3_zero();
    for (u32 i = 0; i < n; i++) {
        u64 rot[24];
        memcpy(rot, rho_coeff + 24 * i, sizeof(rot));
        const u64 *th = theta + (size_t)24 * tau_elems * i;
        for (u32 bi = 0; bi < flat && bi < 24; bi++) {
            fq3 b = {{th[3 * bi], th[3 * bi + 1], th[3 * bi + 2]}};
            for (int j = 0; j < 24; j++) res[j] = fq3_add(res[j], fq3_mul_fq(b, rot[j]));
            rot_x(rot);
        }
    }
    for (int j = 0; j < 24; j++) { out[3 * j] = res[j].c[0]; out[3 * j + 1] = res[j].c[1]; out[3 * j + 2] = res[j].c[2]; }
}

/* GoldilocksChallengeSet::short_challenge_from_random_bytes, rings/goldilocks.rs:36-68:
 * 24 six-bit fields, LSB-first within each 3-byte group, each minus 32. */
int lfo_short_challenge_from_bytes(const uint8_t *bs, size_t n, u64 *coeff_out) {
    if (n != 18) return -1;
    for (int g = 0; g < 6; g++) {
        u32 w = (u32)bs[3 * g] | ((u32)bs[3 * g + 1] << 8) | ((u32)bs[3 * g + 2] << 16);
        for (int j = 0; j < 4; j++) coeff_out[4 * g + j] = fq_from_i64((int64_t)((w >> (6 * j)) & 63) - 32);
    }
    return 0;
}

// lf_verify.h -- host-side NIFSVerifier::verify (crates/latticefold/src/nifs.rs:117-163) for either ring, behind
// lf_verify_host (include/lfhip.h).  O(proof size) work, no GPU.  SURVEY 8(f) rank 3: "the step after the path".
//
// Restates: LFLinearizationVerifier::verify (nifs/linearization.rs:264-290, :193-243), LFDecompositionVerifier::verify
// (nifs/decomposition.rs:94-157), LFFoldingVerifier::verify (nifs/folding.rs:136-195, :273-343; expected value
// folding/utils.rs:366-413; outputs folding/utils.rs:460-521), MLSumcheck::verify_as_subprotocol +
// check_and_generate_subclaim + interpolate_uni_poly (utils/sumcheck.rs:82-110, utils/sumcheck/verifier.rs:92-257).
//
// `V` is a small policy class giving the ring's host arithmetic (GoldV in lf_capi.cpp, BbV in bb_capi.cpp).
#pragma once
#include <string.h>

#include <vector>

#include "../../include/lfhip.h"

namespace lfv {

typedef uint64_t u64;
typedef uint32_t u32;

template <class V>
struct Verifier {
    typedef typename V::Ext Ext;
    typedef typename V::Tr Tr;
    static constexpr int RE = V::RE, TAU = V::TAU;
    const V &v;
    const lf_params &P;
    const u32 *S_off, *S_idx;
    const u64 *cc;   // q ring elements
    int stage = 0;   // which check rejected

    Verifier(const V &vv, const lf_params &p, const u32 *so, const u32 *si, const u64 *c) : v(vv), P(p), S_off(so), S_idx(si), cc(c) {}

    size_t lcccs_len() const { return (size_t)P.s + TAU + P.kappa + P.t + P.l + 1; }
    size_t cccs_len() const { return (size_t)P.kappa + P.l; }
    size_t lin_len() const { return (size_t)P.s * (P.d + 2) + TAU + P.t; }
    size_t dec_len() const { return (size_t)P.K * (P.t + TAU + P.l + 1 + P.kappa); }

    void challenges(Tr &tr, u32 n, std::vector<Ext> &out) const {
        out.resize(n);
        for (u32 i = 0; i < n; i++) out[i] = tr.get_challenge();
    }
    // eq_eval (utils/sumcheck/utils.rs:78-92) on diagonal points
    Ext eq_eval(const Ext *x, const Ext *y, u32 n) const {
        Ext res = v.ext_from_u64(1), one = res;
        for (u32 i = 0; i < n; i++) {
            Ext xy = v.ext_mul(x[i], y[i]);
            Ext u = v.ext_add(v.ext_sub(v.ext_sub(v.ext_add(xy, xy), x[i]), y[i]), one);
            res = v.ext_mul(res, u);
        }
        return res;
    }
    // Lagrange interpolation through (0, p_0), .., (len-1, p_{len-1}) evaluated at `at` (ring elements, slot-constant weights)
    void interpolate(const u64 *p_i, u32 len, Ext at, u64 *out) const {
        u64 res[RE] = {0}, t[RE];
        for (u32 i = 0; i < len; i++) {
            Ext num = v.ext_from_u64(1), den = num;
            for (u32 j = 0; j < len; j++) {
                if (j == i) continue;
                num = v.ext_mul(num, v.ext_sub(at, v.ext_from_u64(j)));
                den = v.ext_mul(den, v.ext_sub(v.ext_from_u64(i), v.ext_from_u64(j)));
            }
            Ext w = v.ext_mul(num, v.ext_inv(den));
            v.mul_ext(p_i + (size_t)i * RE, w, t);
            v.add(res, t, res);
        }
        memcpy(out, res, sizeof(res));
    }
    // verify_as_subprotocol: returns false on a failed round check
    bool sumcheck(Tr &tr, u32 nv, u32 degree, const u64 *claimed, const u64 *msgs, std::vector<Ext> &point, u64 *expected_out) const {
        tr.absorb_u64_as_ring(nv);
        tr.absorb_u64_as_ring(degree);
        point.resize(nv);
        for (u32 i = 0; i < nv; i++) {
            tr.absorb_ring(msgs + (size_t)i * (degree + 1) * RE, degree + 1);
            point[i] = tr.get_challenge();
            v.absorb_ext(tr, point[i]);
        }
        u64 expected[RE], s[RE];
        memcpy(expected, claimed, sizeof(expected));
        for (u32 i = 0; i < nv; i++) {
            const u64 *ev = msgs + (size_t)i * (degree + 1) * RE;
            v.add(ev, ev + RE, s);
            if (memcmp(s, expected, sizeof(s))) return false;
            interpolate(ev, degree + 1, point[i], expected);
        }
        memcpy(expected_out, expected, sizeof(expected));
        return true;
    }
    void write_point(const std::vector<Ext> &pt, u64 *o) const {
        for (size_t i = 0; i < pt.size(); i++) v.from_ext(pt[i], o + i * RE);
    }

    bool verify_linearization(Tr &tr, const u64 *cccs, const u64 *proof, u64 *lcccs_out) {
        tr.absorb_label("beta_s");
        std::vector<Ext> beta, pt;
        challenges(tr, P.s, beta);
        u64 zero[RE] = {0}, s[RE];
        if (!sumcheck(tr, P.s, P.d + 1, zero, proof, pt, s)) { stage = 1; return false; }
        const u64 *vv = proof + (size_t)P.s * (P.d + 2) * RE, *u = vv + (size_t)TAU * RE;
        Ext e = eq_eval(pt.data(), beta.data(), P.s);
        u64 sum[RE] = {0}, term[RE];
        for (u32 i = 0; i < P.q; i++) {
            memcpy(term, cc + (size_t)i * RE, sizeof(term));
            for (u32 k = S_off[i]; k < S_off[i + 1]; k++) v.mul(term, u + (size_t)S_idx[k] * RE, term);
            v.add(sum, term, sum);
        }
        v.mul_ext(sum, e, sum);
        if (memcmp(sum, s, sizeof(s))) { stage = 2; return false; }
        tr.absorb_ring(vv, TAU);
        tr.absorb_ring(u, P.t);
        u64 *o = lcccs_out;
        write_point(pt, o); o += (size_t)P.s * RE;
        memcpy(o, vv, (size_t)TAU * RE * 8); o += (size_t)TAU * RE;
        memcpy(o, cccs, (size_t)P.kappa * RE * 8); o += (size_t)P.kappa * RE;
        memcpy(o, u, (size_t)P.t * RE * 8); o += (size_t)P.t * RE;
        memcpy(o, cccs + (size_t)P.kappa * RE, (size_t)P.l * RE * 8); o += (size_t)P.l * RE;
        v.from_u64(1, o);
        return true;
    }
    bool verify_decomposition(Tr &tr, const u64 *lcccs, const u64 *proof, u64 *out_K, int stage_id) {
        u32 K = P.K;
        size_t ll = lcccs_len();
        const u64 *u_s = proof, *v_s = u_s + (size_t)K * P.t * RE, *x_s = v_s + (size_t)K * TAU * RE, *y_s = x_s + (size_t)K * (P.l + 1) * RE;
        for (u32 k = 0; k < K; k++) {
            const u64 *xk = x_s + (size_t)k * (P.l + 1) * RE, *yk = y_s + (size_t)k * P.kappa * RE;
            const u64 *uk = u_s + (size_t)k * P.t * RE, *vk = v_s + (size_t)k * TAU * RE;
            tr.absorb_ring(xk, P.l + 1);
            tr.absorb_ring(yk, P.kappa);
            tr.absorb_ring(uk, P.t);
            tr.absorb_ring(vk, TAU);
            u64 *o = out_K + (size_t)k * ll * RE;
            memcpy(o, lcccs, (size_t)P.s * RE * 8); o += (size_t)P.s * RE;
            memcpy(o, vk, (size_t)TAU * RE * 8); o += (size_t)TAU * RE;
            memcpy(o, yk, (size_t)P.kappa * RE * 8); o += (size_t)P.kappa * RE;
            memcpy(o, uk, (size_t)P.t * RE * 8); o += (size_t)P.t * RE;
            memcpy(o, xk, (size_t)(P.l + 1) * RE * 8);
        }
        // recomposition checks with b^k (calculate_b_s, recompose_commitment, recompose)
        struct { const u64 *parts; u32 cnt; const u64 *want; } chk[4] = {
            {y_s, P.kappa, lcccs + ((size_t)P.s + TAU) * RE},
            {v_s, (u32)TAU, lcccs + (size_t)P.s * RE},
            {u_s, P.t, lcccs + ((size_t)P.s + TAU + P.kappa) * RE},
            {x_s, P.l + 1, lcccs + ((size_t)P.s + TAU + P.kappa + P.t) * RE},
        };
        for (int c = 0; c < 4; c++)
            for (u32 j = 0; j < chk[c].cnt; j++) {
                u64 a[RE] = {0}, t[RE], bk[RE];
                u64 pw = 1;
                for (u32 k = 0; k < K; k++) {
                    v.from_u64(pw, bk);
                    v.mul(chk[c].parts + ((size_t)k * chk[c].cnt + j) * RE, bk, t);
                    v.add(a, t, a);
                    pw = v.fmul(pw, P.b);
                }
                if (memcmp(a, chk[c].want + (size_t)j * RE, sizeof(a))) { stage = stage_id; return false; }
            }
        return true;
    }

    // returns LF_OK / LF_ERR_REJECT; lcccs_out = folded instance
    // Input validation (untrusted proof bytes): parameter envelope, multiset structure, canonical residues everywhere, diagonal
    // evaluation point.  The arithmetic and the equality checks below assume canonical words; arkworks' deserializer rejects
    // non-canonical field elements the same way (Validate::Yes).
    int validate(const u64 *acc, const u64 *cm_i, const u64 *proof) const {
        if (P.s == 0 || P.s > 40 || P.K == 0 || P.K > 32 || P.q == 0 || P.q > 8 || P.t == 0 || P.t > 16 || P.d == 0 || P.d > 8 ||
            P.b < 2 || P.b > 64 || P.kappa == 0 || P.kappa > 4096 || P.l > 4096 || P.L == 0 || P.L > 64)
            return LF_ERR_UNSUPPORTED;
        if (S_off[0] != 0) return LF_ERR_INVALID;
        for (u32 i = 0; i < P.q; i++)
            if (S_off[i + 1] < S_off[i] || S_off[i + 1] - S_off[i] > P.d) return LF_ERR_INVALID;
        if (S_off[P.q] > 64) return LF_ERR_INVALID;
        for (u32 k = 0; k < S_off[P.q]; k++)
            if (S_idx[k] >= P.t) return LF_ERR_INVALID;
        const u64 p = V::modulus();
        auto canon = [p](const u64 *w, size_t n) {
            for (size_t i = 0; i < n; i++)
                if (w[i] >= p) return false;
            return true;
        };
        const size_t proof_len = lin_len() + 2 * dec_len() + (size_t)P.s * (2 * P.b + 1) + 2 * (size_t)P.K * (TAU + P.t);
        if (!canon(cc, (size_t)P.q * RE) || !canon(acc, lcccs_len() * RE) || !canon(cm_i, cccs_len() * RE) || !canon(proof, proof_len * RE))
            return LF_ERR_INVALID;
        for (u32 i = 0; i < P.s; i++)   // acc.r: diagonal embeddings of F_{p^tau} challenges (eq_eval below reads slot 0 only)
            for (int sl = 1; sl < 8; sl++)
                if (memcmp(acc + (size_t)i * RE, acc + (size_t)i * RE + (size_t)TAU * sl, TAU * sizeof(u64)) != 0) return LF_ERR_UNSUPPORTED;
        return LF_OK;
    }

    int verify(Tr &tr, const u64 *acc, const u64 *cm_i, const u64 *proof, u64 *lcccs_out) {
        int vrc = validate(acc, cm_i, proof);
        if (vrc != LF_OK) return vrc;
        u32 K = P.K, K2 = 2 * K;
        size_t ll = lcccs_len();
        tr.absorb_label("acc");   // absorb_public_input, nifs.rs:175-197
        tr.absorb_ring(acc, ll);
        tr.absorb_label("cm_i");
        tr.absorb_ring(cm_i, cccs_len());
        const u64 *lin_proof = proof, *decl = lin_proof + lin_len() * RE, *decr = decl + dec_len() * RE, *foldp = decr + dec_len() * RE;
        std::vector<u64> lin(ll * RE), parts((size_t)K2 * ll * RE);
        if (!verify_linearization(tr, cm_i, lin_proof, lin.data())) return LF_ERR_REJECT;
        if (!verify_decomposition(tr, acc, decl, parts.data(), 3)) return LF_ERR_REJECT;
        if (!verify_decomposition(tr, lin.data(), decr, parts.data() + (size_t)K * ll * RE, 4)) return LF_ERR_REJECT;

        // LFFoldingVerifier::verify
        std::vector<Ext> alpha, zeta, mu, beta, r0;
        tr.absorb_label("alpha_s"); challenges(tr, K2, alpha);
        tr.absorb_label("zeta_s");  challenges(tr, K2, zeta);
        tr.absorb_label("mu_s");    challenges(tr, K2 - 1, mu);
        mu.push_back(v.ext_from_u64(1));
        tr.absorb_label("beta_s");  challenges(tr, P.s, beta);
        // calculate_claims: sum_i sum_j alpha_i^{j+1} v_ij + zeta_i^{j+1} u_ij
        u64 claim[RE] = {0}, t[RE];
        for (u32 i = 0; i < K2; i++) {
            const u64 *li = parts.data() + (size_t)i * ll * RE;
            Ext pw = alpha[i];
            for (int d = 0; d < TAU; d++) {
                v.mul_ext(li + ((size_t)P.s + d) * RE, pw, t); v.add(claim, t, claim);
                pw = v.ext_mul(pw, alpha[i]);
            }
            pw = zeta[i];
            for (u32 j = 0; j < P.t; j++) {
                v.mul_ext(li + ((size_t)P.s + TAU + P.kappa + j) * RE, pw, t); v.add(claim, t, claim);
                pw = v.ext_mul(pw, zeta[i]);
            }
        }
        u64 expected[RE];
        u32 deg = 2 * P.b;
        if (!sumcheck(tr, P.s, deg, claim, foldp, r0, expected)) { stage = 5; return LF_ERR_REJECT; }
        const u64 *theta = foldp + (size_t)P.s * (deg + 1) * RE, *eta = theta + (size_t)K2 * TAU * RE;
        {   // verify_evaluation / compute_sumcheck_claim_expected_value
            Ext e_ast = eq_eval(beta.data(), r0.data(), P.s);
            u64 total[RE] = {0}, s1[RE], s2[RE], s3[RE], th2[RE], prod[RE], bb[RE], m2[RE];
            std::vector<Ext> ri(P.s);
            for (u32 i = 0; i < K2; i++) {
                const u64 *li = parts.data() + (size_t)i * ll * RE;
                for (u32 q = 0; q < P.s; q++) ri[q] = v.ext_of(li + (size_t)q * RE);
                Ext e_i = eq_eval(ri.data(), r0.data(), P.s);
                memset(s1, 0, sizeof(s1)); memset(s2, 0, sizeof(s2)); memset(s3, 0, sizeof(s3));
                Ext pw = alpha[i];
                for (int d = 0; d < TAU; d++) {
                    v.mul_ext(theta + ((size_t)i * TAU + d) * RE, v.ext_mul(pw, e_i), t); v.add(s1, t, s1);
                    pw = v.ext_mul(pw, alpha[i]);
                }
                pw = mu[i];
                for (int d = 0; d < TAU; d++) {
                    const u64 *th = theta + ((size_t)i * TAU + d) * RE;
                    v.mul(th, th, th2);
                    v.from_u64(1, prod);
                    for (u32 b = 1; b < P.b; b++) { v.from_u64((u64)b * b, bb); v.sub(th2, bb, m2); v.mul(prod, m2, prod); }
                    v.mul(th, prod, t); v.mul_ext(t, pw, t); v.add(s2, t, s2);
                    pw = v.ext_mul(pw, mu[i]);
                }
                v.mul_ext(s2, e_ast, s2);
                pw = zeta[i];
                for (u32 j = 0; j < P.t; j++) {
                    v.mul_ext(eta + ((size_t)i * P.t + j) * RE, pw, t); v.add(s3, t, s3);
                    pw = v.ext_mul(pw, zeta[i]);
                }
                v.mul_ext(s3, e_i, s3);
                v.add(total, s1, total); v.add(total, s2, total); v.add(total, s3, total);
            }
            if (memcmp(total, expected, sizeof(total))) { stage = 6; return LF_ERR_REJECT; }
        }
        tr.absorb_ring(theta, (size_t)K2 * TAU);
        tr.absorb_ring(eta, (size_t)K2 * P.t);
        tr.absorb_label("rho_s");   // get_rhos (folding/utils.rs:116-131)
        std::vector<u64> rho_c((size_t)K2 * RE, 0), rho((size_t)K2 * RE);
        for (u32 i = 0; i + 1 < K2; i++) tr.get_short_challenge(&rho_c[(size_t)i * RE]);
        rho_c[(size_t)(K2 - 1) * RE] = 1;
        for (u32 i = 0; i < K2; i++) v.crt(&rho_c[(size_t)i * RE], &rho[(size_t)i * RE]);
        // compute_v0_u0_x0_cm_0
        u64 *o = lcccs_out;
        write_point(r0, o); o += (size_t)P.s * RE;
        {   // v_0 = rot_lin_combination(rho_coeff, theta) (cyclotomic-rings/src/rotation.rs:85-104)
            std::vector<u64> res((size_t)RE * TAU, 0);
            for (u32 i = 0; i < K2; i++) {
                u64 rot[RE];
                memcpy(rot, &rho_c[(size_t)i * RE], sizeof(rot));
                const u64 *th = theta + (size_t)i * TAU * RE;
                for (int bi = 0; bi < RE; bi++) {
                    const u64 *b = th + (size_t)TAU * bi;
                    for (int j = 0; j < RE; j++)
                        if (rot[j])
                            for (int q = 0; q < TAU; q++) res[(size_t)j * TAU + q] = v.fadd(res[(size_t)j * TAU + q], v.fmul(b[q], rot[j]));
                    v.rot_x(rot);
                }
            }
            memcpy(o, res.data(), res.size() * 8);
            o += (size_t)TAU * RE;
        }
        auto part = [&](u32 i) { return parts.data() + (size_t)i * ll * RE; };
        for (u32 c = 0; c < P.kappa; c++, o += RE) {
            memset(o, 0, RE * 8);
            for (u32 i = 0; i < K2; i++) { v.mul(part(i) + ((size_t)P.s + TAU + c) * RE, &rho[(size_t)i * RE], t); v.add(o, t, o); }
        }
        for (u32 j = 0; j < P.t; j++, o += RE) {
            memset(o, 0, RE * 8);
            for (u32 i = 0; i < K2; i++) { v.mul(&rho[(size_t)i * RE], eta + ((size_t)i * P.t + j) * RE, t); v.add(o, t, o); }
        }
        for (u32 c = 0; c < P.l + 1; c++, o += RE) {
            memset(o, 0, RE * 8);
            for (u32 i = 0; i < K2; i++) { v.mul(&rho[(size_t)i * RE], part(i) + ((size_t)P.s + TAU + P.kappa + P.t + c) * RE, t); v.add(o, t, o); }
        }
        return LF_OK;
    }
};

}  // namespace lfv

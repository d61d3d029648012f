// bb_prove.cpp -- BabyBearRingNTT backend: the host driver that replays `NIFSProver::prove` (crates/latticefold/src/nifs.rs:48-103) on the kernels of
// bb_kernels.hip -- linearization, decomposition and folding provers and the two sumchecks as stand-alone entry points.  Same structure as the Goldilocks
// driver (lf_prove.cpp / lf_fold.cpp: f-hat virtual, Mz restructured, f_0 in the coefficient domain), one stream.  Host <-> device traffic inside a fold
// step is O(proof size).
#include "bb_ctx.h"

namespace lfbb {

// =================================================================================================================================
// the driver
static void sc_prologue(BbTranscript &tr, u32 nv, u32 deg) {   // utils/sumcheck.rs:60-62
    tr.absorb_u64_as_ring(nv);
    tr.absorb_u64_as_ring(deg);
}
static H9 sc_round_transcript(BbTranscript &tr, const u64 *evals, u32 npts) {
    tr.absorb_ring(evals, npts);
    H9 r = tr.get_challenge();
    tr.absorb_h9_as_ring(r);
    return r;
}
static size_t atl(size_t x) { return x < 2 ? 2 : x; }   // leading dimensions stay even (8-byte pair loads)

// linearization sumcheck on device tables mz [t][72][m] (left intact) and eq_beta [9][m]
// `u_dev` (optional): u_j = Mz_j(r), t ring elements (canonical), from the last fix of the tables the rounds work on
static int run_lin_sumcheck(C *c, BbTranscript &tr, const fe *mz, const fe *eqb, u64 *msgs, H9 *point, u64 *u_dev = nullptr) {
    const lf_params &P = c->P;
    u32 deg = P.d + 1;
    size_t m = c->m;
    fe *fx[2], *fq[2];
    i64 *partial;
    u64 *od;
    RET(c->tbuf("lin_fix0", (size_t)P.t * RE * atl(m / 2), &fx[0]));
    RET(c->tbuf("lin_fix1", (size_t)P.t * RE * atl(m / 4), &fx[1]));
    RET(c->tbuf("lin_efix0", TAU * atl(m / 2), &fq[0]));
    RET(c->tbuf("lin_efix1", TAU * atl(m / 4), &fq[1]));
    RET(c->tbuf("round_partial", red_partial_words(5 * RE), &partial));
    od = c->round_out();
    if (!od) return LF_ERR_HIP;
    { HostTimer ht(c); sc_prologue(tr, P.s, deg); }
    const fe *cur = mz, *cure = eqb;
    size_t n = m;
    int flip = 0;
    // the R1CS shape has its own kernel with fix_variables fused in (bb_kernels.hip: k_lin_r1cs)
    const bool r1cs = lin_desc_is_r1cs(c->desc) && deg == 3 && !c->tn.lin_no_r1cs;
    for (u32 round = 1; round <= P.s; round++) {
        bool small = false;
        if (r1cs && (round == 1 || n >= 4)) {
            if (round == 1) launch_lin_r1cs(c->dev, cur, m, cure, m, n / 2, nullptr, nullptr, 0, nullptr, 0, partial, od, c->stream(), c->lin_blocks);
            else {
                const E9PreC r = e9pre_from_h9(point[round - 2], c->ring.T.nu);
                const size_t ldp = round == 2 ? m : atl(n);
                launch_lin_r1cs(c->dev, cur, ldp, cure, ldp, n / 4, &r, fx[flip], atl(n / 2), fq[flip], atl(n / 2), partial, od, c->stream(), c->lin_blocks);
                cur = fx[flip]; cure = fq[flip];
                flip ^= 1;
                n /= 2;
            }
            small = true;   // (the message is on its way)
        } else
        if (round > 1) {
            E9PreC r = e9pre_from_h9(point[round - 2], c->ring.T.nu);
            const size_t ldp = round == 2 ? m : atl(n);
            // small rounds (at most 256 pairs): fix + evaluation + reduction in one launch (they are launch-bound: four launches otherwise)
            small = n >= 4 && n / 4 <= 256 && !c->tn.lin_no_small;
            if (small) launch_lin_small(c->dev, c->desc, cur, ldp, cure, ldp, n, r, fx[flip], atl(n / 2), fq[flip], atl(n / 2), deg, od, c->stream());
            else {
                launch_fix(c->dev, cur, n, fx[flip], atl(n / 2), n, P.t * 8, r, c->stream());
                launch_fix(c->dev, cure, n, fq[flip], atl(n / 2), n, 1, r, c->stream());
            }
            cur = fx[flip]; cure = fq[flip];
            flip ^= 1;
            n /= 2;
        }
        size_t ld = round == 1 ? m : atl(n);
        if (!small) launch_lin_round(c->dev, c->desc, cur, ld, cure, ld, n, deg, partial, od, c->stream(), c->lin_blocks);
        u64 *ev = msgs + (size_t)(round - 1) * (deg + 1) * RE;
        HIPCHK(hipStreamSynchronize(c->stream()));            // the reduce kernel wrote the message into mapped host memory
        memcpy(ev, od, (size_t)(deg + 1) * RE * 8);
        HostTimer ht(c);
        point[round - 1] = sc_round_transcript(tr, ev, deg + 1);
        if (round <= 4 || round == 8) { char nm[32]; snprintf(nm, sizeof nm, "  lin round %u", round); BB_MARK(nm); }
    }
    if (u_dev) launch_fix_final(c->dev, cur, 2, P.t * 8, e9pre_from_h9(point[P.s - 1], c->ring.T.nu), u_dev, c->stream());   // two entries per row left
    return LF_OK;
}

// z tables: head (x.., h) || w, w from the planes
static int build_z(C *c, const int32_t *planes, u32 K, int mode_bits, const u64 *heads /* K*(l+1) ring AoS host */, fe *z /* [K][72][n] */) {
    const lf_params &P = c->P;
    u32 hl = P.l + 1;
    launch_recompose_crt(c->dev, planes, c->N, P.wit_len, P.L, P.B, K, mode_bits, z, c->n, hl, c->stream());
    std::vector<fe> h((size_t)K * RE * hl);
    for (u32 k = 0; k < K; k++)
        for (u32 i = 0; i < hl; i++)
            for (int w = 0; w < RE; w++) h[((size_t)k * RE + w) * hl + i] = from_canon(heads[((size_t)k * hl + i) * RE + w]);
    fe *stage;
    RET(c->tbuf("z_heads", h.size(), &stage));
    HIPCHK(hipMemcpyAsync(stage, h.data(), h.size() * sizeof(fe), hipMemcpyHostToDevice, c->stream()));
    HIPCHK(hipMemcpy2DAsync(z, c->n * sizeof(fe), stage, hl * sizeof(fe), hl * sizeof(fe), (size_t)K * RE, hipMemcpyDeviceToDevice, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    return LF_OK;
}
static int build_z_async(C *c, const int32_t *planes, u32 K, int mode_bits, const u64 *heads, fe *z) {
    const lf_params &P = c->P;
    u32 hl = P.l + 1;
    launch_recompose_crt(c->dev, planes, c->N, P.wit_len, P.L, P.B, K, mode_bits, z, c->n, hl, c->stream());
    size_t cnt = (size_t)K * RE * hl;
    fe *h = (fe *)c->arena_alloc((cnt * sizeof(fe) + 7) / 8);
    if (!h) return LF_ERR_HIP;
    for (u32 k = 0; k < K; k++)
        for (u32 i = 0; i < hl; i++)
            for (int w = 0; w < RE; w++) h[((size_t)k * RE + w) * hl + i] = from_canon(heads[((size_t)k * hl + i) * RE + w]);
    fe *stage;
    RET(c->tbuf("z_heads_async", cnt, &stage));
    HIPCHK(hipMemcpyAsync(stage, h, cnt * sizeof(fe), hipMemcpyHostToDevice, c->stream()));
    HIPCHK(hipMemcpy2DAsync(z, c->n * sizeof(fe), stage, hl * sizeof(fe), hl * sizeof(fe), (size_t)K * RE, hipMemcpyDeviceToDevice, c->stream()));
    return LF_OK;
}
static bool lcccs_point(const lf_params &P, const u64 *lcccs, std::vector<H9> &pt) {
    pt.resize(P.s);
    for (u32 i = 0; i < P.s; i++)
        if (!is_diag(lcccs + (size_t)i * RE, &pt[i])) return false;
    return true;
}

// LFLinearizationProver::prove (nifs/linearization.rs:145-189)

// T[k][c] = sum_i eq[i] digit_k(planes[c][i]) of the K binary digit planes -> out (device, canonical): on the int8 matrix cores (bb_dot_i8.hip) unless
// LF_COEF_VALU is set or the shape is not handled there
static int coef_eval_bits_dev(BbCtxImpl *c, const int32_t *planes, size_t n, const fe *eq, size_t ldeq, u32 K, i64 *partial, u64 *out) {
    if (K <= 16 && n >= 64) {
        unsigned char *EB;
        int32_t *part;
        long long *tot;
        const u32 nwg = 256;
        RET(c->tbuf("ce_eb", coef_eval_i8_eb_bytes(n), &EB));
        RET(c->tbuf("ce_part", coef_eval_i8_part_words(nwg), &part));
        RET(c->tbuf("ce_tot", coef_eval_i8_tot_words(), &tot));
        if (launch_coef_eval_i8(planes, n, n, eq, ldeq, K, EB, nwg, part, tot, out, c->stream()) == 0) return LF_OK;
    }
    launch_coef_eval(c->dev, planes, n, eq, ldeq, K, 1, partial, out, c->stream());
    return LF_OK;
}
static int linearize_impl(C *c, BbTranscript &tr, const u64 *cccs, const lf_witness *wit, u64 *lcccs_out, u64 *proof, fe **eq_r_keep) {
    const lf_params &P = c->P;
    size_t m = c->m, n = c->n;
    size_t ph = c->ev_begin(10);
    std::vector<u64> head((size_t)(P.l + 1) * RE);   // z = x_ccs || 1 || w_ccs (arith.rs:399-409)
    memcpy(head.data(), cccs + (size_t)P.kappa * RE, (size_t)P.l * RE * 8);
    BbHostRing::from_u64(1, head.data() + (size_t)P.l * RE);
    fe *z, *mz, *eqb, *eqr;
    i64 *partial;
    u64 *od;
    RET(c->tbuf("lin_z", RE * n, &z));
    RET(c->tbuf("lin_mz", (size_t)P.t * RE * m, &mz));
    RET(c->tbuf("lin_eqb", TAU * m, &eqb));
    RET(c->tbuf("eq_r_R", TAU * m, &eqr));
    RET(c->tbuf("red_partial", red_partial_words(16 * RE * TAU), &partial));
    RET(c->tbuf("lin_small", 16 * RE * TAU, &od));
    BB_MARK(" lin: buffers");
    RET(build_z(c, wit->planes, 1, 0, head.data(), z));
    BB_MARK(" lin: z built (synced)");
    std::vector<H9> beta(P.s);
    {
        HostTimer ht(c);
        tr.absorb_label("beta_s");
        for (u32 i = 0; i < P.s; i++) beta[i] = tr.get_challenge();
    }
    RET(build_eq_dev(c, beta.data(), P.s, eqb));
    for (u32 j = 0; j < P.t; j++) launch_spmv(c->dev, c->d_rowptr[j], c->d_col[j], c->d_val[j], z, n, mz + (size_t)j * RE * m, m, 0, c->stream());
    std::vector<H9> pt(P.s);
    // v, u at the sumcheck point (linearization.rs:126-139): u from the fully fixed Mz tables of the sumcheck (LF_LIN_U_EVAL=1: dot
    // products with eq(r) over the full tables), v from the witness planes
    const bool u_eval = c->tn.lin_u_eval;
    BB_MARK(" lin: Mz enqueued");
    RET(run_lin_sumcheck(c, tr, mz, eqb, proof, pt.data(), u_eval ? nullptr : od + (size_t)TAU * RE));
    BB_MARK(" lin: rounds done");
    RET(build_eq_dev(c, pt.data(), P.s, eqr));
    u64 *v = proof + (size_t)P.s * (P.d + 2) * RE, *u = v + (size_t)TAU * RE;   // contiguous
    c->vs_wit = nullptr;
    if (P.b == 2 && P.K <= 16) {
        // the K digit-plane evaluations v_s[k] (the decomposition of this instance needs them at the same point anyway): v = sum_k 2^k v_s[k]
        u64 *vs;
        RET(c->tbuf("lin_vs", (size_t)P.K * TAU * RE + 8, &vs));
        RET(coef_eval_bits_dev(c, wit->planes, c->N, eqr, m, P.K, partial, vs));
        launch_vs_combine(vs, P.K, TAU * RE, od, c->stream());
        if (c->vs_keep) { c->vs_wit = wit; c->vs_eq = eqr; c->vs_dev = vs; }
    } else
        launch_coef_eval(c->dev, wit->planes, c->N, eqr, m, 1, 0, partial, od, c->stream());   // T[72][9] flat == v[9][8 slots][9]
    if (u_eval) {
        RET(down_small(c, od, (size_t)TAU * RE, v));
        launch_dot_eq(c->dev, mz, m, P.t, eqr, m, m, partial, od, c->stream());
        RET(down_small(c, od, (size_t)P.t * RE, u));
    } else RET(down_small(c, od, (size_t)TAU * RE + (size_t)P.t * RE, v));
    {
        HostTimer ht(c);
        tr.absorb_ring(v, TAU);
        tr.absorb_ring(u, P.t);
    }
    u64 *o = lcccs_out;
    for (u32 i = 0; i < P.s; i++, o += RE) BbHostRing::from_h9(pt[i], o);
    memcpy(o, v, (size_t)TAU * RE * 8); o += (size_t)TAU * RE;
    memcpy(o, cccs, (size_t)P.kappa * RE * 8); o += (size_t)P.kappa * RE;
    memcpy(o, u, (size_t)P.t * RE * 8); o += (size_t)P.t * RE;
    memcpy(o, cccs + (size_t)P.kappa * RE, (size_t)P.l * RE * 8); o += (size_t)P.l * RE;
    BbHostRing::from_u64(1, o);
    if (eq_r_keep) *eq_r_keep = eqr;
    c->ev_end(ph);
    return LF_OK;
}

// decompose_big_vec_into_k_vec_and_compose_back (nifs/decomposition/utils.rs:12-42) on l+1 elements, host
static void compute_x_s(const C *c, const u64 *xh, u64 *x_s) {
    const lf_params &P = c->P;
    u32 cnt = P.l + 1;
    std::vector<u64> co(RE);
    for (u32 i = 0; i < cnt; i++) {
        c->ring.icrt(xh + (size_t)i * RE, co.data());
        std::vector<int64_t> dB(P.L), dk(P.K);
        std::vector<std::vector<u64>> part(P.K, std::vector<u64>(RE, 0));
        for (int cc = 0; cc < RE; cc++) {
            bb_balanced_digits(co[cc], P.B, P.L, dB.data(), c->digit_mode);
            u64 pw = 1;
            for (u32 l = 0; l < P.L; l++) {
                bb_balanced_digits(hfrom_i64(dB[l]), P.b, P.K, dk.data(), c->digit_mode);
                for (u32 k = 0; k < P.K; k++) part[k][cc] = hadd(part[k][cc], hmul(pw, hfrom_i64(dk[k])));
                pw = hmul(pw, P.B % BB_P);
            }
        }
        for (u32 k = 0; k < P.K; k++) c->ring.crt(part[k].data(), x_s + ((size_t)k * cnt + i) * RE);
    }
}

struct SideState {
    const int32_t *planes;
    fe *z;      // [K][72][n]
    fe *eq_r;   // [9][m]
    std::vector<u64> lcccs;   // K flat LCCCS (host)
};

// LFDecompositionProver::prove (nifs/decomposition.rs:33-88), split so that the GPU work of one side can run while the
// host does something else: `dec_enqueue` launches everything on the current lane's stream and queues the downloads into
// the lane's pinned arena (no host synchronisation), `dec_finish` waits for it, finishes y_0 on the host and absorbs.
struct DecPending {
    u64 *h_y = nullptr, *h_v = nullptr, *h_u = nullptr;   // pinned results
    int side = 0;                                           // 0 left, 1 right: selects the milestone events ev_dec[2*side + ..]
    size_t ph_commit = 0, ph_evals = 0;
};
// The decomposition of one side is queued in two independent parts: the commitment of the K-1 upper bit-planes (a function of
// the witness only) and the evaluations at the point r (for the right side r comes out of the linearization).  Both run on the
// stream of the lane that is current when they are queued.
static int dec_enqueue_commit(C *c, const lf_witness *wit, DecPending &pd) {
    const lf_params &P = c->P;
    size_t N = c->N;
    u32 K = P.K;
    u64 *yd;
    RET(c->tbuf("dec_y", (size_t)K * P.kappa * RE, &yd));
    pd.h_y = c->arena_alloc((size_t)(K - 1) * P.kappa * RE);
    if (!pd.h_y) return LF_ERR_HIP;
    // commit_witnesses (decomposition.rs:178-201): NTT of the K-1 upper bit-planes, one batched pass over A
    pd.ph_commit = c->ev_begin(11);
    if (!c->i8_nch || P.b != 2) return LF_ERR_UNSUPPORTED;
    // int8 matrix cores: digits straight from the coefficient planes, no bit-plane NTTs (this rank's column slice when sharded)
    RET(commit_planes_i8(c, wit->planes + c->A_col0, N, 1, K - 1, yd));
    HIPCHK(hipMemcpyAsync(pd.h_y, yd, (size_t)(K - 1) * P.kappa * RE * 8, hipMemcpyDeviceToHost, c->stream()));
    c->ev_end(pd.ph_commit);
    HIPCHK(hipEventRecord(c->ev_dec[2 * pd.side], c->stream()));
    return LF_OK;
}
// <X_a, Y_b> for na vectors X and nb vectors Y of n columns -> od (device, canonical): on the int8 matrix cores (bb_dot_i8.hip) unless
// LF_DOT_VALU is set or the shape is not handled there
// st / tag: another stream and its own scratch; yb_pre: the Y digits already packed (launch_dot_pack_y) for X vectors of this alignment
static int dot_batch_dev(BbCtxImpl *c, const fe *X, size_t ldx, u32 na, const fe *Y, size_t ldy, u32 nb, size_t n, i64 *partial, u64 *od, hipStream_t st = nullptr,
                         const char *tag = "", unsigned char *yb_pre = nullptr) {
    if (!st) st = c->stream();
    if (!c->tn.dot_valu && n >= c->tn.dot_min && nb <= 3) {
        unsigned char *yb;
        int32_t *part;
        long long *tot;
        if (yb_pre) yb = yb_pre;
        else RET(c->tbuf(std::string("dot_yb") + tag, bbdot_i8_yb_bytes(n + 1), &yb));
        RET(c->tbuf(std::string("dot_i8_part") + tag, bbdot_i8_part_words(n + 1), &part));
        RET(c->tbuf(std::string("dot_i8_tot") + tag, bbdot_i8_tot_words(), &tot));
        bool ok = true;
        for (u32 a0 = 0; a0 < na && ok; a0 += 16)
            ok = launch_dot_batch_i8(c->dev, X + (size_t)a0 * RE * ldx, ldx, na - a0 < 16 ? na - a0 : 16, Y, ldy, nb, n, yb, part, tot, od + (size_t)a0 * nb * RE, st,
                                     yb_pre != nullptr) == 0;
        if (ok) return LF_OK;
    }
    launch_dot_batch(c->dev, X, ldx, na, Y, ldy, nb, n, partial, od, st);
    return LF_OK;
}

static int dec_enqueue_evals(C *c, const u64 *lcccs, const std::vector<H9> &rpt, const lf_witness *wit, const char *side, fe *eq_r, SideState &S,
                             u64 *proof, DecPending &pd) {
    const lf_params &P = c->P;
    size_t m = c->m, n = c->n, N = c->N;
    u32 K = P.K;
    std::string sd(side);
    const u64 *xh = lcccs + ((size_t)P.s + TAU + P.kappa + P.t) * RE;
    u64 *x_s = proof + (size_t)K * P.t * RE + (size_t)K * TAU * RE;
    fe *z, *q;
    i64 *partial;
    u64 *od;
    RET(c->tbuf("red_partial", red_partial_words(16 * RE * TAU), &partial));
    RET(c->tbuf("dec_small", 16 * RE * TAU + 16 * 4 * RE, &od));
    RET(c->tbuf("z_" + sd, (size_t)K * RE * n, &z));
    RET(c->tbuf("dec_q", (size_t)P.t * RE * n, &q));
    if (!eq_r) {
        RET(c->tbuf("eq_r_" + sd, TAU * m, &eq_r));
        RET(build_eq_async(c, rpt.data(), P.s, eq_r));
    }
    S.planes = wit->planes; S.z = z; S.eq_r = eq_r;
    pd.h_v = c->arena_alloc((size_t)K * TAU * RE);
    pd.h_u = c->arena_alloc((size_t)K * P.t * RE);
    if (!pd.h_v || !pd.h_u) return LF_ERR_HIP;
    pd.ph_evals = c->ev_begin(12);
    compute_x_s(c, xh, x_s);   // host, O(l) elements
    // v_s (decomposition.rs:204-211) from the coefficient planes
    if (c->vs_wit == wit && c->vs_eq == eq_r) {   // computed by the linearization of this step at this very point
        HIPCHK(hipMemcpyAsync(od, c->vs_dev, (size_t)K * TAU * RE * 8, hipMemcpyDeviceToDevice, c->stream()));
        c->vs_wit = nullptr;
    } else
        RET(coef_eval_bits_dev(c, wit->planes, N, eq_r, m, K, partial, od));
    HIPCHK(hipMemcpyAsync(pd.h_v, od, (size_t)K * TAU * RE * 8, hipMemcpyDeviceToHost, c->stream()));
    // z_k = x_s[k] || w_k ; u_s[k][j] = <z_k, M_j^T eq(r)>   (decomposition.rs:214-256 restructured)
    RET(build_z_async(c, wit->planes, K, 1, x_s, z));
    for (u32 j = 0; j < P.t; j++)
        launch_spmv_t_eq(c->dev, c->d_colptr[j], c->d_rowidx[j], c->d_valT[j], eq_r, m, q + (size_t)j * RE * n, n, c->stream());
    u64 *od2 = od + 16 * RE * TAU;
    RET(dot_batch_dev(c, z, n, K, q, n, P.t, n, partial, od2));
    HIPCHK(hipMemcpyAsync(pd.h_u, od2, (size_t)K * P.t * RE * 8, hipMemcpyDeviceToHost, c->stream()));
    c->ev_end(pd.ph_evals);
    HIPCHK(hipEventRecord(c->ev_dec[2 * pd.side + 1], c->stream()));
    return LF_OK;
}
static int dec_finish(C *c, BbTranscript &tr, const u64 *lcccs, SideState &S, u64 *proof, DecPending &pd) {
    const lf_params &P = c->P;
    u32 K = P.K;
    const u64 *cm = lcccs + ((size_t)P.s + TAU) * RE;
    u64 *u_s = proof, *v_s = u_s + (size_t)K * P.t * RE, *x_s = v_s + (size_t)K * TAU * RE, *y_s = x_s + (size_t)K * (P.l + 1) * RE;
    HIPCHK(hipEventSynchronize(c->ev_dec[2 * pd.side]));
    HIPCHK(hipEventSynchronize(c->ev_dec[2 * pd.side + 1]));
    memcpy(y_s + (size_t)P.kappa * RE, pd.h_y, (size_t)(K - 1) * P.kappa * RE * 8);
    RET(exchange_modsum(c, y_s + (size_t)P.kappa * RE, (size_t)(K - 1) * P.kappa * RE));   // partial commitments of the column shards
    memcpy(v_s, pd.h_v, (size_t)K * TAU * RE * 8);
    memcpy(u_s, pd.h_u, (size_t)K * P.t * RE * 8);
    HostTimer ht(c);
    {   // y_0 = cm - sum_{k>=1} b^k y_k, as the reference's fold (acc + y_i) * b
        // (b is a base-field constant: in the NTT form the product with it is the word-wise one -- 72 multiplications per element instead of eight F_{p^9} products)
        std::vector<u64> acc((size_t)P.kappa * RE, 0);
        const u64 bq = (u64)P.b % BB_P;
        for (int k = (int)K - 1; k >= 1; k--)
            for (u32 i = 0; i < P.kappa; i++) {
                u64 *a = &acc[(size_t)i * RE];
                const u64 *y = y_s + ((size_t)k * P.kappa + i) * RE;
                for (int w = 0; w < RE; w++) a[w] = hmul(hadd(a[w], y[w] % BB_P), bq);
            }
        for (u32 i = 0; i < P.kappa; i++) BbHostRing::sub(cm + (size_t)i * RE, &acc[(size_t)i * RE], y_s + (size_t)i * RE);
    }
    // transcript (decomposition.rs:65-83): absorb x_k, y_k, u_k, v_k and build the K LCCCS
    size_t ll = bb_lcccs_len(&P);
    S.lcccs.assign((size_t)K * ll * RE, 0);
    for (u32 k = 0; k < K; k++) {
        const u64 *xk = x_s + (size_t)k * (P.l + 1) * RE, *yk = y_s + (size_t)k * P.kappa * RE;
        const u64 *uk = u_s + (size_t)k * P.t * RE, *vk = v_s + (size_t)k * TAU * RE;
        tr.absorb_ring(xk, P.l + 1);
        tr.absorb_ring(yk, P.kappa);
        tr.absorb_ring(uk, P.t);
        tr.absorb_ring(vk, TAU);
        u64 *o = &S.lcccs[(size_t)k * ll * RE];
        memcpy(o, lcccs, (size_t)P.s * RE * 8); o += (size_t)P.s * RE;
        memcpy(o, vk, (size_t)TAU * RE * 8); o += (size_t)TAU * RE;
        memcpy(o, yk, (size_t)P.kappa * RE * 8); o += (size_t)P.kappa * RE;
        memcpy(o, uk, (size_t)P.t * RE * 8); o += (size_t)P.t * RE;
        memcpy(o, xk, (size_t)(P.l + 1) * RE * 8);
    }
    return LF_OK;
}

template <class T>
static int upload_consts(C *c, const std::string &name, const std::vector<T> &v, T **out) {
    RET(c->tbuf(name, v.size() + 8, out));
    HIPCHK(hipMemcpyAsync(*out, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    return LF_OK;
}

// LFFoldingProver::prove (nifs/folding.rs:42-130)
// C_pi(X) of lf_sv_rounds.h for the V weights W_b = eq((r_1..), b) over F_{p^9}: coefficient table [pairs][4][9] (Montgomery words).  The BabyBear twin of
// sv_build_coef in lf_capi.cpp: h = sum_x w_x(X) y_x, w_x = W_x (1 - X) (x < V), W_{x-V} X (x >= V); h^3 - h expanded over y^2 = b, y^3 = y.
static E9 e9_from_h9(const H9 &h) { E9 r; for (int i = 0; i < TAU; i++) r.c[i] = from_canon(h.c[i]); return r; }
static void bbsv_build_coef(int V, const E9 *W, fe nuM, std::vector<fe> &out) {
    const int NX = 2 * V, NPR = lf::sv_num_pairs(V);
    std::vector<E9> Cf((size_t)NPR * 4, e9_zero());
    std::vector<lf::SvPair> prs(NPR);
    for (int i = 0; i < NPR; i++) prs[i] = lf::sv_pair(V, i);
    auto find = [&](unsigned s_, unsigned b_) {
        for (int i = 0; i < NPR; i++)
            if (prs[i].s == s_ && prs[i].b == b_) return i;
        return -1;
    };
    std::vector<E9> wa(NX), wb(NX);
    for (int x = 0; x < NX; x++) {
        if (x < V) { wa[x] = W[x]; wb[x] = e9_neg(W[x]); }
        else { wa[x] = e9_zero(); wb[x] = W[x - V]; }
    }
    auto mul = [&](const E9 &a, const E9 &b) { return e9_mul(a, b, nuM); };
    for (int x = 0; x < NX; x++)
        for (int y = x; y < NX; y++) {
            const E9 p2[3] = {mul(wa[x], wa[y]), e9_add(mul(wa[x], wb[y]), mul(wb[x], wa[y])), mul(wb[x], wb[y])};
            for (int z = y; z < NX; z++) {
                E9 p3[4];
                p3[0] = mul(p2[0], wa[z]);
                p3[1] = e9_add(mul(p2[0], wb[z]), mul(p2[1], wa[z]));
                p3[2] = e9_add(mul(p2[1], wb[z]), mul(p2[2], wa[z]));
                p3[3] = mul(p2[2], wb[z]);
                int mult, idx;
                if (x == y && y == z) { mult = 1; idx = find(1u << x, 1u << x); }
                else if (x == y) { mult = 3; idx = find(1u << z, (1u << x) | (1u << z)); }      // y_x^2 y_z = b_x y_z
                else if (y == z) { mult = 3; idx = find(1u << x, (1u << x) | (1u << y)); }      // y_x y_y^2 = y_x b_y
                else { mult = 6; const unsigned mk = (1u << x) | (1u << y) | (1u << z); idx = find(mk, mk); }
                for (int e = 0; e < 4; e++) {
                    E9 acc = e9_zero();
                    for (int i = 0; i < mult; i++) acc = e9_add(acc, p3[e]);
                    Cf[(size_t)idx * 4 + e] = e9_add(Cf[(size_t)idx * 4 + e], acc);
                }
            }
        }
    for (int x = 0; x < NX; x++) {   // - h
        const int idx = find(1u << x, 1u << x);
        Cf[(size_t)idx * 4] = e9_sub(Cf[(size_t)idx * 4], wa[x]);
        Cf[(size_t)idx * 4 + 1] = e9_sub(Cf[(size_t)idx * 4 + 1], wb[x]);
    }
    out.resize((size_t)NPR * 4 * TAU);
    for (size_t i = 0; i < (size_t)NPR * 4; i++)
        for (int q = 0; q < TAU; q++) out[i * TAU + q] = Cf[i].c[q];
}

static int fold_impl(C *c, BbTranscript &tr, SideState *S, u64 *lcccs_out, lf_witness **w_out, u64 *proof) {
    const lf_params &P = c->P;
    size_t m = c->m, n = c->n, N = c->N;
    u32 K = P.K, K2 = 2 * K, deg = 2 * P.b;
    size_t ll = bb_lcccs_len(&P);
    const u64 nu = c->ring.T.nu;
    std::vector<H9> alpha(K2), zeta(K2), mu(K2), beta(P.s);
    // the bit-plane form of the two witnesses (GEMM rounds below) needs no challenge: built while the host squeezes alpha and zeta
    u32 *svbits[2] = {nullptr, nullptr};
    {
        const size_t sv_min0 = c->tn.sv_min >= 65536 ? 16384 : c->tn.sv_min;
        if (c->sh_world == 1 && !c->tn.fold_no_sv && N <= m && (N & 3) == 0 && P.s >= 4 && c->tn.sv_rounds >= 1 && m / 2 >= sv_min0 && bbsv_shape_ok(1, m / 2, K))
            for (int sd = 0; sd < 2; sd++) {
                RET(c->tbuf(sd ? "sv_bits_R" : "sv_bits_L", bbsv_bits_words(N, K), &svbits[sd]));
                launch_bbsv_bits(S[sd].planes, N, N, K, svbits[sd], c->stream());
            }
    }
    {
        HostTimer ht(c);
        tr.absorb_label("alpha_s");
        for (u32 i = 0; i < K2; i++) alpha[i] = tr.get_challenge();
        tr.absorb_label("zeta_s");
        for (u32 i = 0; i < K2; i++) zeta[i] = tr.get_challenge();
    }
    // The G tables need alpha and zeta only: their chains are enqueued here, and the host squeezes mu and beta while the GPU combines the z_k (the
    // challenge order of the transcript -- alpha, zeta, mu, beta: folding/utils.rs:52-95 -- is untouched)
    size_t ph = c->ev_begin(13);
    // powers x^{j+1}
    std::vector<E9C> mu_c((size_t)K2 * TAU), a_pow((size_t)K2 * TAU);
    std::vector<E9PreC> mu_pre((size_t)K2 * TAU), z_pow((size_t)K2 * P.t);
    for (u32 i = 0; i < K2; i++) {
        H9 pa = alpha[i], pz = zeta[i];
        for (u32 d = 0; d < (u32)TAU; d++) { a_pow[(size_t)i * TAU + d] = e9c_from_h9(pa); pa = c->ring.mul9(pa, alpha[i]); }
        for (u32 j = 0; j < P.t; j++) { z_pow[(size_t)i * P.t + j] = e9pre_from_h9(pz, nu); pz = c->ring.mul9(pz, zeta[i]); }
    }
    E9C *d_mu, *d_ap;
    E9PreC *d_mup, *d_zp;
    RET(upload_consts(c, "c_ap", a_pow, &d_ap));
    RET(upload_consts(c, "c_zp", z_pow, &d_zp));
    fe *G[2], *eqb, *zz;
    i64 *partial;
    u64 *od;
    RET(c->tbuf("fold_G1", RE * m, &G[0]));
    RET(c->tbuf("fold_G2", RE * m, &G[1]));
    RET(c->tbuf("fold_eqb", TAU * m, &eqb));
    RET(c->tbuf("fold_zz", (size_t)P.t * RE * n, &zz));
    RET(c->tbuf("round_partial", fold_partial_words(m), &partial));
    od = c->round_out();
    if (!od) return LF_ERR_HIP;
    {
        // G = sum_j M_j (sum_k zeta_k^{j+1} z_k)  +  sum_k sum_d alpha_k^{d+1} fhat_{k,d}   (folding.rs:208-226, utils.rs:524-546): the two sides are
        // independent chains -- the right one runs on the other (idle) stream, as in the Goldilocks driver
        hipStream_t s0 = c->stream(), s1 = (c->lane == 0 && c->sh_world == 1) ? c->st_lane[1] : s0;
        fe *zz1 = zz;
        if (s1 != s0) {
            RET(c->tbuf("fold_zz1", (size_t)P.t * RE * n, &zz1));
            for (int e = 0; e < 2; e++)
                if (!c->ev_prep[e]) HIPCHK(hipEventCreateWithFlags(&c->ev_prep[e], hipEventDisableTiming));
            HIPCHK(hipEventRecord(c->ev_prep[0], s0));           // the challenge powers were uploaded on s0
            HIPCHK(hipStreamWaitEvent(s1, c->ev_prep[0], 0));
        }
        for (int sd = 0; sd < 2; sd++) {
            hipStream_t st = sd ? s1 : s0;
            fe *zb = sd ? zz1 : zz;
            launch_lincomb_z(c->dev, S[sd].z, n, K, d_zp + (size_t)sd * K * P.t, P.t, n, zb, st);
            launch_spmv_sum(c->dev, P.t, c->d_rowptr.data(), c->d_col.data(), c->d_val.data(), zb, (size_t)RE * n, n, G[sd], m, st);
            launch_add_fhat_comb(c->dev, S[sd].planes, N, K, d_ap + (size_t)sd * K * TAU, G[sd], m, st);
        }
        if (s1 != s0) {
            HIPCHK(hipEventRecord(c->ev_prep[1], s1));
            HIPCHK(hipStreamWaitEvent(s0, c->ev_prep[1], 0));
        }
    }
    {
        HostTimer ht(c);
        tr.absorb_label("mu_s");
        for (u32 i = 0; i + 1 < K2; i++) mu[i] = tr.get_challenge();
        mu[K2 - 1] = h9_one();
        tr.absorb_label("beta_s");
        for (u32 i = 0; i < P.s; i++) beta[i] = tr.get_challenge();
    }
    for (u32 i = 0; i < K2; i++) {
        H9 pm = mu[i];
        for (u32 d = 0; d < (u32)TAU; d++) {
            mu_c[(size_t)i * TAU + d] = e9c_from_h9(pm);
            mu_pre[(size_t)i * TAU + d] = e9pre_from_h9(pm, nu);
            pm = c->ring.mul9(pm, mu[i]);
        }
    }
    BB_MARK(" fold challenges");
    RET(upload_consts(c, "c_mu", mu_c, &d_mu));
    RET(upload_consts(c, "c_mup", mu_pre, &d_mup));
    RET(build_eq_dev(c, beta.data(), P.s, eqb));
    c->ev_end(ph);
    if (g_marks.on) { (void)hipStreamSynchronize(c->stream()); BB_MARK(" fold prepare (synced)"); }

    ph = c->ev_begin(14);
    u64 *msgs = proof;
    std::vector<H9> pt(P.s);
    { HostTimer ht(c); sc_prologue(tr, P.s, deg); }
    // working tables (ping-pong): 5 special tables (eqL eqR eqB G1 G2 = 171 planes) + the 2K*9 materialised f-hat tables
    const size_t T5P = 3 * TAU + 2 * RE;
    fe *F[2], *T5[2];
    RET(c->tbuf("fold_T0", T5P * atl(m / 2), &T5[0]));
    RET(c->tbuf("fold_T1", T5P * atl(m / 4), &T5[1]));
    FoldArgs a;
    a.eqL = S[0].eq_r; a.eqR = S[1].eq_r; a.eqB = eqb; a.G1 = G[0]; a.G2 = G[1]; a.ld = m; a.n = m;
    a.p0 = 0; a.pcnt = m / 2; a.pF0 = 0;
    const fe *curF = nullptr;
    size_t ldF = 0;
    int flip = 0;
    // Sharded rounds (same scheme as the Goldilocks driver): rank g evaluates the pairs of its index slice (high bits:
    // pairs (2j,2j+1) stay local, the f-hat tables exist only for that slice), the 5-element partial messages are all-gathered and
    // added mod p, every rank runs the same transcript.  Below 64 pairs per rank the f-hat slices are gathered and the tail is replicated.
    const size_t Gw = (size_t)c->sh_world, gr = (size_t)c->sh_rank;
    bool sharded = Gw > 1;
    // unsharded, rounds >= 4 with many entries: fix_variables of the f-hat tables is fused into the (ALU-bound) round kernel
    const bool fused = Gw == 1 && !c->tn.fold_unfused;
    const size_t fuse_min = c->tn.fuse_min;   // entries; tests lower it
    const fe *prevF = nullptr;
    size_t prevld = 0;
    // rounds 3 and 4 of large unsharded instances never materialise the m/4-entry tables (k_fold_round modes 3 and 4)
    const size_t lut_min = c->tn.lut_min;   // default 2^15
    const bool use_lut = fused && P.s >= 4 && m / 4 >= lut_min && m / 4 >= 4 && !c->tn.fold_no_lut;
    // round 5 on the planes as well (mode 7): round 4 then stores no tables.  From 2^18 rows on, like the Goldilocks driver
    const bool use_r5 = use_lut && !c->tn.fold_no_r4tab && !c->tn.fold_no_r5tab && P.s >= 5 && (N & 3) == 0 && m / 32 >= c->tn.r5_min;
    // f-hat is materialised after two rounds (m/4 entries, F[0]; round r > 3 writes its m/2^(r-1) entries to F[r odd ? 0 : 1]) -- or later: the
    // look-up-table rounds store their first tables in round 4 (m/8, F[1]), with round 5 on the planes too in round 5 (m/16, F[0])
    RET(c->tbuf("fold_F0", (size_t)K2 * TAU * RE * atl(use_lut ? m / 16 : m / 4), &F[0]));
    RET(c->tbuf("fold_F1", (size_t)K2 * TAU * RE * atl(use_r5 ? m / 32 : m / 8), &F[1]));
    fe *d_lut = nullptr;
    int lut_mode = 0;
    // Rounds 1..3 as exact int8 GEMMs on the matrix cores (bb_sv_rounds.hip: the norm part of the message from the bit-plane form of the witnesses, in the split
    // eq form; the G part from the round kernel run without tables).  Unsharded steps whose witness fills whole super-steps.
    const bool use_sv = Gw == 1 && !c->tn.fold_no_sv && N <= m && (N & 3) == 0 && P.s >= 4;
    const size_t sv_min = c->tn.sv_min >= 65536 ? 16384 : c->tn.sv_min;   // (the default threshold is the Goldilocks driver's; a BabyBear pair carries three times the rows)
    // E_i = eq((beta_{i+1}..beta_s), .): one value per pair of round i; E_1 built, E_2.. pair sums (GEMM rounds and the split table rounds)
    fe *svE[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    u32 svE_level = 0;
    auto svE_ensure = [&](u32 level) -> int {
        static const char *const names[5] = {"sv_E1", "sv_E2", "sv_E3", "sv_E4", "sv_E5"};
        for (; svE_level < level && svE_level < 5; svE_level++) {
            const size_t ne = m >> (svE_level + 1);
            RET(c->tbuf(names[svE_level], (size_t)TAU * atl(ne), &svE[svE_level]));
            if (svE_level == 0) RET(build_eq_dev(c, beta.data() + 1, P.s - 1, svE[0]));
            else launch_bb_eq_pairsum(svE[svE_level - 1], atl(m >> svE_level), ne, svE[svE_level], atl(ne), c->stream());
        }
        return LF_OK;
    };
    // rounds 4 / 5 in the split form (k_fold_round SPLIT): three lazy products per table, the host completes the message
    const bool fr_split = Gw == 1 && !c->tn.fold_rounds_no_split && P.s >= 5;
    c->sv_round_mask = 0;
    c->fold_split_mask = 0;
    for (u32 round = 1; round <= P.s; round++) {
        bool fix_fused = false;
        lut_mode = 0;
        if (round > 1) {
            H9 rh = pt[round - 2];
            E9PreC r = e9pre_from_h9(rh, nu);
            size_t nn = a.n / 2, ldn = atl(nn);
            fe *dst = T5[flip];
            if (round == 2) {   // sources are the five separate full-size tables
                launch_fix(c->dev, a.eqL, a.ld, dst, ldn, a.n, 1, r, c->stream());
                launch_fix(c->dev, a.eqR, a.ld, dst + (size_t)TAU * ldn, ldn, a.n, 1, r, c->stream());
                launch_fix(c->dev, a.eqB, a.ld, dst + (size_t)2 * TAU * ldn, ldn, a.n, 1, r, c->stream());
                launch_fix(c->dev, a.G1, a.ld, dst + (size_t)3 * TAU * ldn, ldn, a.n, 8, r, c->stream());
                launch_fix(c->dev, a.G2, a.ld, dst + (size_t)(3 * TAU + RE) * ldn, ldn, a.n, 8, r, c->stream());
            } else {            // source is the previous 171-plane buffer (same layout): one launch over its 19 F_{p^9} rows
                launch_fix(c->dev, a.eqL, a.ld, dst, ldn, a.n, 19, r, c->stream());
            }
            bool gathered = false;
            if (sharded && nn / 2 < Gw * 64) {
                // transition to the replicated tail: gather the fixed f-hat slices (if they exist yet)
                if (round > 3) {
                    fe *fd = F[(round & 1) ? 0 : 1];
                    size_t lcl = ldF / 2;   // local entries after this fix
                    launch_fix(c->dev, curF, ldF, fd, lcl, ldF, K2 * TAU * 8, r, c->stream());
                    size_t planes = (size_t)K2 * TAU * RE, cnt = planes * lcl, words = (cnt + 1) / 2;   // two 32-bit words per u64
                    std::vector<u64> mine(words, 0), all(words * Gw);
                    std::vector<fe> full(planes * nn);
                    HIPCHK(hipMemcpyAsync(mine.data(), fd, cnt * sizeof(fe), hipMemcpyDeviceToHost, c->stream()));
                    HIPCHK(hipStreamSynchronize(c->stream()));
                    RET(c->comm.allgather_host(mine.data(), all.data(), words, c->stream()));
                    for (size_t rk = 0; rk < Gw; rk++) {
                        const fe *src = (const fe *)&all[rk * words];
                        for (size_t w = 0; w < planes; w++) memcpy(&full[w * nn + rk * lcl], src + w * lcl, lcl * sizeof(fe));
                    }
                    HIPCHK(hipMemcpyAsync(fd, full.data(), full.size() * sizeof(fe), hipMemcpyHostToDevice, c->stream()));
                    HIPCHK(hipStreamSynchronize(c->stream()));
                    curF = fd; ldF = nn;
                    gathered = true;
                }
                sharded = false;
            }
            if (!gathered) {
                if (round == 3) {
                    size_t q = sharded ? nn / Gw : nn, j0 = sharded ? gr * q : 0;   // this rank's slice of the m/4 entries
                    if (use_lut) {
                        std::vector<fe> lut(2 * 81 * TAU);
                        build_fold_lut(pt[0], pt[1], c->ring, lut.data());
                        RET(c->tbuf("fold_lut", 2 * 81 * TAU + 8, &d_lut));
                        HIPCHK(hipMemcpyAsync(d_lut, lut.data(), lut.size() * sizeof(fe), hipMemcpyHostToDevice, c->stream()));
                        HIPCHK(hipStreamSynchronize(c->stream()));   // lut is a stack-lifetime buffer
                        lut_mode = 3;
                        curF = nullptr; ldF = atl(q);
                    } else {
                        launch_fold_materialize2(c->dev, S[0].planes, S[1].planes, N, j0, q, K, pt[0], pt[1], c->ring, F[0], c->stream());
                        curF = F[0]; ldF = atl(q);
                    }
                } else if (round > 3) {
                    fe *fd = F[(round & 1) ? 0 : 1];   // round 4 -> F[1], round 5 -> F[0], ...
                    if (use_lut && round == 4) lut_mode = 4;
                    else if (use_r5 && round == 5) lut_mode = 7;
                    else if (fused && ldF >= fuse_min && ldF >= 4 && nn * 2 == ldF) { prevF = curF; prevld = ldF; fix_fused = true; }
                    else launch_fix(c->dev, curF, ldF, fd, atl(ldF / 2), ldF, K2 * TAU * 8, r, c->stream());
                    curF = fd; ldF = atl(ldF / 2);
                }
            }
            a.eqL = dst; a.eqR = dst + (size_t)TAU * ldn; a.eqB = dst + (size_t)2 * TAU * ldn;
            a.G1 = dst + (size_t)3 * TAU * ldn; a.G2 = dst + (size_t)(3 * TAU + RE) * ldn;
            a.ld = ldn; a.n = nn;
            flip ^= 1;
        }
        if (sharded && a.n / 2 < Gw * 64) sharded = false;   // (round 1 of a tiny instance)
        if (sharded) { a.pcnt = a.n / 2 / Gw; a.p0 = gr * a.pcnt; a.pF0 = a.p0; }
        else { a.p0 = 0; a.pcnt = a.n / 2; a.pF0 = 0; }
        size_t ev = c->ev_begin(0);
        const int svV = 1 << (round - 1);
        bool sv_done = false;
        if (use_sv && (int)round <= c->tn.sv_rounds && round <= 3 && a.pcnt >= sv_min && bbsv_shape_ok(svV, a.pcnt, K)) {
            // weights W_b = eq((r_1..r_{i-1}), b), their digit-monomial coefficients, c_i = prod_{k<i} eq(beta_k, r_k), w_h = c_i eq(beta_i, h)
            const fe nuM = from_canon(nu);
            std::vector<E9> W((size_t)svV, e9_from_h9(h9_one()));
            for (int b = 0; b < svV; b++)
                for (u32 j = 0; j + 1 < round; j++) W[b] = e9_mul(W[b], e9_from_h9(((b >> j) & 1) ? pt[j] : h9_sub(h9_one(), pt[j])), nuM);
            std::vector<fe> coef;
            bbsv_build_coef(svV, W.data(), nuM, coef);
            E9 cc = e9_from_h9(h9_one());
            for (u32 k2 = 1; k2 < round; k2++) {
                const E9 b = e9_from_h9(beta[k2 - 1]), r = e9_from_h9(pt[k2 - 1]), one = e9_from_h9(h9_one());
                cc = e9_mul(cc, e9_add(e9_mul(e9_sub(one, b), e9_sub(one, r), nuM), e9_mul(b, r, nuM)), nuM);
            }
            const E9 bi = e9_from_h9(beta[round - 1]);
            const E9 w0e = e9_mul(cc, e9_sub(e9_from_h9(h9_one()), bi), nuM), w1e = e9_mul(cc, bi, nuM);
            E9C w0, w1;
            for (int q = 0; q < TAU; q++) { w0.c[q] = w0e.c[q]; w1.c[q] = w1e.c[q]; }
            fe *d_coef, *svtp;
            u64 *gtmp;
            unsigned char *sveb;
            int32_t *svpart, *svtot;
            RET(c->tbuf("sv_coef", coef.size() + 8, &d_coef));
            RET(c->tbuf("sv_gtmp", (size_t)5 * RE + 8, &gtmp));
            RET(c->tbuf("sv_tp", bbsv_tp_words(K), &svtp));
            RET(c->tbuf("sv_eb", bbsv_eb_bytes(a.pcnt), &sveb));
            RET(c->tbuf("sv_part", bbsv_part_words(svV, K), &svpart));
            RET(c->tbuf("sv_tot", bbsv_tot_words(svV, K), &svtot));
            HIPCHK(hipMemcpyAsync(d_coef, coef.data(), coef.size() * sizeof(fe), hipMemcpyHostToDevice, c->stream()));
            HIPCHK(hipStreamSynchronize(c->stream()));   // coef is a stack-lifetime buffer
            if (!svbits[0])
                for (int sd = 0; sd < 2; sd++) {
                    RET(c->tbuf(sd ? "sv_bits_R" : "sv_bits_L", bbsv_bits_words(N, K), &svbits[sd]));
                    launch_bbsv_bits(S[sd].planes, N, N, K, svbits[sd], c->stream());
                }
            RET(svE_ensure(round));
            i64 *gpartial;
            RET(c->tbuf("sv_gpartial", red_partial_words(5 * RE), &gpartial));
            // the G part (eqL G1 + eqR G2: the round kernel without tables) on the other, idle stream next to the GEMM chain
            hipStream_t sg = (c->lane == 0 && c->st_lane[1]) ? c->st_lane[1] : c->stream();
            hipEvent_t g_ready = nullptr;
            if (sg != c->stream()) {
                for (int e = 0; e < 2; e++)
                    if (!c->ev_prep[e]) HIPCHK(hipEventCreateWithFlags(&c->ev_prep[e], hipEventDisableTiming));
                HIPCHK(hipEventRecord(c->ev_prep[0], c->stream()));      // the special tables of this round were fixed on this stream
                HIPCHK(hipStreamWaitEvent(sg, c->ev_prep[0], 0));
            }
            launch_fold_round_g(c->dev, a, gpartial, gtmp, sg);
            if (sg != c->stream()) { HIPCHK(hipEventRecord(c->ev_prep[1], sg)); g_ready = c->ev_prep[1]; }
            if (launch_bbsv_round(c->dev, svV, svbits[0], svbits[1], N, svE[round - 1], atl(m >> round), a.pcnt, K, d_mu, d_coef, w0, w1, sveb, svpart, svtot, svtp, gtmp, od,
                                  c->stream(), g_ready) == 0) {
                sv_done = true;
                c->sv_round_mask |= 1u << (round - 1);
            }
        }
        // split form of this round's table kernel?  (modes 6 / 7; c_i and beta_i must be invertible for the host's completion)
        bool split_now = false;
        H9 sp_c = h9_one(), sp_cinv = h9_one(), sp_binv = h9_one();
        const fe *Er = nullptr;
        size_t ldEr = 0;
        // (round 5 from the planes runs on two lanes per pair -- k_fold_round5_2l: all four products, unsplit -- unless LF_FOLD_R5_ONE_LANE=1)
        if (fr_split && !sv_done && round >= 2 && round <= 5 && !sharded && (lut_mode == 4 && !c->tn.fold_no_r4tab) &&
            a.pcnt >= c->tn.fold_split_min) {
            for (u32 k2 = 1; k2 < round; k2++) {
                const H9 b = beta[k2 - 1], r = pt[k2 - 1];
                sp_c = c->ring.mul9(sp_c, h9_add(c->ring.mul9(h9_sub(h9_one(), b), h9_sub(h9_one(), r)), c->ring.mul9(b, r)));
            }
            if (h9_inv(sp_c, nu, &sp_cinv) && h9_inv(beta[round - 1], nu, &sp_binv)) {
                RET(svE_ensure(round));
                Er = svE[round - 1]; ldEr = atl(m >> round);
                split_now = true;
                c->fold_split_mask |= 1u << (round - 1);
            }
        }
        if (sv_done) {}
        else if (round == 1) launch_fold_round1(c->dev, a, S[0].planes, S[1].planes, N, K, d_mu, partial, od, c->stream());
        else if (round == 2) launch_fold_round2(c->dev, a, S[0].planes, S[1].planes, N, K, d_mu, pt[0], c->ring, partial, od, c->stream());
        else if (lut_mode == 3) {
            fe *mutab;
            RET(c->tbuf("fold_mutab", (size_t)3 * K2 * TAU * 81 * 12, &mutab));
            launch_fold_round_lut_mu(c->dev, a, S[0].planes, S[1].planes, N, d_lut, mutab, K, d_mup, partial, od, c->stream());
        }
        else if (lut_mode == 4 && !c->tn.fold_no_r4tab) {
            fe *r4sq, *r4mt;
            RET(c->tbuf("fold_r4sq", (size_t)6561 * 12, &r4sq));
            RET(c->tbuf("fold_r4mt", (size_t)K2 * TAU * 162 * 12, &r4mt));
            launch_fold_round_lut_fix_tab(c->dev, a, S[0].planes, S[1].planes, N, d_lut, pt[round - 2], c->ring, r4sq, r4mt, use_r5 ? nullptr : (fe *)curF, ldF, K, d_mup, partial, od,
                                          c->stream(), Er, ldEr);
        } else if (lut_mode == 7) {
            fe *r5xx, *r5yy, *r5mt;
            RET(c->tbuf("fold_r5xx", (size_t)6561 * 12, &r5xx));
            RET(c->tbuf("fold_r5yy", (size_t)6561 * 12, &r5yy));
            RET(c->tbuf("fold_r5mt", (size_t)K2 * TAU * 324 * 12, &r5mt));
            launch_fold_round_lut_fix5(c->dev, a, S[0].planes, S[1].planes, N, d_lut, pt[round - 3], pt[round - 2], c->ring, r5xx, r5yy, r5mt, (fe *)curF, ldF, K, d_mup, partial, od,
                                       c->stream(), Er, ldEr);
        } else if (lut_mode == 4) launch_fold_round_lut_fix(c->dev, a, S[0].planes, S[1].planes, N, d_lut, pt[round - 2], c->ring, (fe *)curF, ldF, K, d_mup, partial, od, c->stream());
        else if (fix_fused) launch_fold_round_fix(c->dev, a, prevF, prevld, pt[round - 2], c->ring, (fe *)curF, ldF, K, d_mup, partial, od, c->stream());
        else launch_fold_round(c->dev, a, curF, ldF, K, d_mup, partial, od, c->stream());
        if (split_now) {   // the G part of the message (eqL G1 + eqR G2 at X = 0..4) from the round kernel run without tables, behind the three sums of the table kernel
            i64 *gpartial;
            RET(c->tbuf("sv_gpartial", red_partial_words(5 * RE), &gpartial));
            launch_fold_round_g(c->dev, a, gpartial, od + 5 * RE, c->stream());
        }
        c->ev_end(ev);
        u64 *evs = msgs + (size_t)(round - 1) * (deg + 1) * RE;
        HIPCHK(hipStreamSynchronize(c->stream()));            // message is in mapped host memory
        if (split_now) {
            // od[e][slot] (e = 0..2): A_e = sum_p E[p] C_e(p); od[5 + X][slot]: the G part.  g(X) = c l(X) (A0 + A1 X + A2 X^2 + A3 X^3) + G(X), l(X) = eq(beta_i, X);
            // A3 from g(0) + g(1) = the previous message at its challenge (interpolated: no assumption on the claimed sum)
            HostTimer ht2(c);
            const H9 bi = beta[round - 1], obi = h9_sub(h9_one(), bi), x = pt[round - 2];
            H9 wS[5];
            for (u32 j2 = 0; j2 <= deg; j2++) {
                H9 num = h9_one();
                u64 den = 1;
                for (u32 k2 = 0; k2 <= deg; k2++) {
                    if (k2 == j2) continue;
                    H9 xk = x; xk.c[0] = hsub(xk.c[0], k2);
                    num = c->ring.mul9(num, xk);
                    den = hmul(den, j2 > k2 ? (u64)(j2 - k2) : BB_P - (u64)(k2 - j2));
                }
                wS[j2] = h9_scale(num, hinv(den));
            }
            H9 cl[5];   // c l(X)
            {
                H9 l = obi;
                const H9 dl = h9_sub(bi, obi);
                for (u32 X = 0; X <= deg; X++) { cl[X] = c->ring.mul9(sp_c, l); l = h9_add(l, dl); }
            }
            const u64 *pe = msgs + (size_t)(round - 2) * (deg + 1) * RE, *gev = od + 5 * RE;
            auto ld = [&](const u64 *b, u32 e, u32 slot) { return h9_load(b + (size_t)e * RE + TAU * slot); };
            for (u32 slot = 0; slot < 8; slot++) {
                H9 Sv = ld(pe, 0, slot);
                Sv = c->ring.mul9(wS[0], Sv);
                for (u32 j2 = 1; j2 <= deg; j2++) Sv = h9_add(Sv, c->ring.mul9(wS[j2], ld(pe, j2, slot)));
                const H9 A0 = ld(od, 0, slot), A1 = ld(od, 1, slot), A2 = ld(od, 2, slot);
                const H9 Gsum = h9_add(ld(gev, 0, slot), ld(gev, 1, slot));
                const H9 T1 = c->ring.mul9(h9_sub(c->ring.mul9(h9_sub(Sv, Gsum), sp_cinv), c->ring.mul9(obi, A0)), sp_binv);
                const H9 A3 = h9_sub(h9_sub(h9_sub(T1, A0), A1), A2);
                for (u32 X = 0; X <= deg; X++) {
                    const H9 T = h9_add(A0, h9_scale(h9_add(A1, h9_scale(h9_add(A2, h9_scale(A3, X)), X)), X));
                    const H9 g = h9_add(c->ring.mul9(cl[X], T), ld(gev, X, slot));
                    memcpy(evs + (size_t)X * RE + TAU * slot, g.c, sizeof(g.c));
                }
            }
        } else
        memcpy(evs, od, (size_t)(deg + 1) * RE * 8);
        if (sharded) RET(exchange_modsum(c, evs, (size_t)(deg + 1) * RE));
        HostTimer ht(c);
        pt[round - 1] = sc_round_transcript(tr, evs, deg + 1);
        if (round <= 6 || round == 10) { char nm[32]; snprintf(nm, sizeof nm, "  round %u", round); BB_MARK(nm); }
    }
    c->ev_end(ph);
    BB_MARK(" fold sumcheck");

    ph = c->ev_begin(15);
    // theta, eta at r_0 (folding.rs:236-256)
    u64 *theta = proof + (size_t)P.s * (deg + 1) * RE, *eta = theta + (size_t)K2 * TAU * RE;
    fe *eq0, *q;
    i64 *red;
    u64 *sm;
    RET(c->tbuf("fold_eq0", TAU * m, &eq0));
    RET(c->tbuf("dec_q", (size_t)P.t * RE * n, &q));
    RET(c->tbuf("red_partial", red_partial_words(16 * RE * TAU), &red));
    RET(c->tbuf("dec_small", 16 * RE * TAU + 16 * 4 * RE, &sm));
    RET(build_eq_dev(c, pt.data(), P.s, eq0));
    // the other stream is idle here: every second q_j = M_j^T eq(r_o) is gathered there, and the eta products of the right side run there (as in the Goldilocks driver)
    hipStream_t s1f = (c->lane == 0 && c->sh_world == 1 && c->st_lane[1]) ? c->st_lane[1] : c->stream();
    if (s1f != c->stream()) {
        for (int e = 0; e < 2; e++)
            if (!c->ev_prep[e]) HIPCHK(hipEventCreateWithFlags(&c->ev_prep[e], hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->ev_prep[0], c->stream()));           // eq(r_o) is built
        HIPCHK(hipStreamWaitEvent(s1f, c->ev_prep[0], 0));
    }
    for (u32 j = 0; j < P.t; j++)
        launch_spmv_t_eq(c->dev, c->d_colptr[j], c->d_rowidx[j], c->d_valT[j], eq0, m, q + (size_t)j * RE * n, n, (j & 1) ? s1f : c->stream());
    if (s1f != c->stream()) {
        HIPCHK(hipEventRecord(c->ev_prep[1], s1f));
        HIPCHK(hipStreamWaitEvent(c->stream(), c->ev_prep[1], 0));
    }
    // theta for both sides first, then eta; the host absorbs theta while the GPU still computes the eta dot products
    u64 *fsm;
    size_t nth = (size_t)K2 * TAU * RE, net = (size_t)K2 * P.t * RE;
    RET(c->tbuf("fold_small", nth + net + 64, &fsm));
    u64 *hp = c->arena_alloc(nth + net);
    if (!hp) return LF_ERR_HIP;
    u64 *d_theta = fsm, *d_eta = fsm + nth;
    (void)sm;
    // theta = f-hat_{k,d}(r_o): the sumcheck's f-hat tables, fixed at r_1..r_{s-1}, have two entries left -- one more fix gives the
    // evaluations (exact arithmetic: the same words as evaluate_mles on the witness).  LF_THETA_EVAL=1 / fewer than 4 variables:
    // stand-alone evaluation.
    if (P.s >= 4 && curF && ldF == 2 && !c->tn.theta_eval) launch_fix_final(c->dev, curF, ldF, K2 * TAU * 8, e9pre_from_h9(pt[P.s - 1], c->ring.T.nu), d_theta, c->stream());
    else
        for (int sd = 0; sd < 2; sd++) RET(coef_eval_bits_dev(c, S[sd].planes, N, eq0, m, K, red, d_theta + (size_t)sd * K * TAU * RE));
    HIPCHK(hipMemcpyAsync(hp, d_theta, nth * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipEventRecord(c->ev_side[0], c->stream()));
    if (s1f != c->stream()) {
        unsigned char *ybq = nullptr;   // the digits of q are the same for both sides: packed once, before the streams part
        if (!c->tn.dot_valu && n >= c->tn.dot_min && P.t <= 3 && K <= 16 && ((((size_t)S[0].z) ^ ((size_t)S[1].z)) & 7) == 0) {
            RET(c->tbuf("dot_yb", bbdot_i8_yb_bytes(n + 1), &ybq));
            if (launch_dot_pack_y(S[0].z, q, n, P.t, n, ybq, c->stream()) != 0) ybq = nullptr;
        }
        HIPCHK(hipEventRecord(c->ev_prep[0], c->stream()));           // q (and its digits) are ready
        HIPCHK(hipStreamWaitEvent(s1f, c->ev_prep[0], 0));
        i64 *red1;
        RET(c->tbuf("red_partial1", red_partial_words(16 * RE * TAU), &red1));
        RET(dot_batch_dev(c, S[1].z, n, K, q, n, P.t, n, red1, d_eta + (size_t)K * P.t * RE, s1f, "_1", ybq));
        HIPCHK(hipEventRecord(c->ev_prep[1], s1f));
        RET(dot_batch_dev(c, S[0].z, n, K, q, n, P.t, n, red, d_eta, nullptr, "", ybq));
        HIPCHK(hipStreamWaitEvent(c->stream(), c->ev_prep[1], 0));
    } else
        for (int sd = 0; sd < 2; sd++) RET(dot_batch_dev(c, S[sd].z, n, K, q, n, P.t, n, red, d_eta + (size_t)sd * K * P.t * RE));
    HIPCHK(hipMemcpyAsync(hp + nth, d_eta, net * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipEventSynchronize(c->ev_side[0]));
    BB_MARK("  theta down");
    memcpy(theta, hp, nth * 8);
    {
        HostTimer ht(c);
        tr.absorb_ring(theta, (size_t)K2 * TAU);
    }
    BB_MARK("  theta absorbed");
    HIPCHK(hipStreamSynchronize(c->stream()));
    BB_MARK("  eta down");
    memcpy(eta, hp + nth, net * 8);
    std::vector<u64> rho_c((size_t)K2 * RE, 0), rho((size_t)K2 * RE);
    std::vector<int8_t> rho8((size_t)K2 * 24, 0);
    {
        HostTimer ht(c);
        tr.absorb_ring(eta, (size_t)K2 * P.t);
        tr.absorb_label("rho_s");   // get_rhos (folding/utils.rs:116-131)
        for (u32 i = 0; i + 1 < K2; i++) tr.get_short_challenge(&rho_c[(size_t)i * RE]);
        rho_c[(size_t)(K2 - 1) * RE] = 1;
        for (u32 i = 0; i < K2; i++) {
            c->ring.crt(&rho_c[(size_t)i * RE], &rho[(size_t)i * RE]);
            for (int q2 = 0; q2 < 24; q2++) {
                u64 v = rho_c[(size_t)i * RE + q2];
                rho8[(size_t)i * 24 + q2] = (int8_t)(v > BB_P / 2 ? -(int64_t)(BB_P - v) : (int64_t)v);
            }
        }
    }
    // f_0 in the coefficient domain -> new witness
    int8_t *d_rho;
    RET(c->tbuf("c_rho", (size_t)K2 * 24 + 64, &d_rho));
    HIPCHK(hipMemcpyAsync(d_rho, rho8.data(), rho8.size(), hipMemcpyHostToDevice, c->stream()));
    int32_t *npl;
    RET(lf_planes_alloc(c->owner, N * RE * 4, &npl));
    launch_fold_witness(S[0].planes, S[1].planes, N, K, d_rho, npl, c->stream());
    // Witness::from_f (arith.rs:299-313): f = CRT(f_coeff) and w_ccs = CRT(recompose(f_coeff, B, L)) of the folded witness, behind compute_f_0 on the same stream
    fe *nf = nullptr, *nw = nullptr;
    const size_t nf_bytes = N * RE * sizeof(fe), nw_bytes = (size_t)P.wit_len * RE * sizeof(fe);
    RET(lf_planes_alloc(c->owner, nf_bytes, (int32_t **)&nf));
    RET(lf_planes_alloc(c->owner, nw_bytes, (int32_t **)&nw));
    launch_recompose_crt(c->dev, npl, N, (u32)N, 1, P.B, 1, 0, nf, N, 0, c->stream());
    launch_recompose_crt(c->dev, npl, N, P.wit_len, P.L, P.B, 1, 0, nw, P.wit_len, 0, c->stream());
    BB_MARK("  eta absorbed, rho drawn, fold_witness enqueued");

    // compute_v0_u0_x0_cm_0 (folding/utils.rs:460-521) on the host while the GPU folds the witness
    {
    HostTimer ht(c);
    u64 *o = lcccs_out;
    for (u32 i = 0; i < P.s; i++, o += RE) BbHostRing::from_h9(pt[i], o);
    {   // v_0 = rot_lin_combination(rho_coeff, theta) (cyclotomic-rings/src/rotation.rs:85-104)
        // the rotations of a short challenge stay small signed integers (X^72 = X^36 - 1 adds at most one more term per step) and theta words are < 2^31:
        // plain 64-bit integer multiply-accumulates, one reduction per output word (32 * 72 terms of < 2^31 * 2^12 fit easily)
        // As one polynomial product per i: full[a + b] += rho_a theta_b over the (at most 24 non-zero) coefficients of rho and all of theta -- the inner loop is one
        // contiguous multiply-add over theta's 72 x 9 words -- and ONE reduction of the degree-142 product by X^72 = X^36 - 1 at the end (the rotation-by-rotation
        // form walked 72 x 72 pairs per i with rotations that fill up: twice the multiply-adds, none of them contiguous).  |full| < 72 * 32 * 2^5 * 2^31 < 2^48.
        std::vector<int64_t> acc((size_t)(2 * RE) * TAU, 0);
        std::vector<u64> res((size_t)RE * TAU, 0);   // res[j] in F_{p^9}
        for (u32 i = 0; i < K2; i++) {
            const u64 *th = theta + (size_t)i * TAU * RE;
            for (int a = 0; a < RE; a++) {
                const u64 rc = rho_c[(size_t)i * RE + a] % BB_P;
                const int64_t ra = rc > BB_P / 2 ? (int64_t)rc - (int64_t)BB_P : (int64_t)rc;
                if (!ra) continue;
                int64_t *dst = acc.data() + (size_t)a * TAU;
                for (int x = 0; x < RE * TAU; x++) dst[x] += (int64_t)th[x] * ra;
            }
        }
        for (int d = 2 * RE - 2; d >= RE; d--)
            for (int q2 = 0; q2 < TAU; q2++) {
                const int64_t v = acc[(size_t)d * TAU + q2];
                acc[(size_t)(d - RE / 2) * TAU + q2] += v;
                acc[(size_t)(d - RE) * TAU + q2] -= v;
            }
        for (size_t x = 0; x < res.size(); x++) { const int64_t r = acc[x] % (int64_t)BB_P; res[x] = (u64)(r < 0 ? r + (int64_t)BB_P : r); }
        memcpy(o, res.data(), res.size() * 8);
        o += (size_t)TAU * RE;
    }
    u64 tmp[RE];
    auto part = [&](u32 i) { return &S[i < K ? 0 : 1].lcccs[(size_t)(i % K) * ll * RE]; };
    for (u32 q2 = 0; q2 < P.kappa; q2++, o += RE) {
        memset(o, 0, RE * 8);
        for (u32 i = 0; i < K2; i++) { c->ring.mul_ntt(part(i) + ((size_t)P.s + TAU + q2) * RE, &rho[(size_t)i * RE], tmp); BbHostRing::add(o, tmp, o); }
    }
    for (u32 j = 0; j < P.t; j++, o += RE) {
        memset(o, 0, RE * 8);
        for (u32 i = 0; i < K2; i++) { c->ring.mul_ntt(&rho[(size_t)i * RE], eta + ((size_t)i * P.t + j) * RE, tmp); BbHostRing::add(o, tmp, o); }
    }
    for (u32 q2 = 0; q2 < P.l + 1; q2++, o += RE) {
        memset(o, 0, RE * 8);
        for (u32 i = 0; i < K2; i++) { c->ring.mul_ntt(&rho[(size_t)i * RE], part(i) + ((size_t)P.s + TAU + P.kappa + P.t + q2) * RE, tmp); BbHostRing::add(o, tmp, o); }
    }
    }
    BB_MARK("  folded instance on the host");
    HIPCHK(hipStreamSynchronize(c->stream()));
    *w_out = new lf_witness{c->owner, npl, N, lf_ctx_device(c->owner), N * RE * 4};
    if (nf) { (*w_out)->f_ntt = (uint64_t *)nf; (*w_out)->f_bytes = nf_bytes; (*w_out)->w_ccs = (uint64_t *)nw; (*w_out)->w_bytes = nw_bytes; }
    c->ev_end(ph);
    return LF_OK;
}

int BbCtx::linearize(BbTranscript &tr, const uint64_t *cccs, const lf_witness *wit, uint64_t *lcccs_out, uint64_t *lin_proof_out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    if (wit->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    c->tn = Tunables::read((size_t)1 << 15);
    c->ev_reset();
    c->host_tr_ms = 0;
    int rc = linearize_impl(c, tr, cccs, wit, lcccs_out, lin_proof_out, nullptr);
    c->ev_collect();
    return rc;
}

int BbCtx::fold_step(BbTranscript &tr, const uint64_t *acc, const lf_witness *w_acc, const uint64_t *cm_i, const lf_witness *w_i,
                     uint64_t *lcccs_out, lf_witness **w_out, uint64_t *proof) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs || !c->dAb) return LF_ERR_STATE;
    const lf_params &P = c->P;
    if (c->kappa != P.kappa || c->nA_total != c->N || w_acc->N != c->N || w_i->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    std::vector<H9> rL;
    if (!lcccs_point(P, acc, rL)) return LF_ERR_UNSUPPORTED;   // evaluation points are always diagonal challenges
    c->tn = Tunables::read((size_t)1 << 15);
    c->ev_reset();
    c->host_tr_ms = 0;
    size_t tot = c->ev_begin(17);
    size_t ll = bb_lcccs_len(&P);
    u64 *lin_proof = proof, *decl = lin_proof + lin_proof_len(&P) * RE, *decr = decl + dec_proof_len(&P) * RE, *foldp = decr + dec_proof_len(&P) * RE;
    std::vector<u64> lin(ll * RE);
    fe *eq_r_R = nullptr;
    SideState S[2];
    // Schedule (transcript order is fixed, compute order is not).  Lane 1 (high-priority stream) gets everything that does not
    // depend on the linearization, queued up front: the left decomposition and the RIGHT commit (a function of w_i only); lane 0
    // runs the latency-bound linearization rounds meanwhile and then the right evaluations at the new point; the host absorbs the
    // left decomposition while the GPU still works on the right one.
    c->arena_used[0] = c->arena_used[1] = 0;
    DecPending pdL, pdR;
    pdL.side = 0; pdR.side = 1;
    c->lane = 1;
    g_marks.start();
    // (LF_BB_EVALS_FIRST=1: the left evaluations before the left commit -- the two large linearization rounds then run next to them instead of next to a commit)
    int rc = LF_OK;
    if (rc == LF_OK) rc = dec_enqueue_commit(c, w_acc, pdL);
    BB_MARK("L1: left commit enqueued");
    if (rc == LF_OK) rc = dec_enqueue_evals(c, acc, rL, w_acc, "L", nullptr, S[0], decl, pdL);
    BB_MARK("L1: left evals enqueued");
    if (rc == LF_OK) rc = dec_enqueue_commit(c, w_i, pdR);
    BB_MARK("L1: right commit enqueued");
    c->lane = 0;
    c->lin_blocks = 0;
    {   // absorb_public_input (nifs.rs:175-197) -- while the GPU already works on the left decomposition
        HostTimer ht(c);
        tr.absorb_label("acc");
        tr.absorb_ring(acc, ll);
        tr.absorb_label("cm_i");
        tr.absorb_ring(cm_i, bb_cccs_len(&P));
    }
    c->vs_keep = true;
    BB_MARK("public input absorbed");
    if (rc == LF_OK) rc = linearize_impl(c, tr, cm_i, w_i, lin.data(), lin_proof, &eq_r_R);
    BB_MARK("linearization done");
    c->vs_keep = false;
    std::vector<H9> rR;
    if (rc == LF_OK) {
        lcccs_point(P, lin.data(), rR);
        rc = dec_enqueue_evals(c, lin.data(), rR, w_i, "R", eq_r_R, S[1], decr, pdR);
    }
    c->vs_wit = nullptr;
    c->lin_blocks = 0;
    BB_MARK("right evals enqueued");
    if (rc == LF_OK) rc = dec_finish(c, tr, acc, S[0], decl, pdL);
    BB_MARK("left absorb done");
    if (rc == LF_OK) rc = dec_finish(c, tr, lin.data(), S[1], decr, pdR);
    BB_MARK("right absorb done");
    (void)hipStreamSynchronize(c->st_lane[1]);
    if (rc == LF_OK) rc = fold_impl(c, tr, S, lcccs_out, w_out, foldp);
    BB_MARK("fold done");
    c->ev_end(tot);
    c->ev_collect();
    return rc;
}

// LFDecompositionProver::prove (nifs/decomposition.rs:33-88) as its own entry point
int BbCtx::decomposition_prove(BbTranscript &tr, const uint64_t *lcccs, const lf_witness *wit, uint64_t *lcccs_s_out, uint64_t *dec_proof_out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs || !c->dAb) return LF_ERR_STATE;
    const lf_params &P = c->P;
    if (c->kappa != P.kappa || c->nA_total != c->N || wit->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    std::vector<H9> r;
    if (!lcccs_point(P, lcccs, r)) return LF_ERR_UNSUPPORTED;
    c->tn = Tunables::read((size_t)1 << 15);
    c->ev_reset();
    c->host_tr_ms = 0;
    c->arena_used[0] = c->arena_used[1] = 0;
    c->lane = 0;
    DecPending pd;
    pd.side = 0;
    SideState S;
    RET(dec_enqueue_commit(c, wit, pd));
    RET(dec_enqueue_evals(c, lcccs, r, wit, "L", nullptr, S, dec_proof_out, pd));
    RET(dec_finish(c, tr, lcccs, S, dec_proof_out, pd));
    if (lcccs_s_out) memcpy(lcccs_s_out, S.lcccs.data(), S.lcccs.size() * 8);
    c->ev_collect();
    return LF_OK;
}

// LFFoldingProver::prove (nifs/folding.rs:42-130) as its own entry point (see lf_folding_prove)
int BbCtx::folding_prove(BbTranscript &tr, const uint64_t *lcccs_s, const lf_witness *w_left, const lf_witness *w_right, uint64_t *lcccs_out,
                         lf_witness **w_out, uint64_t *fold_proof_out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    const lf_params &P = c->P;
    if (w_left->N != c->N || w_right->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    c->tn = Tunables::read((size_t)1 << 15);
    c->ev_reset();
    c->host_tr_ms = 0;
    c->arena_used[0] = c->arena_used[1] = 0;
    c->lane = 0;
    const size_t ll = bb_lcccs_len(&P);
    const u32 K = P.K, hl = P.l + 1;
    SideState S[2];
    for (int sd = 0; sd < 2; sd++) {
        const u64 *base = lcccs_s + (size_t)sd * K * ll * RE;
        std::vector<H9> r;
        if (!lcccs_point(P, base, r)) return LF_ERR_UNSUPPORTED;
        for (u32 k = 1; k < K; k++)
            if (memcmp(base, base + (size_t)k * ll * RE, (size_t)P.s * RE * 8) != 0) return LF_ERR_INVALID;
        const lf_witness *w = sd ? w_right : w_left;
        fe *z, *eq_r;
        RET(c->tbuf(sd ? "z_R" : "z_L", (size_t)K * RE * c->n, &z));
        RET(c->tbuf(sd ? "eq_r_R" : "eq_r_L", TAU * c->m, &eq_r));
        std::vector<u64> heads((size_t)K * hl * RE);
        for (u32 k = 0; k < K; k++)
            memcpy(&heads[(size_t)k * hl * RE], base + ((size_t)k * ll + P.s + TAU + P.kappa + P.t) * RE, (size_t)hl * RE * 8);
        RET(build_z(c, w->planes, K, 1, heads.data(), z));
        RET(build_eq_dev(c, r.data(), P.s, eq_r));
        S[sd].planes = w->planes; S[sd].z = z; S[sd].eq_r = eq_r;
        S[sd].lcccs.assign(base, base + (size_t)K * ll * RE);
    }
    int rc = fold_impl(c, tr, S, lcccs_out, w_out, fold_proof_out);
    c->ev_collect();
    return rc;
}

// ---- generic linearization-shaped sumcheck through the ABI -----------------------------------------------------------------------
int BbCtx::sumcheck_lin_begin(const uint64_t *tables, const uint64_t *eq_point) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    size_t m = c->m;
    fe *mz, *eqb;
    RET(c->tbuf("sc_tab0", (size_t)P.t * RE * m, &mz));
    RET(c->tbuf("sc_eq0", TAU * m, &eqb));
    for (u32 j = 0; j < P.t; j++) RET(up_ring(c, tables + (size_t)j * m * RE, m, mz + (size_t)j * RE * m));
    std::vector<H9> pt(P.s);
    for (u32 i = 0; i < P.s; i++) pt[i] = h9_load(eq_point + (size_t)TAU * i);
    RET(build_eq_dev(c, pt.data(), P.s, eqb));
    c->sc_round = 0; c->sc_n = m; c->sc_cur = 0;
    return LF_OK;
}
int BbCtx::sumcheck_lin_round(const uint64_t *r_prev, uint64_t *evals_out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->sc_round < 0 || c->sc_round >= (int)c->P.s) return LF_ERR_STATE;
    if ((c->sc_round == 0) != (r_prev == nullptr)) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    size_t m = c->m;
    fe *tab[2], *eq[2];
    i64 *partial;
    u64 *od;
    RET(c->tbuf("sc_tab0", (size_t)P.t * RE * m, &tab[0]));
    RET(c->tbuf("sc_tab1", (size_t)P.t * RE * atl(m / 2), &tab[1]));
    RET(c->tbuf("sc_eq0", TAU * m, &eq[0]));
    RET(c->tbuf("sc_eq1", TAU * atl(m / 2), &eq[1]));
    RET(c->tbuf("round_partial", red_partial_words(5 * RE), &partial));
    RET(c->tbuf("round_out", 5 * RE, &od));
    if (r_prev) {
        E9PreC r = e9pre_from_h9(h9_load(r_prev), c->ring.T.nu);
        int src = c->sc_cur, dst = src ^ 1;
        size_t ldi = c->sc_n == m ? m : atl(c->sc_n);
        launch_fix(c->dev, tab[src], ldi, tab[dst], atl(c->sc_n / 2), c->sc_n, P.t * 8, r, c->stream());
        launch_fix(c->dev, eq[src], ldi, eq[dst], atl(c->sc_n / 2), c->sc_n, 1, r, c->stream());
        c->sc_cur = dst; c->sc_n /= 2;
    }
    size_t ld = c->sc_n == m ? m : atl(c->sc_n);
    launch_lin_round(c->dev, c->desc, tab[c->sc_cur], ld, eq[c->sc_cur], ld, c->sc_n, P.d + 1, partial, od, c->stream());
    c->sc_round++;
    return down_small(c, od, (size_t)(P.d + 2) * RE, evals_out);
}
// ---- the folding sumcheck through the ABI (see lf_sumcheck_fold_* in lf_capi.cpp / include/lfhip.h) -----------------------------
int BbCtx::sumcheck_fold_begin(const uint64_t *tables, const uint64_t *mu) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    const size_t m = c->m;
    if (m < 2) return LF_ERR_UNSUPPORTED;
    const u32 K2 = 2 * P.K;
    static const int eq_idx[3] = {0, 2, 4};
    for (int e = 0; e < 3; e++) {
        const u64 *tb = tables + (size_t)eq_idx[e] * m * RE;
        for (size_t i = 0; i < m; i++)
            for (int sl = 1; sl < 8; sl++)
                if (memcmp(tb + i * RE, tb + i * RE + TAU * sl, TAU * 8) != 0) return LF_ERR_UNSUPPORTED;
    }
    const size_t T5P = 3 * TAU + 2 * RE;
    fe *T, *F, *tmp;
    RET(c->tbuf("sf_T0", T5P * m, &T));
    RET(c->tbuf("sf_F0", (size_t)K2 * TAU * RE * m, &F));
    RET(c->tbuf("sf_tmp", RE * m, &tmp));
    for (int e = 0; e < 3; e++) {
        RET(up_ring(c, tables + (size_t)eq_idx[e] * m * RE, m, tmp));
        HIPCHK(hipMemcpyAsync(T + (size_t)TAU * e * m, tmp, TAU * m * sizeof(fe), hipMemcpyDeviceToDevice, c->stream()));
        HIPCHK(hipStreamSynchronize(c->stream()));
    }
    RET(up_ring(c, tables + (size_t)1 * m * RE, m, T + (size_t)3 * TAU * m));
    RET(up_ring(c, tables + (size_t)3 * m * RE, m, T + (size_t)(3 * TAU + RE) * m));
    for (u32 i = 0; i < K2 * TAU; i++) RET(up_ring(c, tables + (size_t)(5 + i) * m * RE, m, F + (size_t)i * RE * m));
    std::vector<E9PreC> mu_pre((size_t)K2 * TAU);
    for (u32 i = 0; i < K2; i++) {
        H9 mi = h9_load(mu + (size_t)TAU * i), pm = mi;
        for (u32 d = 0; d < (u32)TAU; d++) { mu_pre[(size_t)i * TAU + d] = e9pre_from_h9(pm, c->ring.T.nu); pm = c->ring.mul9(pm, mi); }
    }
    E9PreC *d_mup;
    RET(upload_consts(c, "sf_mup", mu_pre, &d_mup));
    c->sf_round = 0; c->sf_n = m; c->sf_cur = 0;
    return LF_OK;
}
int BbCtx::sumcheck_fold_round(const uint64_t *r_prev, uint64_t *evals_out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->sf_round < 0 || c->sf_round >= (int)c->P.s) return LF_ERR_STATE;
    if ((c->sf_round == 0) != (r_prev == nullptr)) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    const size_t m = c->m;
    const u32 K2 = 2 * P.K;
    const size_t T5P = 3 * TAU + 2 * RE;
    fe *T[2], *F[2];
    i64 *partial;
    u64 *od;
    E9PreC *d_mup;
    RET(c->tbuf("sf_T0", T5P * m, &T[0]));
    RET(c->tbuf("sf_T1", T5P * atl(m / 2), &T[1]));
    RET(c->tbuf("sf_F0", (size_t)K2 * TAU * RE * m, &F[0]));
    RET(c->tbuf("sf_F1", (size_t)K2 * TAU * RE * atl(m / 2), &F[1]));
    RET(c->tbuf("sf_mup", (size_t)K2 * TAU + 8, &d_mup));
    RET(c->tbuf("round_partial", fold_partial_words(m), &partial));
    RET(c->tbuf("round_out", 5 * RE, &od));
    if (r_prev) {
        E9PreC r = e9pre_from_h9(h9_load(r_prev), c->ring.T.nu);
        int src = c->sf_cur, dst = src ^ 1;
        size_t ldi = c->sf_n == m ? m : atl(c->sf_n), ldo = atl(c->sf_n / 2);
        launch_fix(c->dev, T[src], ldi, T[dst], ldo, c->sf_n, 19, r, c->stream());
        launch_fix(c->dev, F[src], ldi, F[dst], ldo, c->sf_n, K2 * TAU * 8, r, c->stream());
        c->sf_cur = dst; c->sf_n /= 2;
    }
    const size_t n = c->sf_n, ld = n == m ? m : atl(n);
    const fe *t5 = T[c->sf_cur];
    FoldArgs a;
    a.eqL = t5; a.eqR = t5 + (size_t)TAU * ld; a.eqB = t5 + (size_t)2 * TAU * ld; a.G1 = t5 + (size_t)3 * TAU * ld; a.G2 = t5 + (size_t)(3 * TAU + RE) * ld;
    a.ld = ld; a.n = n; a.p0 = 0; a.pcnt = n / 2; a.pF0 = 0;
    launch_fold_round(c->dev, a, F[c->sf_cur], ld, P.K, d_mup, partial, od, c->stream());
    c->sf_round++;
    return down_small(c, od, (size_t)(2 * P.b + 1) * RE, evals_out);
}
int BbCtx::sumcheck_fold_end() {
    std::lock_guard<std::mutex> g(p->mu);
    p->sf_round = -1;
    return LF_OK;
}
// compute_f_0 (nifs/folding.rs:258-268) with ring-element coefficients
int BbCtx::lincomb(const uint64_t *coef, const uint64_t *tables, size_t n_terms, size_t len, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *X, *o;
    RET(c->tbuf("io_a", n_terms * len * RE, &X));
    RET(c->tbuf("io_b", len * RE, &o));
    for (size_t i = 0; i < n_terms; i++) RET(up_ring(c, tables + i * len * RE, len, X + i * RE * len));
    std::vector<E9PreC> cf(n_terms * 8);
    for (size_t i = 0; i < n_terms; i++)
        for (int sl = 0; sl < 8; sl++) cf[i * 8 + sl] = e9pre_from_h9(h9_load(coef + i * RE + (size_t)TAU * sl), c->ring.T.nu);
    E9PreC *d_cf;
    RET(upload_consts(c, "lc_coef", cf, &d_cf));
    launch_lincomb_z(c->dev, X, len, (u32)n_terms, d_cf, 1, len, o, c->stream(), 1);
    return down_ring(c, o, len, out);
}
// calculate_challenged_mz_mle (nifs/folding.rs:208-226) / prepare_g1_and_3_k_mles_list (folding/utils.rs:524-546)
int BbCtx::horner_combine(const uint64_t *tables, size_t groups, size_t per_group, size_t len, const uint64_t *challenges, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    const size_t nt = groups * per_group;
    fe *X, *o;
    RET(c->tbuf("io_a", nt * len * RE, &X));
    RET(c->tbuf("io_b", len * RE, &o));
    for (size_t i = 0; i < nt; i++) RET(up_ring(c, tables + i * len * RE, len, X + i * RE * len));
    std::vector<E9PreC> cf(nt);
    for (size_t i = 0; i < groups; i++) {
        H9 ci = h9_load(challenges + (size_t)TAU * i), pw = ci;
        for (size_t j = 0; j < per_group; j++) { cf[i * per_group + j] = e9pre_from_h9(pw, c->ring.T.nu); pw = c->ring.mul9(pw, ci); }
    }
    E9PreC *d_cf;
    RET(upload_consts(c, "lc_coef", cf, &d_cf));
    launch_lincomb_z(c->dev, X, len, (u32)nt, d_cf, 1, len, o, c->stream(), 0);
    return down_ring(c, o, len, out);
}
int BbCtx::sumcheck_lin_end() {
    std::lock_guard<std::mutex> g(p->mu);
    p->sc_round = -1;
    return LF_OK;
}

}  // namespace lfbb

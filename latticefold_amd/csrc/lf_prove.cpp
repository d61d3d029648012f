// lf_prove.cpp -- the host driver that replays `NIFSProver::prove` (crates/latticefold/src/nifs.rs:48-103) on the GPU kernels: the linearization prover
// (nifs/linearization.rs:213-260), the decomposition prover (nifs/decomposition.rs:70-160) and the entry points lf_linearize / lf_fold_step /
// lf_decomposition_prove / lf_folding_prove.  The folding prover proper is lf_fold.cpp.
//
// Host <-> device traffic inside a fold step is O(proof size): per sumcheck round (D+1) ring elements come back
// and one F_{p^3} challenge goes down; everything of size N stays in HBM.
#include "lf_ctx.h"

// =================================================================================================================================
// the driver

// sumcheck transcript prologue: absorb R::from(nvars), R::from(degree)  (utils/sumcheck.rs:60-62)
void sc_prologue(Transcript &tr, u32 nv, u32 deg) {
    tr.absorb_u64_as_ring(nv);
    tr.absorb_u64_as_ring(deg);
}
Fq3 sc_round_transcript(Transcript &tr, const u64 *evals, u32 npts) {
    tr.absorb_ring(evals, npts);
    Fq3 r = tr.get_challenge();
    tr.absorb_fq3_as_ring(r);
    return r;
}

// linearization sumcheck on device tables mz [t][24][m] (left intact) and eq_beta [3][m]
// `u_dev` (optional): the Mz tables fixed at the whole point, i.e. u_j = Mz_j(r) (t ring elements, canonical) -- the last fix of the
// tables the rounds work on, so linearization.rs:136's evaluate_mles pass over the full tables is not needed.
// after_round (optional): called with the round number as soon as that round's challenge is known
// beta (optional): the point of eqb.  With it the large rounds of an unsharded run use the split form of the eq factor (k_lin_round SPLIT): the kernel sums
// E_i[p] h(X, p) at d of the d + 2 points and the host completes the message -- g_i(X) = c_i eq(beta_i, X) T_i(X), T_i(1) from g_i(0) + g_i(1) = g_{i-1}(r_{i-1}),
// the top point by extrapolation of the degree-d T_i -- in exact field arithmetic: the words of the reference's message.
static int run_lin_sumcheck(lf_ctx *c, Transcript &tr, const u64 *mz, const u64 *eqb, u64 *msgs /* s*(d+2) ring */, Fq3 *point, u64 *u_dev = nullptr,
                            const std::function<void(u32)> *after_round = nullptr, const Fq3 *beta = nullptr) {
    const lf_params &P = c->P;
    u32 deg = P.d + 1;
    size_t m = c->m;
    u64 *fx[2], *fe[2], *partial, *od;
    RET(c->tbuf("lin_fix0", (size_t)P.t * 24 * (m / 2), &fx[0]));
    RET(c->tbuf("lin_fix1", (size_t)P.t * 24 * (m / 4 ? m / 4 : 1), &fx[1]));
    RET(c->tbuf("lin_efix0", 3 * (m / 2), &fe[0]));
    RET(c->tbuf("lin_efix1", 3 * (m / 4 ? m / 4 : 1), &fe[1]));
    RET(c->tbuf("round_partial", round_partial_words(), &partial));
    od = c->round_out();
    if (!od) return LF_ERR_HIP;
    { HostTimer ht(c); sc_prologue(tr, P.s, deg); }
    const u64 *cur = mz, *cure = eqb;
    size_t n = m;
    int flip = 0;
    // Sharded rounds: rank g owns the entries [g*n/G, (g+1)*n/G) of every table (high index bits: pairs stay local), fixes
    // and evaluates only those; the (deg+1)-element partial messages are all-gathered and added mod p on the device.  Below 64 pairs per
    // rank the slices are gathered and the tail rounds are replicated.
    const size_t Gw = (size_t)c->sh_world, gr = (size_t)c->sh_rank;
    bool sharded = shard_keep(c, 0, m);
    u64 *od_dev = nullptr;
    if (Gw > 1) RET(c->tbuf("lin_round_out", 5 * 24 + 8, &od_dev));
    // split form: while `split` is set, cure is the per-pair table E_i of the round i that ran last (in fe[(i - 1) & 1]) and c_lvl = c_i = prod_{k<i} eq(beta_k, r_k)
    const u32 dT = P.d;                                          // degree of T_i; the message has degree dT + 1 = deg
    bool split = beta && Gw == 1 && !c->tn.lin_no_split && P.s >= 2 && m >= c->tn.lin_split_min && m >= 16 && dT >= 1 && deg <= 4;
    Fq3 c_lvl = fq3_one();
    auto f3zero = [](const Fq3 &x) { return !(x.c[0] | x.c[1] | x.c[2]); };
    auto eq1 = [&](const Fq3 &b, const Fq3 &r) {   // eq(beta, r) = (1 - beta)(1 - r) + beta r
        return fq3_add(c->ring.mul3(fq3_sub(fq3_one(), b), fq3_sub(fq3_one(), r)), c->ring.mul3(b, r));
    };
    if (split) RET(build_eq_dev(c, beta + 1, P.s - 1, fe[0]));   // E_1 = eq((beta_2..beta_s), .), m / 2 entries
    c->lin_split_rounds = 0;
    for (u32 round = 1; round <= P.s; round++) {
        if (split && round >= 2) {
            // stay in the split form?  Not into the persistent tail, not below the size where it pays, not when c_i or beta_i cannot be divided by
            const bool tail_next = !c->tn.no_tail && n <= c->tn.tail_n && n >= 4 && P.s - round + 1 <= TAIL_MAX_ROUNDS;
            const Fq3 c_next = c->ring.mul3(c_lvl, eq1(beta[round - 2], point[round - 2]));
            if (tail_next || n < 8 || n / 2 < c->tn.lin_split_min || f3zero(c_next) || f3zero(beta[round - 1])) {
                // back to the ordinary table of the previous round's n entries: eq(beta, (r_1..r_{i-1}, b, p)) = c_i eq(beta_i, b) E_i[p] at entry 2p + b
                u64 *ex;
                RET(c->tbuf("lin_eexp", 3 * n, &ex));
                const Fq3 bi = beta[round - 2];
                launch_eq_expand(c->dcrt, cure, n / 2, n / 2, f3c(c->ring.mul3(c_lvl, fq3_sub(fq3_one(), bi))), f3c(c->ring.mul3(c_lvl, bi)), ex, n, c->stream());
                cure = ex;
                split = false;
            } else c_lvl = c_next;
        }
        // persistent tail (k_lin_tail): all remaining rounds in one launch once the tables are small, as in the folding sumcheck
        if (!sharded && !c->tn.no_tail && round >= 2 && n <= c->tn.tail_n && n >= 4 && P.s - round + 1 <= TAIL_MAX_ROUNDS) {
            int trc = lin_tail_rounds(c, tr, cur, cure, n, fx[flip], partial, round, point, msgs, deg, after_round);
            if (trc == LF_OK) { cur = fx[flip]; n = 2; break; }
            if (trc != LF_ERR_UNSUPPORTED) return trc;
        }
        bool fused_now = false;
        const u64 *prev = cur, *preve = cure;
        size_t prevn = n;
        if (round > 1) {
            Fq3Const r = f3c(point[round - 2]);
            if (sharded) {
                const size_t e0 = gr * (n / Gw), ecnt = n / Gw;   // this rank's entries of the previous tables -> entries [e0/2, (e0+ecnt)/2)
                launch_fix_many(c->dcrt, cur + e0, n, fx[flip] + e0 / 2, n / 2, ecnt, P.t * 8, r, c->stream());
                launch_fix_many(c->dcrt, cure + e0, n, fe[flip] + e0 / 2, n / 2, ecnt, 1, r, c->stream());
            } else if (split || (Gw == 1 && n >= 8)) {
                fused_now = true;   // fix_variables inside the round kernel (one pass over the previous tables instead of a k_fix pass + a read)
            } else {
                launch_fix_many(c->dcrt, cur, n, fx[flip], n / 2, n, P.t * 8, r, c->stream());
                launch_fix_many(c->dcrt, cure, n, fe[flip], n / 2, n, 1, r, c->stream());
            }
            cur = fx[flip]; cure = split ? fe[(round - 1) & 1] : fe[flip];   // (split: E_round, one entry per pair of the new tables)
            flip ^= 1;
            n /= 2;
            if (sharded && !shard_keep(c, 0, n)) {   // hand-over to the replicated rounds: the Mz tables and eq in one exchange
                const size_t lcl = n / Gw;
                const GatherPart gp[2] = {{cur + gr * lcl, n, (u64 *)cur, (size_t)P.t * 24}, {cure + gr * lcl, n, (u64 *)cure, 3}};
                RET(gather_parts(c, gp, 2, lcl));
                sharded = false;
            }
        }
        u64 *ev = msgs + (size_t)(round - 1) * (deg + 1) * 24;
        if (sharded) {
            const size_t p0 = gr * (n / 2 / Gw), pcnt = n / 2 / Gw;
            launch_lin_round(c->dcrt, c->desc, cur + 2 * p0, n, cure + 2 * p0, n, 2 * pcnt, deg, partial, od_dev, c->stream(), c->lin_blocks);
            RET(exchange_modsum_dev(c, od_dev, (size_t)(deg + 1) * 24));
            HIPCHK(hipMemcpyAsync(od, od_dev, (size_t)(deg + 1) * 24 * 8, hipMemcpyDeviceToHost, c->stream()));
        } else if (split) {
            // the points the kernel evaluates: all of 0..dT in round 1 (no previous message to take T(1) from), 0 and 2..dT afterwards
            const u32 xmask = round == 1 ? (1u << (dT + 1)) - 1 : (((1u << (dT + 1)) - 1) & ~2u);
            if (round == 1) launch_lin_round(c->dcrt, c->desc, cur, n, fe[0], n / 2, n, deg, partial, od, c->stream(), c->lin_blocks, xmask);
            else launch_lin_round_fused(c->dcrt, c->desc, prev, prevn, preve, prevn / 2, f3c(point[round - 2]), (u64 *)cur, n, (u64 *)cure, n / 2, n, deg, partial, od, c->stream(),
                                        c->lin_blocks, xmask);
            if (round == 1) cure = fe[0];
        } else if (fused_now)
            launch_lin_round_fused(c->dcrt, c->desc, prev, prevn, preve, prevn, f3c(point[round - 2]), (u64 *)cur, n, (u64 *)cure, n, n, deg, partial, od, c->stream(), c->lin_blocks);
        else launch_lin_round(c->dcrt, c->desc, cur, n, cure, n, n, deg, partial, od, c->stream(), c->lin_blocks);
        RET(c->lane_sync());                                  // the message is in mapped host memory
        if (split) {
            // od[X][slot] = T(X) = sum_p E[p] h(X, p) at the evaluated X: complete the message g(X) = c_i eq(beta_i, X) T(X), X = 0..deg
            HostTimer ht2(c);
            c->lin_split_rounds++;
            const Fq3 bi = beta[round - 1], obi = fq3_sub(fq3_one(), bi);
            Fq3 wS[5];   // Lagrange weights of the previous message at r_{i-1} (nodes 0..deg)
            Fq3 cinv = fq3_one(), binv = fq3_one();
            if (round >= 2) {
                const Fq3 x = point[round - 2];
                for (u32 j = 0; j <= deg; j++) {
                    Fq3 num = fq3_one();
                    u64 den = 1;
                    for (u32 k = 0; k <= deg; k++) {
                        if (k == j) continue;
                        num = c->ring.mul3(num, fq3_sub(x, fq3_make(k, 0, 0)));
                        den = fq_mul(den, j > k ? (u64)(j - k) : LF_P - (u64)(k - j));
                    }
                    const u64 di = fq_inv(den);
                    wS[j] = fq3_make(fq_mul(num.c[0], di), fq_mul(num.c[1], di), fq_mul(num.c[2], di));
                }
                cinv = c->ring.inv3(c_lvl);
                binv = c->ring.inv3(bi);
            }
            const u64 *prev_ev = round >= 2 ? msgs + (size_t)(round - 2) * (deg + 1) * 24 : nullptr;
            static const int binom[5][6] = {{1}, {1, 1}, {1, 2, 1}, {1, 3, 3, 1}, {1, 4, 6, 4, 1}};
            for (u32 slot = 0; slot < 8; slot++) {
                Fq3 T[5];
                for (u32 X = 0; X <= dT; X++) T[X] = fq3_make(od[X * 24 + 3 * slot], od[X * 24 + 3 * slot + 1], od[X * 24 + 3 * slot + 2]);
                if (round >= 2) {
                    Fq3 S = fq3_zero();
                    for (u32 j = 0; j <= deg; j++)
                        S = fq3_add(S, c->ring.mul3(wS[j], fq3_make(prev_ev[j * 24 + 3 * slot], prev_ev[j * 24 + 3 * slot + 1], prev_ev[j * 24 + 3 * slot + 2])));
                    // c (l(0) T(0) + l(1) T(1)) = S,  l(0) = 1 - beta_i, l(1) = beta_i
                    T[1] = c->ring.mul3(fq3_sub(c->ring.mul3(S, cinv), c->ring.mul3(obi, T[0])), binv);
                }
                // T has degree dT: its value at dT + 1 from the dT + 1 below (the (dT+1)-th finite difference vanishes)
                Fq3 top = fq3_zero();
                for (u32 j = 0; j <= dT; j++) {
                    Fq3 term = T[j];
                    Fq3 acc = fq3_zero();
                    for (int q = 0; q < binom[dT + 1][j]; q++) acc = fq3_add(acc, term);
                    top = ((dT - j) & 1) ? fq3_sub(top, acc) : fq3_add(top, acc);
                }
                T[dT + 1] = top;
                Fq3 l = obi;   // eq(beta_i, X) = (1 - beta_i) + X (2 beta_i - 1)
                const Fq3 dl = fq3_sub(bi, obi);
                for (u32 X = 0; X <= deg; X++) {
                    const Fq3 g = c->ring.mul3(c->ring.mul3(c_lvl, l), T[X]);
                    ev[X * 24 + 3 * slot] = g.c[0]; ev[X * 24 + 3 * slot + 1] = g.c[1]; ev[X * 24 + 3 * slot + 2] = g.c[2];
                    l = fq3_add(l, dl);
                }
            }
        } else
        memcpy(ev, od, (size_t)(deg + 1) * 24 * 8);
        HostTimer ht(c);
        point[round - 1] = sc_round_transcript(tr, ev, deg + 1);
        if (after_round) (*after_round)(round);
        if (round == 1) TL_MARK("  lin round 1");
        if (round == 2) TL_MARK("  lin round 2");
        if (round == 4) TL_MARK("  lin round 4");
        if (round == 8) TL_MARK("  lin round 8");
    }
    TL_MARK("  lin rounds done");
    if (u_dev) launch_fix_final(c->dcrt, cur, P.t * 8, f3c(point[P.s - 1]), u_dev, c->stream());   // n == 2 here (ld 2)
    return LF_OK;
}

// z = head (x.. , h) || w where w comes from the planes; K = 1 & mode 0 for the full witness
// Columns [*lo, *hi) of z that rows [r0, r0 + rcnt) of the t constraint matrices refer to -- from the device CSR, once per (CCS, slice).  A sharded rank
// needs the z-space combinations (sum_k zeta_k z_k) only there: for column-local systems (R1CS rows over their own variables, the bench's identity /
// diagonal matrices) that is its own n / G columns, for an arbitrary CCS the whole range -- never more work than the replicated step did.
int shard_col_range(lf_ctx *c, size_t r0, size_t rcnt, size_t *lo, size_t *hi) {
    static std::mutex mu;   // (the two lanes of a step may ask at the same time)
    std::lock_guard<std::mutex> g(mu);
    if (c->shc_r0 != r0 || c->shc_rcnt != rcnt) {
        size_t mn = c->n, mx = 0;
        std::vector<u32> rp(2), cl;
        for (u32 j = 0; j < c->P.t; j++) {
            HIPCHK(hipMemcpy(&rp[0], c->d_rowptr[j] + r0, 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&rp[1], c->d_rowptr[j] + r0 + rcnt, 4, hipMemcpyDeviceToHost));
            if (rp[1] <= rp[0]) continue;
            cl.resize(rp[1] - rp[0]);
            HIPCHK(hipMemcpy(cl.data(), c->d_col[j] + rp[0], cl.size() * 4, hipMemcpyDeviceToHost));
            for (u32 v : cl) { if (v < mn) mn = v; if ((size_t)v + 1 > mx) mx = (size_t)v + 1; }
        }
        if (mx <= mn) { mn = 0; mx = 0; }
        c->shc_r0 = r0; c->shc_rcnt = rcnt; c->shc_lo = mn; c->shc_hi = mx;
    }
    *lo = c->shc_lo; *hi = c->shc_hi;
    return LF_OK;
}
// w0 / wcnt (optional): only the witness columns [w0, w0 + wcnt) -- z columns l + 1 + w0 .. -- are built (a sharded rank's slice; the heads always)
static int build_z(lf_ctx *c, const int32_t *planes, u32 K, int mode_bits, const u64 *heads /* K*(l+1) ring AoS host */, u64 *z /* [K][24][n] */,
                   size_t w0 = 0, size_t wcnt = (size_t)-1) {
    const lf_params &P = c->P;
    u32 hl = P.l + 1;
    if (wcnt == (size_t)-1) { w0 = 0; wcnt = P.wit_len; }
    if (wcnt) launch_recompose_crt(c->dcrt, planes + w0 * P.L, c->N, (u32)wcnt, P.L, P.B, K, mode_bits, z, c->n, hl + w0, c->stream());
    // heads: write plane entries 0..l of each table
    std::vector<u64> h((size_t)K * 24 * hl);
    for (u32 k = 0; k < K; k++)
        for (u32 i = 0; i < hl; i++)
            for (int w = 0; w < 24; w++) h[((size_t)k * 24 + w) * hl + i] = heads[((size_t)k * hl + i) * 24 + w];
    u64 *stage;
    RET(c->tbuf("z_heads", h.size(), &stage));
    RET(c->h2d_small(stage, h.data(), h.size() * 8));   // pinned ring: no synchronisation for the stack buffer
    HIPCHK(hipMemcpy2DAsync(z, c->n * 8, stage, hl * 8, hl * 8, (size_t)K * 24, hipMemcpyDeviceToDevice, c->stream()));
    return LF_OK;
}

struct LinOut {
    std::vector<Fq3> r;  // point
};

static bool lcccs_point(const lf_params &P, const u64 *lcccs, std::vector<Fq3> &pt) {
    pt.resize(P.s);
    for (u32 i = 0; i < P.s; i++)
        if (!HostRing::is_diag(lcccs + (size_t)i * 24, &pt[i])) return false;
    return true;
}

static int linearize_impl(lf_ctx *c, Transcript &tr, const u64 *cccs, const lf_witness *wit, u64 *lcccs_out, u64 *proof, u64 **eq_r_keep) {
    const lf_params &P = c->P;
    size_t m = c->m, n = c->n;
    size_t ph = c->ev_begin(10);
    // z = x_ccs || 1 || w_ccs (arith.rs:399-409)
    std::vector<u64> head((size_t)(P.l + 1) * 24);
    memcpy(head.data(), cccs + (size_t)P.kappa * 24, (size_t)P.l * 24 * 8);
    HostRing::from_u64(1, head.data() + (size_t)P.l * 24);
    u64 *z, *mz, *eqb, *eqr, *partial, *od;
    RET(c->tbuf("lin_z", 24 * n, &z));
    RET(c->tbuf("lin_mz", (size_t)P.t * 24 * m, &mz));
    RET(c->tbuf("lin_eqb", 3 * m, &eqb));
    RET(c->tbuf("eq_r_R", 3 * m, &eqr));
    RET(c->tbuf("red_partial", 256 * 4096, &partial));
    RET(c->tbuf("lin_small", 4096, &od));
    RET(build_z(c, wit->planes, 1, 0, head.data(), z));
    TL_MARK("  lin z enqueued");
    std::vector<Fq3> beta(P.s);
    {
        HostTimer ht(c);
        tr.absorb_label("beta_s");
        for (u32 i = 0; i < P.s; i++) beta[i] = tr.get_challenge();
    }
    RET(build_eq_dev(c, beta.data(), P.s, eqb));
    {
        // a sharded rank evaluates and fixes the entries [rank m/G, (rank+1) m/G) of the Mz tables until the fixed slices are gathered (run_lin_sumcheck):
        // it computes only those rows (z itself stays whole: a row refers to arbitrary columns)
        const size_t Gw = (size_t)c->sh_world;
        const bool rows_sliced = shard_keep(c, 0, m) && !c->tn.lin_u_eval;
        const size_t r0 = rows_sliced ? (size_t)c->sh_rank * (m / Gw) : 0, rcnt = rows_sliced ? m / Gw : m;
        if (c->ccs_general) {      // general matrices: whole-element gathers from one element-major copy of z
            u64 *zaos;
            RET(c->tbuf("spmv_zaos", (size_t)P.t * n * 24, &zaos));
            launch_soa_to_aos(z, zaos, n, c->stream());
            for (u32 j = 0; j < P.t; j++)
                launch_spmv_rows(c->dcrt, 1, &c->d_rowptr[j], &c->d_col[j], &c->d_val[j], nullptr, 0, n, zaos, mz + (size_t)j * 24 * m, m, 0, c->stream(), r0, rcnt);
        } else
        for (u32 j = 0; j < P.t; j++)
            launch_spmv(c->dcrt, c->d_rowptr[j], c->d_col[j], c->d_val[j], z, n, mz + (size_t)j * 24 * m, m, 0, c->stream(), r0, rcnt);
    }
    std::vector<Fq3> pt(P.s);
    // v, u at the sumcheck point (linearization.rs:126-139): u from the fully fixed Mz tables of the sumcheck (LF_LIN_U_EVAL=1: dot
    // products with eq(r) over the full tables), v from the witness planes
    const bool u_eval = c->tn.lin_u_eval;
    TL_MARK("  lin Mz enqueued");
    // The K digit-plane evaluations v_s at the sumcheck point r are the longest piece between the last round and the absorb of v (0.5 of 0.7-1.0 ms at
    // 2^20 rows, on the critical path of both lanes).  eq(r, i) = eq((r_1..r_J), i mod 2^J) * eq((r_J+1..r_s), i >> J): the pass over the witness only needs
    // the first J coordinates, so it starts on a side stream as soon as round J's challenge is there (launch_sv_vs_blocks: one partial sum per block of 2^J
    // positions) and runs under the last s - J rounds; afterwards 2^(s-J) weighted partial sums remain (launch_sv_vs_combine).
    struct VsSplit {
        bool armed = false, launched = false, failed = false;
        u32 J = 0, nblocks = 0;
        int sd = 0;
        unsigned char *EB = nullptr;
        int32_t *part = nullptr;
        u64 *eqlo = nullptr, *scr = nullptr, *wts = nullptr;
        Fq3Const *rd = nullptr;
    } vsp;
    if (P.b == 2 && c->sh_world == 1 && !c->tn.force_exchange && P.s >= 12 &&
        P.K <= 16) {
        const u32 back = 6;                                  // rounds before the last one after which the pass starts (2^6 blocks of partial sums)
        const u32 J = P.s < back + 10 ? 10 : P.s - back;
        const size_t bs = (size_t)1 << J;
        for (int sd = 0; sd < 2; sd++)
            if (c->bits_wit[sd] == wit && c->bits_ptr[sd] && J < P.s && c->N % bs == 0 && c->N / bs <= sv_vs_max_blocks(P.K) && c->N / bs >= 1) {
                vsp.J = J; vsp.nblocks = (u32)(c->N / bs); vsp.sd = sd;
                bool ok = c->tbuf("vs_eb", sv_eb_bytes(c->N / 2), &vsp.EB) == LF_OK && c->tbuf("vs_part_blocks", sv_vs_blocks_part_words(vsp.nblocks, P.K), &vsp.part) == LF_OK &&
                          c->tbuf("vs_eqlo", 3 * bs, &vsp.eqlo) == LF_OK && c->tbuf("vs_eq_scratch", build_eq_scratch_words(J), &vsp.scr) == LF_OK &&
                          c->tbuf("vs_wts", (size_t)3 * vsp.nblocks + 8, &vsp.wts) == LF_OK && c->tbuf("vs_eq_point", 64, &vsp.rd) == LF_OK;
                if (ok && !c->st_aux) ok = hipStreamCreateWithFlags(&c->st_aux, hipStreamNonBlocking) == hipSuccess;
                if (ok && !c->ev_aux) ok = hipEventCreateWithFlags(&c->ev_aux, hipEventDisableTiming) == hipSuccess;
                if (ok && !c->h_aux) ok = hipHostMalloc((void **)&c->h_aux, 1024, hipHostMallocDefault) == hipSuccess;
                vsp.armed = ok;
                break;
            }
    }
    const std::function<void(u32)> vs_hook = [&](u32 round) {
        if (!vsp.armed || round != vsp.J) return;
        Fq3Const *h = (Fq3Const *)c->h_aux;
        for (u32 i = 0; i < vsp.J; i++) h[i] = f3c(pt[i]);
        bool ok = hipMemcpyAsync(vsp.rd, h, vsp.J * sizeof(Fq3Const), hipMemcpyHostToDevice, c->st_aux) == hipSuccess;
        if (ok) {
            launch_build_eq2(c->dcrt, vsp.rd, vsp.J, vsp.scr, vsp.eqlo, c->st_aux);
            ok = hipStreamWaitEvent(c->st_aux, c->bits_ev[vsp.sd], 0) == hipSuccess &&
                 launch_sv_vs_blocks(c->bits_ptr[vsp.sd], c->N, vsp.eqlo, (size_t)1 << vsp.J, vsp.J, P.K, vsp.EB, vsp.part, c->st_aux) == 0 &&
                 hipEventRecord(c->ev_aux, c->st_aux) == hipSuccess;
        }
        vsp.launched = ok;
        vsp.failed = !ok;
    };
    RET(run_lin_sumcheck(c, tr, mz, eqb, proof, pt.data(), u_eval ? nullptr : od + 72, vsp.armed ? &vs_hook : nullptr, beta.data()));
    if (vsp.failed) { (void)hipStreamSynchronize(c->st_aux); return LF_ERR_HIP; }
    if (!vsp.launched) RET(build_eq_dev(c, pt.data(), P.s, eqr));
    u64 *v = proof + (size_t)P.s * (P.d + 2) * 24, *u = v + 3 * 24;   // contiguous: v[3 ring] u[t ring]
    {   // T[24][3] flat == v[3][8 slots][3]; sharded: each rank sums its index slice, partial sums exchanged on the device
        size_t i0, cnt;
        shard_slice(c, c->N, &i0, &cnt);
        c->vs_wit = nullptr;
        if (P.b == 2 && c->sh_world == 1 && !c->tn.force_exchange) {
            // the K digit-plane evaluations v_s[k] (needed by the decomposition of this instance at the same point anyway) instead of the
            // evaluation of the full coefficients: v = sum_k 2^k v_s[k]
            u64 *vs;
            RET(c->tbuf("lin_vs", (size_t)P.K * 72 + 8, &vs));
            if (vsp.launched) {
                // w_b = eq((r_J+1..r_s), b): index bit j of the block number belongs to coordinate J + 1 + j
                std::vector<u64> w((size_t)3 * vsp.nblocks);
                for (u32 b = 0; b < vsp.nblocks; b++) {
                    Fq3 acc = fq3_one();
                    for (u32 j = 0; vsp.J + j < P.s; j++) acc = c->ring.mul3(acc, ((b >> j) & 1) ? pt[vsp.J + j] : fq3_sub(fq3_one(), pt[vsp.J + j]));
                    w[3 * b] = acc.c[0]; w[3 * b + 1] = acc.c[1]; w[3 * b + 2] = acc.c[2];
                }
                RET(c->h2d_small(vsp.wts, w.data(), w.size() * 8));
                HIPCHK(hipStreamWaitEvent(c->stream(), c->ev_aux, 0));
                launch_sv_vs_combine(c->dcrt, vsp.part, vsp.nblocks, vsp.wts, P.K, vs, c->stream());
            } else
                RET(coef_eval_dev(c, wit->planes, c->N, eqr, m, P.K, 1, partial, vs, c->N, wit));
            launch_vs_combine(vs, P.K, od, c->stream());
            if (c->vs_keep) { c->vs_wit = wit; c->vs_eq = eqr; c->vs_dev = vs; }
        } else {
            RET(coef_eval_dev(c, wit->planes + i0, cnt, eqr + i0, m, 1, 0, partial, od, c->N));
            RET(exchange_modsum_dev(c, od, 72));
        }
    }
    if (u_eval) {
        RET(down_small(c, od, 72, v));
        launch_dot_eq(c->dcrt, mz, m, P.t, eqr, m, m, partial, od, c->stream());
        RET(down_small(c, od, (size_t)P.t * 24, u));
    } else RET(down_small(c, od, 72 + (size_t)P.t * 24, v));
    if (vsp.launched) RET(build_eq_dev(c, pt.data(), P.s, eqr));   // (the evaluations at r that follow need it; v did not)
    {
        HostTimer ht(c);
        tr.absorb_ring(v, 3);
        tr.absorb_ring(u, P.t);
    }
    u64 *o = lcccs_out;
    for (u32 i = 0; i < P.s; i++, o += 24) HostRing::from_fq3(pt[i], o);
    memcpy(o, v, 72 * 8); o += 72;
    memcpy(o, cccs, (size_t)P.kappa * 24 * 8); o += (size_t)P.kappa * 24;
    memcpy(o, u, (size_t)P.t * 24 * 8); o += (size_t)P.t * 24;
    memcpy(o, cccs + (size_t)P.kappa * 24, (size_t)P.l * 24 * 8); o += (size_t)P.l * 24;
    HostRing::from_u64(1, o);
    if (eq_r_keep) *eq_r_keep = eqr;
    c->ev_end(ph);
    return LF_OK;
}

// decompose_big_vec_into_k_vec_and_compose_back (nifs/decomposition/utils.rs:12-42) on l+1 elements, host
static void compute_x_s(const lf_ctx *c, const u64 *xh /* (l+1) NTT */, u64 *x_s /* K*(l+1) NTT */) {
    const lf_params &P = c->P;
    u32 cnt = P.l + 1;
    std::vector<u64> co(24);
    for (u32 i = 0; i < cnt; i++) {
        c->ring.icrt(xh + (size_t)i * 24, co.data());
        // per coefficient: L digits base B, each K digits base b
        std::vector<int64_t> dB(P.L), dk(P.K);
        std::vector<std::vector<u64>> part(P.K, std::vector<u64>(24, 0));
        for (int cc = 0; cc < 24; cc++) {
            balanced_digits(co[cc], P.B, P.L, dB.data(), c->digit_mode);
            u64 pw = 1;
            for (u32 l = 0; l < P.L; l++) {
                balanced_digits(fq_from_i64(dB[l]), P.b, P.K, dk.data(), c->digit_mode);
                for (u32 k = 0; k < P.K; k++) {
                    u64 term = fq_mul(pw, fq_from_i64(dk[k]));
                    part[k][cc] = fq_add(part[k][cc], term);
                }
                pw = fq_mul(pw, P.B % LF_P);
            }
        }
        for (u32 k = 0; k < P.K; k++) c->ring.crt(part[k].data(), x_s + ((size_t)k * cnt + i) * 24);
    }
}


// LFDecompositionProver::prove (nifs/decomposition.rs:33-88)
// commit_witnesses (decomposition.rs:178-201): NTT of the K-1 upper bit-planes, one batched pass over A, then
// y_0 = cm - sum_{k>=1} b^k y_k on the host.  Depends only on the witness and on cm -- not on the evaluation point.
// `enqueue_only`: leave the result in flight on the lane's stream (finished later by decompose_commit_finish).
static int decompose_commit_enqueue(lf_ctx *c, const lf_witness *wit, u64 **yd_out, size_t *ev_out, const char *ybuf = "dec_y") {
    const lf_params &P = c->P;
    size_t N = c->N;
    u32 K = P.K;
    u64 *yd;
    RET(c->tbuf(ybuf, (size_t)K * P.kappa * 24, &yd));
    size_t ph = c->ev_begin(11);
    if (!c->i8_nch) return LF_ERR_STATE;
    // int8 matrix cores: digits straight from the coefficient planes, no bit-plane NTTs (this rank's column slice when sharded)
    RET(commit_planes_i8(c, wit->planes + c->A_col0, N, 1, K - 1, yd, wit));
    *yd_out = yd;
    *ev_out = ph;
    return LF_OK;
}
// early / early_ev (optional): the commitments were already copied to this pinned buffer behind the commit (event early_ev)
static int decompose_commit_finish(lf_ctx *c, const u64 *cm, u64 *yd, size_t ev, u64 *proof, const u64 *early = nullptr, hipEvent_t early_ev = nullptr) {
    const lf_params &P = c->P;
    u32 K = P.K;
    u64 *y_s = proof + (size_t)K * P.t * 24 + (size_t)K * 72 + (size_t)K * (P.l + 1) * 24;
    if (early && early_ev) {
        HIPCHK(hipEventSynchronize(early_ev));
        memcpy(y_s + (size_t)P.kappa * 24, early, (size_t)(K - 1) * P.kappa * 24 * 8);
    } else RET(commit_download(c, yd, (size_t)(K - 1) * P.kappa * 24, y_s + (size_t)P.kappa * 24));
    c->ev_end(ev);
    // y_0 = cm - sum_{k>=1} b^k y_k, as the reference's fold (acc + y_i) * b
    // (b is a base-field constant: in the NTT form the product with it is the word-wise one -- 24 multiplications per element instead of eight F_{p^3} products)
    std::vector<u64> acc((size_t)P.kappa * 24, 0);
    const u64 bq = (u64)P.b % LF_P;
    for (int k = (int)K - 1; k >= 1; k--)
        for (u32 i = 0; i < P.kappa; i++) {
            u64 *a = &acc[(size_t)i * 24];
            const u64 *y = y_s + ((size_t)k * P.kappa + i) * 24;
            for (int w = 0; w < 24; w++) a[w] = fq_mul(fq_add(a[w], y[w]), bq);
        }
    for (u32 i = 0; i < P.kappa; i++) HostRing::sub(cm + (size_t)i * 24, &acc[(size_t)i * 24], y_s + (size_t)i * 24);
    return LF_OK;
}

// v / v_s / theta: sum_j eq[j] * (digit planes or coefficients of the witness planes) -> od (device, canonical).  On the int8 matrix cores
// (launch_coef_eval_i8) unless LF_COEF_VALU is set or the shape is not handled there.
int coef_eval_dev(lf_ctx *c, const int32_t *planes, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits, u64 *partial, u64 *od, size_t ldp,
                         const lf_witness *wit) {
    // the whole witness of a running fold step, digit planes: from its bit-plane form with the round-1 GEMM of lf_sv_rounds.hip (the digit
    // cutting of k_coef_eval_i8 from the int32 planes is what that kernel spends its time on)
    if (wit && mode_bits && planes == wit->planes && n == c->N)
        for (int sd = 0; sd < 2; sd++)
            if (c->bits_wit[sd] == wit && c->bits_ptr[sd]) {
                unsigned char *EB;
                int32_t *part, *tot;
                RET(c->tbuf("vs_eb", sv_eb_bytes(n / 2), &EB));
                RET(c->tbuf("vs_part", sv_vs_part_words(n, K), &part));
                RET(c->tbuf("vs_tot", sv_vs_tot_words(K), &tot));
                if (c->stream() != c->st_lane[1]) HIPCHK(hipStreamWaitEvent(c->stream(), c->bits_ev[sd], 0));
                if (launch_sv_vs(c->bits_ptr[sd], n, eq, ldeq, K, EB, part, tot, od, c->stream()) == 0) return LF_OK;
                break;
            }
    if (n >= 64) {
        unsigned char *EB;
        int32_t *part;
        long long *sum;
        const u32 nwg = 512;
        RET(c->tbuf("ce_eb", coef_eval_i8_eb_bytes(n), &EB));
        RET(c->tbuf("ce_part", coef_eval_i8_part_words(nwg), &part));
        RET(c->tbuf("ce_sum", (size_t)24 * 2 * 256, &sum));
        if (launch_coef_eval_i8(planes, ldp ? ldp : n, n, eq, ldeq, K, mode_bits, c->P.B / 2, EB, nwg, part, sum, od, c->stream()) == 0) return LF_OK;
    }
    launch_coef_eval(c->dcrt, planes, n, eq, ldeq, K, mode_bits, partial, od, c->stream(), ldp);
    return LF_OK;
}

// <X_a, Y_b> for na vectors X and nb vectors Y of n columns -> od (device, canonical): on the int8 matrix cores (lf_dot_i8.hip) unless
// LF_DOT_VALU is set or the shape is not handled there
// (st / tag: a second call in flight on another stream uses its own scratch buffers)
// yb_pre (optional): the Y digits, already packed by launch_dot_pack_y for X vectors of this alignment (the eta products of the two sides share them)
int dot_batch_dev(lf_ctx *c, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n, u64 *dpart, u64 *od, hipStream_t st,
                         const char *tag, unsigned char *yb_pre) {
    if (!st) st = c->stream();
    if (!c->tn.dot_valu && n >= c->tn.dot_min && nb <= 4) {
        unsigned char *yb;
        int32_t *part;
        long long *tot;
        if (yb_pre && nb <= 3) yb = yb_pre;
        else
        RET(c->tbuf(std::string("dot_yb") + tag, dot_i8_yb_bytes(n + 1), &yb));            // (+1: an odd column slice starts one column early)
        RET(c->tbuf(std::string("dot_i8_part") + tag, dot_i8_part_words(n + 1), &part));
        RET(c->tbuf(std::string("dot_i8_tot") + tag, dot_i8_tot_words(), &tot));
        bool ok = true;
        // a launch takes at most three vectors Y (their 72 digit rows fill its column tiles): the four matrices of a degree-three CCS (arith/ccs.rs:14-43) go in two
        // groups of two (the 64-bit VALU kernel this shape used to fall back to took 4 x 1.03 ms of a 19.3 ms C4 step)
        const u32 gsz = nb <= 3 ? nb : 2;
        for (u32 b0 = 0; b0 < nb && ok; b0 += gsz)
            for (u32 a0 = 0; a0 < na && ok; a0 += 16)
                ok = launch_dot_batch_i8(c->dcrt, X + (size_t)a0 * 24 * ldx, ldx, na - a0 < 16 ? na - a0 : 16, Y + (size_t)b0 * 24 * ldy, ldy, nb - b0 < gsz ? nb - b0 : gsz, n, yb, part, tot,
                                         od + (size_t)a0 * nb * 24, st, yb_pre != nullptr && nb <= 3, nb, b0) == 0;
        if (ok) return LF_OK;
    }
    launch_dot_batch(c->dcrt, X, ldx, na, Y, ldy, nb, n, dpart, od, st);
    return LF_OK;
}
// the point-dependent half of LFDecompositionProver::prove (decomposition.rs:33-88): x_s, v_s, z_k, u_s
// The part of a decomposition's evaluations that does not depend on the evaluation point: x_s (host, into the proof) and the K vectors
// z_k = x_s[k] || w_k on the device.  Runs on the calling lane; another lane's consumer waits for S.z_ev on its own stream.
static int decompose_prepare_z(lf_ctx *c, const u64 *xh /* (l+1) elements: x_w || h */, const lf_witness *wit, const char *side, SideState &S, u64 *proof) {
    const lf_params &P = c->P;
    u32 K = P.K;
    u64 *x_s = proof + (size_t)K * P.t * 24 + (size_t)K * 72, *z;
    int rc = c->tbuf("z_" + std::string(side), (size_t)K * 24 * c->n, &z);
    if (rc == LF_OK) {
        compute_x_s(c, xh, x_s);
        // a sharded rank reads z_k in its column slice (the u_s / eta inner products) and in the columns its rows of G refer to (the z-space
        // combination of fold prepare): it builds the range that covers both -- its own n / G columns for a column-local constraint system
        size_t w0 = 0, wcnt = (size_t)-1;
        const size_t Gw = (size_t)c->sh_world, hl = P.l + 1;
        if (shard_keep(c, 1, c->m)) {
            size_t c0, ccnt, lo, hi;
            shard_slice(c, c->n, &c0, &ccnt);
            rc = shard_col_range(c, (size_t)c->sh_rank * (c->m / Gw), c->m / Gw, &lo, &hi);
            if (hi <= lo) { lo = c0; hi = c0 + ccnt; }
            if (c0 < lo) lo = c0;
            if (c0 + ccnt > hi) hi = c0 + ccnt;
            w0 = lo > hl ? lo - hl : 0;
            const size_t w1 = hi > hl ? hi - hl : 0;
            wcnt = w1 > w0 ? w1 - w0 : 0;
            if (w0 + wcnt > P.wit_len) wcnt = P.wit_len > w0 ? P.wit_len - w0 : 0;
        }
        if (rc == LF_OK) rc = build_z(c, wit->planes, K, 1, x_s, z, w0, wcnt);
    }
    if (rc == LF_OK && !S.z_ev && hipEventCreateWithFlags(&S.z_ev, hipEventDisableTiming) != hipSuccess) rc = LF_ERR_HIP;
    if (rc == LF_OK && hipEventRecord(S.z_ev, c->stream()) != hipSuccess) rc = LF_ERR_HIP;
    S.z = z;
    S.z_state.store(rc == LF_OK ? 1 : -1, std::memory_order_release);
    return rc;
}
static int decompose_evals(lf_ctx *c, const u64 *lcccs, const std::vector<Fq3> &rpt, const lf_witness *wit, const char *side,
                           u64 *eq_r /* built already or nullptr */, SideState &S, u64 *proof) {
    const lf_params &P = c->P;
    size_t m = c->m, n = c->n, N = c->N;
    u32 K = P.K;
    std::string sd(side);
    const u64 *xh = lcccs + ((size_t)P.s + 3 + P.kappa + P.t) * 24;
    u64 *u_s = proof, *v_s = u_s + (size_t)K * P.t * 24;
    u64 *partial, *od, *q;
    RET(c->tbuf("red_partial", 256 * 4096, &partial));
    if (K > 32 || P.t > 4) return LF_ERR_UNSUPPORTED;       // the fixed layout of dec_small below (v_s: 32 x 72 words, u_s: 32 x 4 elements); lf_ccs_load enforces the same envelope
    RET(c->tbuf("dec_small", 32 * 72 + 32 * 4 * 24 + 64, &od));
    RET(c->tbuf("dec_q", (size_t)P.t * 24 * n, &q));
    u64 *od_v = od, *od_u = od + 32 * 72;
    if (!eq_r) {
        RET(c->tbuf("eq_r_" + sd, 3 * m, &eq_r));
        RET(build_eq_dev(c, rpt.data(), P.s, eq_r));
    }
    S.planes = wit->planes; S.eq_r = eq_r;
    size_t ph = c->ev_begin(12);
    // z_k: built here unless the other lane has published it already (it does not depend on the point)
    if (S.z_state.load(std::memory_order_acquire) == 1) HIPCHK(hipStreamWaitEvent(c->stream(), S.z_ev, 0));
    else if (S.z_state.load(std::memory_order_acquire) != 2) RET(decompose_prepare_z(c, xh, wit, side, S, proof));
    u64 *z = S.z;
    if (t_lane == 0) TL_MARK("  evals: buffers + z");
    // v_s (decomposition.rs:204-211) from the coefficient planes
    {
        size_t i0, cnt;
        shard_slice(c, N, &i0, &cnt);   // sharded: this rank's index slice; partial sums exchanged on the device
        if (c->vs_wit == wit && c->vs_eq == eq_r && c->sh_world == 1) {   // computed by the linearization of this step at this very point
            HIPCHK(hipMemcpyAsync(od_v, c->vs_dev, (size_t)K * 72 * 8, hipMemcpyDeviceToDevice, c->stream()));
            c->vs_wit = nullptr;
        } else {
            if (c->sh_world > 1) HIPCHK(hipMemsetAsync(od, 0, (32 * 72 + 32 * 4 * 24) * 8, c->stream()));   // (one exchange carries v_s and u_s: the gaps of the buffer must be canonical)
            RET(coef_eval_dev(c, wit->planes + i0, cnt, eq_r + i0, m, K, 1, partial, od_v, N, c->sh_world == 1 ? wit : nullptr));
        }
    }
    if (t_lane == 0) TL_MARK("  evals: v_s enqueued");
    // u_s[k][j] = <z_k, M_j^T eq(r)>   (decomposition.rs:214-256 restructured)
    u64 *dpart;
    RET(c->tbuf("dot_partial", dot_partial_words(K, P.t), &dpart));
    {
        size_t c0, cnt;
        shard_slice(c, n, &c0, &cnt);   // sharded: dot products over this rank's column slice of z_k and q_j -- and only that slice of q_j is computed
        for (u32 j = 0; j < P.t; j++)
            launch_spmv_t_eq(c->dcrt, c->d_colptr[j], c->d_rowidx[j], c->d_valT[j], eq_r, m, q + (size_t)j * 24 * n, n, c->stream(), c0, cnt);
        RET(dot_batch_dev(c, z + c0, n, K, q + c0, n, P.t, cnt, dpart, od_u));
        RET(exchange_modsum_dev(c, od, (size_t)32 * 72 + (size_t)K * P.t * 24));   // sharded: the partial v_s and u_s of this rank's slices, ONE all-gather + modular sum
    }
    // one download (one stream synchronisation) for both result sets
    {
        const size_t words = (size_t)32 * 72 + (size_t)K * P.t * 24;
        RET(c->pin(words));
        HIPCHK(hipMemcpyAsync(c->h_pin_ref(), od, words * 8, hipMemcpyDeviceToHost, c->stream()));
        RET(c->lane_sync());
        memcpy(v_s, c->h_pin_ref(), (size_t)K * 72 * 8);
        memcpy(u_s, c->h_pin_ref() + 32 * 72, (size_t)K * P.t * 24 * 8);
    }
    LF_TRACE(c, "decompose evals");
    c->ev_end(ph);
    return LF_OK;
}

// transcript part of the decomposition (decomposition.rs:65-83): absorb x_k, y_k, u_k, v_k and build the K LCCCS.
// No challenge is drawn here, so for the left instance it runs on a host thread while the GPU decomposes the right one.
static double absorb_decomposition(const lf_params &P, Transcript &tr, const u64 *lcccs, const u64 *proof, SideState &S, u32 k0 = 0, u32 k1 = ~0u) {
    auto t0 = std::chrono::steady_clock::now();
    u32 K = P.K;
    const u64 *u_s = proof, *v_s = u_s + (size_t)K * P.t * 24, *x_s = v_s + (size_t)K * 72, *y_s = x_s + (size_t)K * (P.l + 1) * 24;
    size_t ll = lf_lcccs_len(&P);
    if (k1 > K) k1 = K;
    if (k0 == 0) S.lcccs.assign((size_t)K * ll * 24, 0);
    for (u32 k = k0; k < k1; k++) {
        const u64 *xk = x_s + (size_t)k * (P.l + 1) * 24, *yk = y_s + (size_t)k * P.kappa * 24;
        const u64 *uk = u_s + (size_t)k * P.t * 24, *vk = v_s + (size_t)k * 72;
        tr.absorb_ring(xk, P.l + 1);
        tr.absorb_ring(yk, P.kappa);
        tr.absorb_ring(uk, P.t);
        tr.absorb_ring(vk, 3);
        u64 *o = &S.lcccs[(size_t)k * ll * 24];
        memcpy(o, lcccs, (size_t)P.s * 24 * 8); o += (size_t)P.s * 24;
        memcpy(o, vk, 72 * 8); o += 72;
        memcpy(o, yk, (size_t)P.kappa * 24 * 8); o += (size_t)P.kappa * 24;
        memcpy(o, uk, (size_t)P.t * 24 * 8); o += (size_t)P.t * 24;
        memcpy(o, xk, (size_t)(P.l + 1) * 24 * 8);
    }
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int lf_linearize(lf_ctx *c, lf_transcript *t, const uint64_t *cccs, const lf_witness *wit, uint64_t *lcccs_out, uint64_t *lin_proof_out) {
    if (LF_XB(c) && t && cccs && lcccs_out && lin_proof_out && c->have_ccs_any()) {
        XB x(c);
        const lf_params &P = c->params_any();
        const int ring = lf_ctx_ring(c);
        x.transcript(t);
        int rc = lf_linearize(c, t, x.ring_in(cccs, lf_cccs_len_ring(&P, ring)), wit, lcccs_out, lin_proof_out);
        if (rc == LF_OK) { x.ring_out(lcccs_out, lf_lcccs_len_ring(&P, ring)); x.ring_out(lin_proof_out, (size_t)P.s * (P.d + 2) + x.TAU + P.t); }
        return rc;
    }
    if (!c || !t || !cccs || !wit || !lcccs_out || !lin_proof_out || wit->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return t->bb ? c->bb->linearize(*t->bb, cccs, wit, lcccs_out, lin_proof_out) : LF_ERR_INVALID;
    if (t->bb) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    if (wit->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    c->tn = Tunables::read((size_t)1 << 14);
    c->ev_reset();
    c->host_tr_ms = 0;
    int rc = linearize_impl(c, t->t, cccs, wit, lcccs_out, lin_proof_out, nullptr);
    c->ev_collect();
    return rc;
}

int lf_fold_step(lf_ctx *c, lf_transcript *t, const uint64_t *acc, const lf_witness *w_acc, const uint64_t *cm_i, const lf_witness *w_i,
                 uint64_t *lcccs_out, lf_witness **w_out, uint64_t *proof) {
    if (LF_XB(c) && t && acc && cm_i && lcccs_out && proof && c->have_ccs_any()) {
        XB x(c);
        const lf_params &P = c->params_any();
        const int ring = lf_ctx_ring(c);
        x.transcript(t);
        int rc = lf_fold_step(c, t, x.ring_in(acc, lf_lcccs_len_ring(&P, ring)), w_acc, x.ring_in(cm_i, lf_cccs_len_ring(&P, ring)), w_i, lcccs_out, w_out, proof);
        if (rc == LF_OK) { x.ring_out(lcccs_out, lf_lcccs_len_ring(&P, ring)); x.ring_out(proof, lf_proof_len_ring(&P, ring)); }
        return rc;
    }
    if (!c || !t || !acc || !w_acc || !cm_i || !w_i || !lcccs_out || !w_out || !proof) return LF_ERR_INVALID;
    if (w_acc->ctx != c || w_i->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return t->bb ? c->bb->fold_step(*t->bb, acc, w_acc, cm_i, w_i, lcccs_out, w_out, proof) : LF_ERR_INVALID;
    if (t->bb) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs || !c->A_loaded) return LF_ERR_STATE;
    const lf_params &P = c->P;
    if (c->kappa != P.kappa || c->nA_total != c->N || w_acc->N != c->N || w_i->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    std::vector<Fq3> rL;
    if (!lcccs_point(P, acc, rL)) return LF_ERR_UNSUPPORTED;  // evaluation points are always diagonal challenges
    c->tn = Tunables::read((size_t)1 << 14);
    Timeline tl;
    t_tl = &tl;
    c->ev_reset();
    c->host_tr_ms = 0;
    size_t tot = c->ev_begin(17);
    Transcript &tr = t->t;
    size_t ll = lf_lcccs_len(&P);
    u64 *lin_proof = proof, *decl = lin_proof + lin_proof_len(&P) * 24, *decr = decl + dec_proof_len(&P) * 24, *foldp = decr + dec_proof_len(&P) * 24;
    std::vector<u64> lin(ll * 24);
    u64 *eq_r_R = nullptr;
    SideState S[2];
    // Schedule (transcript order is fixed, compute order is not):
    //   lane 1 (helper thread, own stream): left decomposition (needs nothing from the linearization), then the RIGHT commit
    //           (depends only on w_i and cm_i), and -- while that commit runs on the GPU -- the host absorbs of the left part;
    //   lane 0 (this thread): linearization (latency-bound rounds), then the right evaluations at the new point.
    std::promise<int> lin_done_p;
    std::shared_future<int> lin_done = lin_done_p.get_future().share();
    // Large instances: lane 1 (two commits back to back) is the critical path and lane 0 has several ms of slack, so the
    // linearization rounds run on 16 workgroups per slot and leave the CUs to the commit kernels (C4: 44.6 -> 43.7 ms/step).
    // (with the digit-plane commits on the matrix cores lane 1 is no longer the critical path: no bound then -- C4 27.2 -> 26.1 ms/step)
    c->lin_blocks = 0;
    int rc;
    std::vector<Fq3> rR;
    // (after an RCCL handshake the agreed value decides: a rank-local environment switch must not make this rank issue a different collective sequence)
    const bool shard_threads = c->agreed_two_lanes >= 0 ? c->agreed_two_lanes == 1 : (c->tn.shard_two_lanes == 1 || (c->tn.shard_two_lanes < 0 && c->two_lanes_ok));
    if (c->sh_world > 1 && !shard_threads) {
        // Sharded step: ONE host thread issues every exchange in program order (collectives of the ranks can then never cross), the two
        // streams still overlap the right commit with the linearization rounds on the GPU.  (LF_SHARD_TWO_LANES=1: the threaded schedule
        // below with one communicator per lane.)
        auto on_lane1 = [&](auto &&fn) -> int { t_lane = 1; int r = fn(); t_lane = 0; return r; };
        u64 *ydL = nullptr, *ydR = nullptr;
        size_t evL = 0, evR = 0;
        rc = on_lane1([&]() -> int {
            RET(decompose_commit_enqueue(c, w_acc, &ydL, &evL));
            RET(decompose_commit_finish(c, acc + ((size_t)P.s + 3) * 24, ydL, evL, decl));
            RET(decompose_evals(c, acc, rL, w_acc, "L", nullptr, S[0], decl));
            return decompose_commit_enqueue(c, w_i, &ydR, &evR);               // right commit in flight on stream 1 ...
        });
        {
            HostTimer ht(c);
            tr.absorb_label("acc");
            tr.absorb_ring(acc, ll);
            tr.absorb_label("cm_i");
            tr.absorb_ring(cm_i, lf_cccs_len(&P));
        }
        if (rc == LF_OK) rc = linearize_impl(c, tr, cm_i, w_i, lin.data(), lin_proof, &eq_r_R);   // ... while the linearization runs on stream 0
        if (rc == LF_OK) {
            lcccs_point(P, lin.data(), rR);
            rc = decompose_evals(c, lin.data(), rR, w_i, "R", eq_r_R, S[1], decr);
        }
        if (rc == LF_OK) c->host_tr_ms += absorb_decomposition(P, tr, acc, decl, S[0]);
        if (rc == LF_OK) rc = on_lane1([&]() -> int { return decompose_commit_finish(c, cm_i, ydR, evR, decr); });
        c->lin_blocks = 0;
        if (rc == LF_OK) c->host_tr_ms += absorb_decomposition(P, tr, lin.data(), decr, S[1]);
    } else {
    c->bits_wit[0] = c->bits_wit[1] = nullptr;
    if (!c->tn.fold_no_sv && !c->tn.force_exchange && c->N <= c->m && (c->N & 3) == 0 && !c->tn.fold_tab_r1 && (c->m >> 1) >= c->tn.sv_min) {
        // bit-plane form of both witnesses (GEMM rounds of the folding sumcheck, v_s evaluations): first thing on the helper lane's stream,
        // enqueued from here so that the events below are recorded before anybody can wait for them
        const lf_witness *ws[2] = {w_acc, w_i};
        for (int sd = 0; sd < 2; sd++) {
            u32 *bits;
            if (c->tbuf(sd ? "sv_bits_R" : "sv_bits_L", sv_bits_words(c->N, P.K), &bits) != LF_OK) break;
            if (!c->bits_ev[sd] && hipEventCreateWithFlags(&c->bits_ev[sd], hipEventDisableTiming) != hipSuccess) break;
            launch_sv_bits(ws[sd]->planes, c->N, c->N, P.K, bits, c->st_lane[1]);
            if (hipEventRecord(c->bits_ev[sd], c->st_lane[1]) != hipSuccess) break;
            c->bits_wit[sd] = ws[sd]; c->bits_ptr[sd] = bits;
            S[sd].sv_bits = bits;
        }
    }
    c->lane1.submit([&]() -> int {
        t_lane = 1;
        struct Publish {   // whatever path this lane takes, the main thread learns whether the right side's z_k are coming
            SideState &s;
            ~Publish() { int e = 0; s.z_state.compare_exchange_strong(e, -1, std::memory_order_release); }
        } publish{S[1]};
        if (hipSetDevice(c->device) != hipSuccess) return LF_ERR_HIP;
        Timeline *const tl1 = &tl;
        u64 *yd = nullptr, *ydL = nullptr;
        size_t ev = 0;
        // the right side's z_k depend on the witness and on x_w || h = x_ccs || 1 only, not on the point r: they are built on this lane's stream behind the left
        // evaluations, and lane 0's u_s inner products wait for them (S[1].z_ev)
        bool yR_early = false;
        const u64 *yR_host = nullptr;
        // The left evaluations first, then the two commits back to back.  A commit workgroup fills its CU (registers, LDS): while one runs,
        // the other lane's kernels have the 32 CUs it leaves free -- and the linearization is bandwidth-hungry exactly at its start (z, the
        // three M z, its first rounds: 2.2 ms next to a commit, ~1.2 ms next to the evaluations), latency-bound afterwards.
        size_t evL = 0;
        RET(decompose_evals(c, acc, rL, w_acc, "L", nullptr, S[0], decl));
        tl1->mark1("L1: left evals down");
        {
            std::vector<u64> xh((size_t)(P.l + 1) * 24);
            memcpy(xh.data(), cm_i + (size_t)P.kappa * 24, (size_t)P.l * 24 * 8);
            HostRing::from_u64(1, xh.data() + (size_t)P.l * 24);
            (void)decompose_prepare_z(c, xh.data(), w_i, "R", S[1], decr);   // on failure lane 0 builds them itself
        }
        RET(decompose_commit_enqueue(c, w_acc, &ydL, &evL));
        // The download of a commit's results is enqueued right behind it -- ahead of whatever this stream is given next -- and its finish waits for that
        // event only.  (Round 3 copied y_L behind the RIGHT commit: the left absorb, the head of a 2.3 ms host chain, started when both commits were done.)
        const size_t ywords = (size_t)(P.K - 1) * P.kappa * 24;
        const bool early = c->sh_world == 1 && !c->tn.force_exchange && c->pin2(2 * ywords) == LF_OK &&
                           (c->ev_yL || hipEventCreateWithFlags(&c->ev_yL, hipEventDisableTiming) == hipSuccess) &&
                           (c->ev_yR || hipEventCreateWithFlags(&c->ev_yR, hipEventDisableTiming) == hipSuccess);
        bool yL_early = false;
        if (early && hipMemcpyAsync(c->h_pin2, ydL, ywords * 8, hipMemcpyDeviceToHost, c->stream()) == hipSuccess && hipEventRecord(c->ev_yL, c->stream()) == hipSuccess)
            yL_early = true;
        RET(decompose_commit_enqueue(c, w_i, &yd, &ev, "dec_y2"));          // right commit behind it on the same stream
        if (early && hipMemcpyAsync(c->h_pin2 + ywords, yd, ywords * 8, hipMemcpyDeviceToHost, c->stream()) == hipSuccess && hipEventRecord(c->ev_yR, c->stream()) == hipSuccess) {
            yR_early = true;
            yR_host = c->h_pin2 + ywords;
        }
        tl1->mark1("L1: commits + z_R enqueued");
        RET(decompose_commit_finish(c, acc + ((size_t)P.s + 3) * 24, ydL, evL, decl, yL_early ? c->h_pin2 : nullptr, yL_early ? c->ev_yL : nullptr));
        tl1->mark1("L1: y_L down");
        if (lin_done.get() != LF_OK) return LF_OK;                      // (the main thread reports its own error)
        tl1->mark1("L1: left absorb starts");
        absorb_decomposition(P, tr, acc, decl, S[0]);                   // ... while the host absorbs the left decomposition
        tl1->mark1("L1: left absorb done");
        return decompose_commit_finish(c, cm_i, yd, ev, decr, yR_early ? yR_host : nullptr, yR_early ? c->ev_yR : nullptr);   // cm of the linearized instance = cm_i.cm
    });
    {   // absorb_public_input (nifs.rs:175-197) -- after lane 1 has been started: the left decomposition does not depend on it
        HostTimer ht(c);
        tr.absorb_label("acc");
        tr.absorb_ring(acc, ll);
        tr.absorb_label("cm_i");
        tr.absorb_ring(cm_i, lf_cccs_len(&P));
    }
    TL_MARK("public input absorbed");
    c->vs_keep = true;
    rc = linearize_impl(c, tr, cm_i, w_i, lin.data(), lin_proof, &eq_r_R);
    c->vs_keep = false;
    TL_MARK("linearization done");
    lin_done_p.set_value(rc);
    // From here the host runs a serial Poseidon chain (left absorb, right absorb, folding challenges: ~2.4 ms at 2^20 rows) next to which the GPU only has the
    // right evaluations (0.5 ms)
    if (rc == LF_OK) {
        lcccs_point(P, lin.data(), rR);
        while (S[1].z_state.load(std::memory_order_acquire) == 0) std::this_thread::yield();   // published by lane 1 within its first millisecond (or -1)
        rc = decompose_evals(c, lin.data(), rR, w_i, "R", eq_r_R, S[1], decr);
    }
    c->vs_wit = nullptr;
    TL_MARK("right evals done");
    int rc1 = c->lane1.wait();
    c->lin_blocks = 0;
    TL_MARK("lane 1 joined");
    if (rc == LF_OK) rc = rc1;
    if (rc == LF_OK) c->host_tr_ms += absorb_decomposition(P, tr, lin.data(), decr, S[1]);
    }
    TL_MARK("right absorb done");
    if (rc == LF_OK) rc = fold_impl(c, tr, S, lcccs_out, w_out, foldp);
    c->bits_wit[0] = c->bits_wit[1] = nullptr;
    TL_MARK("fold done");
    tl.merge();
    tl.dump();
    c->tl_marks = tl.marks;
    t_tl = nullptr;
    c->ev_end(tot);
    c->ev_collect();
    if (rc != LF_OK && c->sh_world > 1) { c->comm[0].abort_peers(); c->comm[1].abort_peers(); }   // peers blocked in a collective error out instead of waiting forever
    return rc;
}

// LFDecompositionProver::prove (nifs/decomposition.rs:33-88) as its own entry point: the reference exposes the three sub-provers as
// public traits; this is the middle one.  The K decomposed witnesses stay virtual (bit-planes of `wit`).
int lf_decomposition_prove(lf_ctx *c, lf_transcript *t, const uint64_t *lcccs, const lf_witness *wit, uint64_t *lcccs_s_out, uint64_t *dec_proof_out) {
    if (LF_XB(c) && t && lcccs && dec_proof_out && c->have_ccs_any()) {
        XB x(c);
        const lf_params &P = c->params_any();
        const int ring = lf_ctx_ring(c);
        const size_t ll = lf_lcccs_len_ring(&P, ring);
        x.transcript(t);
        int rc = lf_decomposition_prove(c, t, x.ring_in(lcccs, ll), wit, lcccs_s_out, dec_proof_out);
        if (rc == LF_OK) { x.ring_out(lcccs_s_out, (size_t)P.K * ll); x.ring_out(dec_proof_out, (size_t)P.K * (P.t + x.TAU + P.l + 1 + P.kappa)); }
        return rc;
    }
    if (!c || !t || !lcccs || !wit || !dec_proof_out || wit->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return t->bb ? c->bb->decomposition_prove(*t->bb, lcccs, wit, lcccs_s_out, dec_proof_out) : LF_ERR_INVALID;
    if (t->bb) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs || !c->A_loaded) return LF_ERR_STATE;
    const lf_params &P = c->P;
    if (c->kappa != P.kappa || c->nA_total != c->N || wit->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    std::vector<Fq3> r;
    if (!lcccs_point(P, lcccs, r)) return LF_ERR_UNSUPPORTED;
    c->tn = Tunables::read((size_t)1 << 14);
    c->ev_reset();
    c->host_tr_ms = 0;
    u64 *yd = nullptr;
    size_t ev = 0;
    SideState S;
    RET(decompose_commit_enqueue(c, wit, &yd, &ev));
    RET(decompose_commit_finish(c, lcccs + ((size_t)P.s + 3) * 24, yd, ev, dec_proof_out));
    RET(decompose_evals(c, lcccs, r, wit, "L", nullptr, S, dec_proof_out));
    c->host_tr_ms += absorb_decomposition(P, t->t, lcccs, dec_proof_out, S);
    if (lcccs_s_out) memcpy(lcccs_s_out, S.lcccs.data(), S.lcccs.size() * 8);
    c->ev_collect();
    return LF_OK;
}

// LFFoldingProver::prove (nifs/folding.rs:42-130) as its own entry point.  lcccs_s = the 2K decomposed LCCCS (K of the accumulator's
// decomposition, then K of the linearized instance's), w_left / w_right = the witnesses whose base-b parts they commit to.
int lf_folding_prove(lf_ctx *c, lf_transcript *t, const uint64_t *lcccs_s, const lf_witness *w_left, const lf_witness *w_right,
                     uint64_t *lcccs_out, lf_witness **w_out, uint64_t *fold_proof_out) {
    if (LF_XB(c) && t && lcccs_s && lcccs_out && fold_proof_out && c->have_ccs_any()) {
        XB x(c);
        const lf_params &P = c->params_any();
        const int ring = lf_ctx_ring(c);
        const size_t ll = lf_lcccs_len_ring(&P, ring);
        x.transcript(t);
        int rc = lf_folding_prove(c, t, x.ring_in(lcccs_s, 2 * (size_t)P.K * ll), w_left, w_right, lcccs_out, w_out, fold_proof_out);
        if (rc == LF_OK) { x.ring_out(lcccs_out, ll); x.ring_out(fold_proof_out, (size_t)P.s * (2 * P.b + 1) + 2 * (size_t)P.K * (x.TAU + P.t)); }
        return rc;
    }
    if (!c || !t || !lcccs_s || !w_left || !w_right || !lcccs_out || !w_out || !fold_proof_out) return LF_ERR_INVALID;
    if (w_left->ctx != c || w_right->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return t->bb ? c->bb->folding_prove(*t->bb, lcccs_s, w_left, w_right, lcccs_out, w_out, fold_proof_out) : LF_ERR_INVALID;
    if (t->bb) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    const lf_params &P = c->P;
    if (w_left->N != c->N || w_right->N != c->N) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    c->tn = Tunables::read((size_t)1 << 14);
    c->ev_reset();
    c->host_tr_ms = 0;
    const size_t ll = lf_lcccs_len(&P);
    const u32 K = P.K, hl = P.l + 1;
    SideState S[2];
    for (int sd = 0; sd < 2; sd++) {
        const u64 *base = lcccs_s + (size_t)sd * K * ll * 24;
        std::vector<Fq3> r;
        if (!lcccs_point(P, base, r)) return LF_ERR_UNSUPPORTED;
        for (u32 k = 1; k < K; k++)   // the K parts of one side share r (folding/utils.rs:232-250)
            if (memcmp(base, base + (size_t)k * ll * 24, (size_t)P.s * 24 * 8) != 0) return LF_ERR_INVALID;
        const lf_witness *w = sd ? w_right : w_left;
        u64 *z, *eq_r;
        RET(c->tbuf(sd ? "z_R" : "z_L", (size_t)K * 24 * c->n, &z));
        RET(c->tbuf(sd ? "eq_r_R" : "eq_r_L", 3 * c->m, &eq_r));
        std::vector<u64> heads((size_t)K * hl * 24);
        for (u32 k = 0; k < K; k++)
            memcpy(&heads[(size_t)k * hl * 24], base + ((size_t)k * ll + P.s + 3 + P.kappa + P.t) * 24, (size_t)hl * 24 * 8);
        RET(build_z(c, w->planes, K, 1, heads.data(), z));
        RET(build_eq_dev(c, r.data(), P.s, eq_r));
        S[sd].planes = w->planes; S[sd].z = z; S[sd].eq_r = eq_r;
        S[sd].lcccs.assign(base, base + (size_t)K * ll * 24);
    }
    int rc = fold_impl(c, t->t, S, lcccs_out, w_out, fold_proof_out);
    c->ev_collect();
    return rc;
}


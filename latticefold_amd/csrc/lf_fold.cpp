// lf_fold.cpp -- the folding prover (nifs/folding.rs:74-179 with the sumcheck of nifs/folding/utils.rs:273-325) on the GPU kernels: the int8 GEMM rounds,
// the look-up-table and table rounds, the persistent tail and the fold of the witnesses; plus the two sumchecks as stand-alone ABI entry points.
#include "lf_ctx.h"

static int upload_consts(lf_ctx *c, const std::string &name, const std::vector<Fq3Const> &v, Fq3Const **out) {
    RET(c->tbuf(name, v.size() + 8, out));
    return c->h2d_small(*out, v.data(), v.size() * sizeof(Fq3Const));
}

// Host side of the mailbox protocol of a persistent tail kernel (k_fold_tail / k_lin_tail): per round poll the message, run the transcript
// (unless the device sponge does), write the challenge back.  msgs = slot of the first tail round's message, pt = its challenge.
// after_round (optional): called with the 1-based round number once that round's challenge is known (round0 = number of the first tail round)
static int tail_host_rounds(lf_ctx *c, Transcript &tr, u32 epoch, u32 nr, u32 npts, bool dev_transcript, u64 *msgs, Fq3 *pt,
                            const std::function<void(u32)> *after_round = nullptr, u32 round0 = 0) {
    TailMail *mail = c->tail_mail;
    const auto t_start = std::chrono::steady_clock::now();
    double host_us = 0, wait_us = 0;
    auto t_mark = t_start;
    const bool tl_on = t_tl && t_tl->on;
    for (u32 i = 0; i < nr; i++) {
        u32 spins = 0;
        while (__atomic_load_n(&mail->msg_seq[i], __ATOMIC_ACQUIRE) != epoch) {
            __builtin_ia32_pause();
            if ((++spins & 0xfff) == 0) {
                if (__atomic_load_n(&mail->err, __ATOMIC_RELAXED) == epoch ||
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 10.0) {
                    __atomic_store_n(&mail->abort_seq, epoch, __ATOMIC_RELEASE);   // the kernel gives up at its next wait
                    (void)hipStreamSynchronize(c->stream());
                    return LF_ERR_HIP;
                }
            }
        }
        if (tl_on) { auto nw = std::chrono::steady_clock::now(); wait_us += std::chrono::duration<double, std::micro>(nw - t_mark).count(); t_mark = nw; }
        u64 *evs = msgs + (size_t)i * npts * 24;
        memcpy(evs, (const void *)mail->msg[i], (size_t)npts * 24 * 8);
        HostTimer ht(c);
        Fq3 r;
        if (dev_transcript) r = fq3_make(mail->chal_out[i][0], mail->chal_out[i][1], mail->chal_out[i][2]);   // drawn by the device sponge
        else r = sc_round_transcript(tr, evs, npts);
        pt[i] = r;
        if (i + 1 < nr && !dev_transcript) {
            mail->chal[i][0] = r.c[0]; mail->chal[i][1] = r.c[1]; mail->chal[i][2] = r.c[2];
            __atomic_store_n(&mail->chal_seq[i], epoch, __ATOMIC_RELEASE);
        }
        if (after_round) (*after_round)(round0 + i);      // (behind the hand-over of the challenge: the device is not kept waiting)
        if (tl_on) { auto nw = std::chrono::steady_clock::now(); host_us += std::chrono::duration<double, std::micro>(nw - t_mark).count(); t_mark = nw; }
    }
    if (dev_transcript) tr.set_state((const u64 *)mail->sponge);   // the host transcript continues where the device sponge stopped
    if (tl_on) fprintf(stderr, "[timeline]    tail: %u rounds, host transcript %.1f us, waiting for the GPU %.1f us\n", nr, host_us, wait_us);
    return LF_OK;
}
// device-transcript mode: hand the sponge to the device (state_out = device [26])
static int tail_sponge_to_device(lf_ctx *c, Transcript &tr, u64 **state_out) {
    RET(c->poseidon_setup());
    *state_out = c->tail_dev_chal + 4 * TAIL_MAX_ROUNDS;   // behind the published challenges (same 4 KB scratch)
    u64 st[26];
    tr.get_state(st);
    HIPCHK(hipMemcpyAsync(*state_out, st, sizeof(st), hipMemcpyHostToDevice, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));   // st is a stack buffer
    return LF_OK;
}
// Tail rounds `round`..s of the linearization sumcheck (k_lin_tail).  cur / cure = the Mz and eq tables of round-1 (n entries, ld n).
int lin_tail_rounds(lf_ctx *c, Transcript &tr, const u64 *cur, const u64 *cure, size_t n, u64 *tout, u64 *partial, u32 round, Fq3 *point,
                           u64 *msgs, u32 deg, const std::function<void(u32)> *after_round) {
    const lf_params &P = c->P;
    RET(c->tail_setup());
    LinTailArgs A;
    A.T = cur; A.E = cure; A.Tout = tout; A.n0 = n; A.rounds = P.s - round + 1; A.deg = deg; A.partial = partial;
    RET(c->tbuf("lin_tail_priv", lin_tail_priv_words(n, P.t), &A.priv));
    A.counters = c->tail_counters; A.dev_chal = c->tail_dev_chal;
    HIPCHK(hipHostGetDevicePointer((void **)&A.mail, c->tail_mail, 0));
    if (++c->tail_epoch >= (1u << 30)) c->tail_epoch = 1;
    A.epoch = c->tail_epoch;
    A.r_first = f3c(point[round - 2]);
    A.dev_transcript = (c->tn.device_transcript && !c->xb.on) ? 1u : 0u;   // the device sponge absorbs internal-basis words
    A.pos_ark = A.pos_mds = nullptr; A.sponge_state = nullptr;
    if (A.dev_transcript) {
        RET(tail_sponge_to_device(c, tr, &A.sponge_state));
        A.pos_ark = c->d_poseidon; A.pos_mds = c->d_poseidon + 720;
    }
    if (launch_lin_tail(c->dcrt, c->desc, A, c->stream()) == 0) return LF_ERR_UNSUPPORTED;
    if (hipGetLastError() != hipSuccess) return LF_ERR_HIP;
    return tail_host_rounds(c, tr, A.epoch, A.rounds, deg + 1, A.dev_transcript != 0, msgs + (size_t)(round - 1) * (deg + 1) * 24, &point[round - 1], after_round, round);
}
// Tail rounds `round`..s of the folding sumcheck in one persistent kernel (lf_kernels.hip: k_fold_tail).  On entry `a` / `curF`
// describe the tables of round-1 (a.n entries each, leading dimension a.n) and pt[round-2] is the challenge that fixes them.
// The host side of the mailbox protocol: poll the message of a round, run the transcript, write the challenge back.
static int fold_tail_rounds(lf_ctx *c, Transcript &tr, const FoldRoundArgs &a, u64 *curF, u64 *const Fbuf[2], u64 *T_other, const Fq3Const *d_mu,
                            u64 *partial, u32 round, std::vector<Fq3> &pt, u64 *msgs, u32 deg) {
    const lf_params &P = c->P;
    RET(c->tail_setup());
    const u32 nr = P.s - round + 1;
    FoldTailArgs A;
    A.T[0] = (u64 *)a.eqL; A.T[1] = T_other;
    A.F[0] = curF; A.F[1] = curF == Fbuf[0] ? Fbuf[1] : Fbuf[0];
    A.n0 = a.n; A.rounds = nr; A.K = P.K; A.mu_pow = d_mu; A.partial = partial;
    A.counters = c->tail_counters; A.dev_chal = c->tail_dev_chal;
    RET(c->tbuf("tail_eqpriv", fold_tail_eqpriv_words(a.n, P.K), &A.eqpriv));
    HIPCHK(hipHostGetDevicePointer((void **)&A.mail, c->tail_mail, 0));
    if (++c->tail_epoch >= (1u << 30)) c->tail_epoch = 1;
    A.epoch = c->tail_epoch;
    A.r_first = f3c(pt[round - 2]);
    A.dev_transcript = (c->tn.device_transcript && !c->xb.on) ? 1u : 0u;   // the device sponge absorbs internal-basis words
    A.pos_ark = A.pos_mds = nullptr; A.sponge_state = nullptr;
    if (A.dev_transcript) {   // LF_DEVICE_TRANSCRIPT=1: hand the sponge to the device for the tail rounds
        RET(tail_sponge_to_device(c, tr, &A.sponge_state));
        A.pos_ark = c->d_poseidon; A.pos_mds = c->d_poseidon + 720;
    }
    if (launch_fold_tail(c->dcrt, A, c->num_cus, c->stream()) == 0) return LF_ERR_UNSUPPORTED;
    if (hipGetLastError() != hipSuccess) return LF_ERR_HIP;
    return tail_host_rounds(c, tr, A.epoch, nr, deg + 1, A.dev_transcript != 0, msgs + (size_t)(round - 1) * (deg + 1) * 24, &pt[round - 1]);
}

// C_pi(X) of lf_sv_rounds.h for the V weights W_b = eq((r_1..), b): coefficient table [pairs][4][3] (internal basis words)
static void sv_build_coef(lf_ctx *c, int V, const Fq3 *W, std::vector<u64> &out) {
    const int NX = 2 * V, NPR = sv_num_pairs(V);
    std::vector<Fq3> C((size_t)NPR * 4, fq3_zero());
    std::vector<SvPair> prs(NPR);
    for (int i = 0; i < NPR; i++) prs[i] = sv_pair(V, i);
    auto find = [&](unsigned s, unsigned b) {
        for (int i = 0; i < NPR; i++)
            if (prs[i].s == s && prs[i].b == b) return i;
        return -1;
    };
    // w_x(X) = wa_x + wb_x X
    std::vector<Fq3> wa(NX), wb(NX);
    for (int x = 0; x < NX; x++) {
        if (x < V) { wa[x] = W[x]; wb[x] = fq3_neg(W[x]); }
        else { wa[x] = fq3_zero(); wb[x] = W[x - V]; }
    }
    // h^3: multisets {x <= y <= z} with their multinomial multiplicity
    for (int x = 0; x < NX; x++)
        for (int y = x; y < NX; y++) {
            const Fq3 p2[3] = {c->ring.mul3(wa[x], wa[y]), fq3_add(c->ring.mul3(wa[x], wb[y]), c->ring.mul3(wb[x], wa[y])), c->ring.mul3(wb[x], wb[y])};
            for (int z = y; z < NX; z++) {
                Fq3 p3[4];
                p3[0] = c->ring.mul3(p2[0], wa[z]);
                p3[1] = fq3_add(c->ring.mul3(p2[0], wb[z]), c->ring.mul3(p2[1], wa[z]));
                p3[2] = fq3_add(c->ring.mul3(p2[1], wb[z]), c->ring.mul3(p2[2], wa[z]));
                p3[3] = c->ring.mul3(p2[2], wb[z]);
                int mult, idx;
                if (x == y && y == z) { mult = 1; idx = find(1u << x, 1u << x); }
                else if (x == y) { mult = 3; idx = find(1u << z, (1u << x) | (1u << z)); }      // y_x^2 y_z = b_x y_z
                else if (y == z) { mult = 3; idx = find(1u << x, (1u << x) | (1u << y)); }      // y_x y_y^2 = y_x b_y
                else { mult = 6; const unsigned mk = (1u << x) | (1u << y) | (1u << z); idx = find(mk, mk); }
                for (int e = 0; e < 4; e++) {
                    Fq3 t = p3[e], acc = fq3_zero();
                    for (int i = 0; i < mult; i++) acc = fq3_add(acc, t);
                    C[(size_t)idx * 4 + e] = fq3_add(C[(size_t)idx * 4 + e], acc);
                }
            }
        }
    for (int x = 0; x < NX; x++) {   // - h
        const int idx = find(1u << x, 1u << x);
        C[(size_t)idx * 4] = fq3_sub(C[(size_t)idx * 4], wa[x]);
        C[(size_t)idx * 4 + 1] = fq3_sub(C[(size_t)idx * 4 + 1], wb[x]);
    }
    out.resize((size_t)NPR * 12);
    for (size_t i = 0; i < (size_t)NPR * 4; i++)
        for (int q = 0; q < 3; q++) out[i * 3 + q] = C[i].c[q];
}

// LFFoldingProver::prove (nifs/folding.rs:42-130)
int fold_impl(lf_ctx *c, Transcript &tr, SideState *S /* [2] */, u64 *lcccs_out, lf_witness **w_out, u64 *proof) {
    const lf_params &P = c->P;
    size_t m = c->m, n = c->n, N = c->N;
    u32 K = P.K, K2 = 2 * K, deg = 2 * P.b;
    size_t ll = lf_lcccs_len(&P);
    std::vector<Fq3> alpha(K2), zeta(K2), mu(K2), beta(P.s);
    {
        HostTimer ht(c);
        tr.absorb_label("alpha_s");
        for (u32 i = 0; i < K2; i++) alpha[i] = tr.get_challenge();
        tr.absorb_label("zeta_s");
        for (u32 i = 0; i < K2; i++) zeta[i] = tr.get_challenge();
    }
    // The G tables need alpha and zeta only: their chains are enqueued HERE, and the host squeezes mu and beta (~100 permutations) while the GPU combines
    // the z_k -- the challenge order of the transcript (alpha, zeta, mu, beta: folding/utils.rs:52-95) is untouched.
    size_t ph = c->ev_begin(13);
    // powers x^{j+1}
    std::vector<Fq3Const> mu_pow((size_t)K2 * 3), a_pow((size_t)K2 * 3), z_pow((size_t)K2 * P.t);
    for (u32 i = 0; i < K2; i++) {
        Fq3 pa = alpha[i], pz = zeta[i];
        for (u32 d = 0; d < 3; d++) { a_pow[(size_t)i * 3 + d] = f3c(pa); pa = c->ring.mul3(pa, alpha[i]); }
        for (u32 j = 0; j < P.t; j++) { z_pow[(size_t)i * P.t + j] = f3c(pz); pz = c->ring.mul3(pz, zeta[i]); }
    }
    Fq3Const *d_mu, *d_ap, *d_zp;
    RET(upload_consts(c, "c_ap", a_pow, &d_ap));
    RET(upload_consts(c, "c_zp", z_pow, &d_zp));
    u64 *G[2], *eqb, *zz, *partial, *od;
    RET(c->tbuf("fold_G1", 24 * m, &G[0]));
    RET(c->tbuf("fold_G2", 24 * m, &G[1]));
    RET(c->tbuf("fold_eqb", 3 * m, &eqb));
    RET(c->tbuf("fold_zz", (size_t)P.t * 24 * n, &zz));
    RET(c->tbuf("round_partial", round_partial_words(), &partial));
    od = c->round_out();
    if (!od) return LF_ERR_HIP;
    u64 *const od_host = od, *od_shard = nullptr;
    if (c->sh_world > 1) {   // the round kernels of a sharded step leave their partial message in device memory (exchanged there)
        RET(c->tbuf("fold_round_out", 5 * 24 + 8, &od_shard));
        od = od_shard;
    }
    // sharded from round 1 on (the condition of the round loop below): the special tables live as entry slices until the hand-over to the replicated tail
    const bool shard_tabs = shard_keep(c, 1, m);
    const size_t g_r0 = shard_tabs ? (size_t)c->sh_rank * (m / (size_t)c->sh_world) : 0, g_rcnt = shard_tabs ? m / (size_t)c->sh_world : (size_t)-1;
    size_t zc_lo = 0, zc_hi = n;
    if (shard_tabs) RET(shard_col_range(c, g_r0, g_rcnt, &zc_lo, &zc_hi));
    {
        // G = sum_j M_j (sum_k zeta_k^{j+1} z_k)  +  sum_k sum_d alpha_k^{d+1} fhat_{k,d}: the two sides are independent chains -- the right one runs on
        // the (idle) stream of the helper lane next to the left one: the SpMV gathers of one side overlap the multiply-bound combination of the other
        hipStream_t s0 = c->stream(), s1 = (t_lane == 0) ? c->st_lane[1] : s0;
        u64 *zz1 = zz;
        if (s1 != s0) {
            RET(c->tbuf("fold_zz1", (size_t)P.t * 24 * n, &zz1));
            if (!c->ev_prep[0]) { HIPCHK(hipEventCreateWithFlags(&c->ev_prep[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_prep[1], hipEventDisableTiming)); }
            HIPCHK(hipEventRecord(c->ev_prep[0], s0));           // the challenge powers were uploaded on s0
            HIPCHK(hipStreamWaitEvent(s1, c->ev_prep[0], 0));
        }
        for (int sd = 0; sd < 2; sd++) {
            hipStream_t st = sd ? s1 : s0;
            u64 *zb = sd ? zz1 : zz;
            launch_lincomb_z(c->dcrt, S[sd].z + zc_lo, n, K, d_zp + (size_t)sd * K * P.t, P.t, zc_hi - zc_lo, zb + zc_lo, st);   // (sharded: the columns the rank's rows of G read)
            // (a sharded rank evaluates and fixes only the entries [rank m/G, (rank+1) m/G) of the special tables until they are gathered: only those rows of G)
            if (c->ccs_general) {
                u64 *zaos;
                RET(c->tbuf(sd ? "spmv_zaos_R" : "spmv_zaos_L", (size_t)P.t * n * 24, &zaos));
                launch_spmv_rows(c->dcrt, P.t, c->d_rowptr.data(), c->d_col.data(), c->d_val.data(), zb, (size_t)24 * n, n, zaos, G[sd], m, 0, st, g_r0, g_rcnt);
            } else
            launch_spmv_sum(c->dcrt, P.t, c->d_rowptr.data(), c->d_col.data(), c->d_val.data(), zb, (size_t)24 * n, n, G[sd], m, st, g_r0, g_rcnt);
            launch_add_fhat_comb(c->dcrt, S[sd].planes, N, K, d_ap + (size_t)sd * K * 3, G[sd], m, st, g_r0, g_rcnt);
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
        mu[K2 - 1] = fq3_one();
        tr.absorb_label("beta_s");
        for (u32 i = 0; i < P.s; i++) beta[i] = tr.get_challenge();
    }
    TL_MARK(" fold challenges");
    for (u32 i = 0; i < K2; i++) {
        Fq3 pm = mu[i];
        for (u32 d = 0; d < 3; d++) { mu_pow[(size_t)i * 3 + d] = f3c(pm); pm = c->ring.mul3(pm, mu[i]); }
    }
    RET(upload_consts(c, "c_mu", mu_pow, &d_mu));
    RET(build_eq_dev(c, beta.data(), P.s, eqb));
    // split form of the GEMM rounds (lf_sv_rounds.h): eqB fixed at r_1..r_{i-1} is c_i eq(beta_i, b) E_i[p] at entry 2p + b, E_i = eq((beta_{i+1}..beta_s), .) -- one
    // value per pair, so the GEMM of round i runs against 24 digit columns instead of 48.  E_1 here, E_2 / E_3 as pair sums when their round comes.
    u64 *svE[3] = {nullptr, nullptr, nullptr};
    const bool sv_split = !c->tn.fold_sv_no_split && P.s >= 4 && m >= 256;
    if (sv_split) {
        RET(c->tbuf("fold_svE1", 3 * (m / 2), &svE[0]));
        RET(c->tbuf("fold_svE2", 3 * (m / 4), &svE[1]));
        RET(c->tbuf("fold_svE3", 3 * (m / 8), &svE[2]));
        RET(build_eq_dev(c, beta.data() + 1, P.s - 1, svE[0]));
    }
    Fq3 sv_c = fq3_one();   // c_i = prod_{k<i} eq(beta_k, r_k)
    u32 svE_level = 1;      // E_1 .. E_level exist
    // the same form in the large table rounds after the GEMMs (k_fold_round SPLIT: three lazy products per table instead of four, the host completes the message);
    // E_l for l >= 4 share one buffer
    const bool fr_split = sv_split && c->sh_world == 1 && c->dcrt.nu2p40 && !c->tn.fold_rounds_no_split;
    u64 *svE_rest = nullptr;
    if (fr_split) RET(c->tbuf("fold_svE_rest", 3 * (m / 8) + 64, &svE_rest));
    auto svE_ptr = [&](u32 l) -> u64 * {
        if (l <= 3) return svE[l - 1];
        size_t off = 0;
        for (u32 q = 4; q < l; q++) off += 3 * (m >> q);
        return svE_rest + off;
    };
    auto svE_ensure = [&](u32 l) {   // E_{q+1} = pair sums of E_q
        for (; svE_level < l; svE_level++)
            launch_eq_pairsum(svE_ptr(svE_level), m >> svE_level, m >> (svE_level + 1), svE_ptr(svE_level + 1), m >> (svE_level + 1), c->stream());
    };
    auto sv_c_at = [&](u32 round, const std::vector<Fq3> &ptv) {   // c_round from the challenges so far
        Fq3 cc = fq3_one();
        for (u32 k = 1; k < round; k++) {
            const Fq3 b = beta[k - 1], r = ptv[k - 1];
            cc = c->ring.mul3(cc, fq3_add(c->ring.mul3(fq3_sub(fq3_one(), b), fq3_sub(fq3_one(), r)), c->ring.mul3(b, r)));
        }
        return cc;
    };
    LF_TRACE(c, "fold prepare");
    c->ev_end(ph);
    if (t_tl && t_tl->on) { (void)hipStreamSynchronize(c->stream()); TL_MARK(" fold prepare (synced)"); }

    ph = c->ev_begin(14);
    u64 *msgs = proof;
    std::vector<Fq3> pt(P.s);
    { HostTimer ht(c); sc_prologue(tr, P.s, deg); }
    const size_t Gw = (size_t)c->sh_world, gr = (size_t)c->sh_rank;
    bool sharded = Gw > 1;
    // Rounds >= 4 with many pairs: fix_variables of the f-hat tables is fused into the (ALU-bound) round kernel,
    // so the separate memory-bound pass over them vanishes (rounds 4-6 at 2^20 rows: 6.25 -> 5.6 ms).
    const bool fused = !c->tn.fold_unfused && (Gw == 1 || !c->tn.shard_plain_rounds);   // (sharded: the kernels offset their table pointers by the rank's first pair)
    const size_t fuse_min = c->tn.fuse_min;   // entries (measured: 65536 -> 16384 = -0.3 ms at 2^20 rows); tests lower it
    int fmode = 0;                 // producer of this round's pairs: 0 tables, 1 fused fix, 3 / 4 digit look-up table (rounds 3 / 4)
    const u64 *prevF = nullptr;
    size_t prevld = 0;
    // Rounds 3 and 4 of large unsharded instances never materialise the m/4-entry tables: their entries are one of 81 values
    // (four ternary digits) and come from a look-up table in LDS (k_fold_round modes 3 and 4); P.s >= 4 and m/4 >= lut_min entries.
    const size_t lut_min = c->tn.lut_min;   // default 2^14 entries (measured: C2 6.65 -> 6.52 ms, 2^18 rows 10.7 -> 9.6 ms against 2^17)
    const size_t tab_min = c->tn.tab_min;   // pairs; rounds 1-2 as table look-ups above this
    const bool use_lut = fused && P.s >= 4 && m / 4 >= lut_min && m / 4 >= 4 && !c->tn.fold_no_lut;
    // rounds 4 and 5 from product-free tables over the digit codes (k_fold_round modes 6 and 7); with round 5 on the planes too, round 4 stores no tables
    const bool use_r4tab = use_lut && c->dcrt.nu2p40 && !c->tn.fold_no_r4tab;
    // (not when the persistent tail may take over at round 5: it starts from the materialised round-4 tables)
    const bool use_r5 = use_r4tab && !c->tn.fold_no_r5tab && Gw == 1 && P.s >= 5 && (N & 3) == 0 && (c->tn.no_tail || m / 8 > c->tn.tail_n) && m / 32 >= c->tn.r5_min;
    // working tables (ping-pong): 5 special tables + materialised f-hat
    u64 *F[2], *T5[2];
    size_t half = m / 2;
    // f-hat is materialised after two rounds (m/4 entries, F[0]; round r > 3 writes its m/2^(r-1) entries to F[r odd ? 0 : 1]) -- or later: the
    // look-up-table rounds store their first tables in round 4 (m/8, F[1]), with round 5 on the planes too in round 5 (m/16, F[0]).  Sized
    // for what this step will write: 6.8 GiB -> 1.4 GiB at 2^20 rows.
    const size_t f0_ent = use_lut ? m / 16 : m / 4, f1_ent = use_r5 ? m / 32 : m / 8;
    RET(c->tbuf("fold_F0", (size_t)K2 * 3 * 24 * (f0_ent ? f0_ent : 1), &F[0]));
    RET(c->tbuf("fold_F1", (size_t)K2 * 3 * 24 * (f1_ent ? f1_ent : 1), &F[1]));
    // T5 layout per buffer: eqL[3] eqR[3] eqB[3] G1[24] G2[24] = 57 planes
    RET(c->tbuf("fold_T0", 57 * (half ? half : 1), &T5[0]));
    RET(c->tbuf("fold_T1", 57 * (half / 2 ? half / 2 : 1), &T5[1]));
    FoldRoundArgs a;
    a.eqL = S[0].eq_r; a.eqR = S[1].eq_r; a.eqB = eqb; a.G1 = G[0]; a.G2 = G[1]; a.ld = m; a.n = m;
    a.p0 = 0; a.pcnt = m / 2; a.pF0 = 0;
    const u64 *curF = nullptr;
    size_t ldF = 0;
    int flip = 0;
    // Sharded rounds: rank g evaluates the pairs of its index slice (high bits: pairs (2j,2j+1) stay local, the
    // f-hat tables exist only for that slice), the (D+1)-element partial messages are all-gathered and added mod p, every rank
    // runs the same transcript.  Once fewer than 64 pairs per rank remain the f-hat slices are gathered and the tail is replicated.
    u64 *d_lut = nullptr;
    c->sv_round_mask = 0;
    c->fold_split_mask = 0;
    u32 *sv_bits[2] = {nullptr, nullptr};
    const bool sv_two_streams = t_lane == 0 && Gw == 1;
    hipStream_t sv_g_stream = c->stream();
    const bool use_sv = !c->tn.force_exchange && !c->tn.fold_no_sv && (Gw == 1 || !c->tn.shard_plain_rounds) && N <= m && (N & 3) == 0 && !c->tn.fold_tab_r1;
    for (u32 round = 1; round <= P.s; round++) {
        fmode = 0;
        // Persistent tail: once the materialised tables are small, ONE kernel runs all remaining rounds and exchanges messages /
        // challenges with this thread through a host-mapped mailbox (k_fold_tail) -- no launches and no stream sync per round.
        if (!sharded && !c->tn.no_tail && round >= 5 && fmode == 0 && curF && ldF == a.n && a.n <= c->tn.tail_n && a.n >= 4 &&   // (a sharded step: once its tables are replicated, every rank runs its own tail)
            P.s - round + 1 <= TAIL_MAX_ROUNDS) {
            int trc = fold_tail_rounds(c, tr, a, (u64 *)curF, F, T5[flip], d_mu, partial, round, pt, msgs, deg);
            if (trc == LF_OK) {
                curF = (u64 *)curF == F[0] ? F[1] : F[0];   // the tail leaves the fully fixed tables (2 entries per row) in the other buffer
                ldF = 2;
                break;
            }
            if (trc != LF_ERR_UNSUPPORTED) return trc;   // LF_ERR_UNSUPPORTED: not launchable here -> ordinary rounds
        }
        if (round > 1) {
            Fq3Const r = f3c(pt[round - 2]);
            size_t nn = a.n / 2;
            u64 *dst = T5[flip];
            const bool handover = sharded && !shard_keep(c, 1, nn);   // this round's fix is the last one on slices: the tables are gathered, the rounds from here on replicated
            // GEMM rounds (below): the norm part needs eqB only, the G part the other four tables -- their fixes (and the G kernel) run on the
            // helper lane's idle stream next to the GEMM chain
            hipStream_t sg = c->stream();
            if (use_sv && sv_two_streams && !sharded && (int)round <= c->tn.sv_rounds && round <= 3 && nn / 2 >= c->tn.sv_min && sv_shape_ok(1 << (round - 1), nn / 2, K))
                sg = c->st_lane[1];
            sv_g_stream = sg;
            if (sharded) {
                // this rank's entries [rank nn/G, (rank+1) nn/G) of the new tables come from its own entries of the old ones
                const size_t j0 = gr * (nn / Gw), jc = nn / Gw;
                if (round == 2) {
                    launch_fix_many(c->dcrt, a.eqL + 2 * j0, a.ld, dst + j0, nn, 2 * jc, 1, r, sg);
                    launch_fix_many(c->dcrt, a.eqR + 2 * j0, a.ld, dst + 3 * nn + j0, nn, 2 * jc, 1, r, sg);
                    launch_fix_many(c->dcrt, a.eqB + 2 * j0, a.ld, dst + 6 * nn + j0, nn, 2 * jc, 1, r, sg);
                    launch_fix_many(c->dcrt, a.G1 + 2 * j0, a.ld, dst + 9 * nn + j0, nn, 2 * jc, 8, r, sg);
                    launch_fix_many(c->dcrt, a.G2 + 2 * j0, a.ld, dst + 33 * nn + j0, nn, 2 * jc, 8, r, sg);
                } else {
                    launch_fix_many(c->dcrt, a.eqL + 2 * j0, a.ld, dst + j0, nn, 2 * jc, 19, r, sg);
                }
                if (handover && round <= 3) RET(gather_slices(c, dst, 57, nn));   // (the f-hat tables are still virtual: the special tables alone; later rounds gather both in one exchange below)
            } else if (round == 2) {   // sources are the five separate full-size tables
                launch_fix_many(c->dcrt, a.eqL, a.ld, dst, nn, a.n, 1, r, sg);
                launch_fix_many(c->dcrt, a.eqR, a.ld, dst + 3 * nn, nn, a.n, 1, r, sg);
                launch_fix_many(c->dcrt, a.eqB, a.ld, dst + 6 * nn, nn, a.n, 1, r, c->stream());
                launch_fix_many(c->dcrt, a.G1, a.ld, dst + 9 * nn, nn, a.n, 8, r, sg);
                launch_fix_many(c->dcrt, a.G2, a.ld, dst + 33 * nn, nn, a.n, 8, r, sg);
            } else if (sg != c->stream()) {   // the 57-plane buffer in three pieces: eqL eqR | eqB | G1 G2
                launch_fix_many(c->dcrt, a.eqL, a.ld, dst, nn, a.n, 2, r, sg);
                launch_fix_many(c->dcrt, a.eqB, a.ld, dst + 6 * nn, nn, a.n, 1, r, c->stream());
                launch_fix_many(c->dcrt, a.G1, a.ld, dst + 9 * nn, nn, a.n, 16, r, sg);
            } else {            // source is the previous 57-plane buffer (same layout): one launch over its 19 F_{p^3} rows
                launch_fix_many(c->dcrt, a.eqL, a.ld, dst, nn, a.n, 19, r, c->stream());
            }
            if (handover) {
                // transition to the replicated rounds: the 57 special planes and the fixed f-hat slices in ONE all-gather (RCCL over xGMI), interleaved into full tables
                if (round > 3) {
                    u64 *fd = F[(round & 1) ? 0 : 1];
                    size_t lcl = ldF / 2;  // local entries after this fix
                    launch_fix_many(c->dcrt, curF, ldF, fd, lcl, ldF, K2 * 3 * 8, r, c->stream());
                    // fd is source (local layout [planes][lcl]) and destination (full tables, the parity an ordinary fix output has: the ping-pong of the following rounds stays valid)
                    const GatherPart gp[2] = {{dst + gr * lcl, nn, dst, 57}, {fd, lcl, fd, (size_t)K2 * 3 * 24}};
                    RET(gather_parts(c, gp, 2, lcl));
                    curF = fd; ldF = nn;
                    sharded = false;
                    a.eqL = dst; a.eqR = dst + 3 * nn; a.eqB = dst + 6 * nn; a.G1 = dst + 9 * nn; a.G2 = dst + 33 * nn;
                    a.ld = nn; a.n = nn; a.p0 = 0; a.pcnt = nn / 2; a.pF0 = 0;
                    flip ^= 1;
                    goto tables_ready;
                }
                sharded = false;
            }
            if (round == 3) {
                // W_b = eq((r1, r2), b), b = b0 + 2 b1 (LSB-first)
                Fq3 r1 = pt[0], r2 = pt[1], o1 = fq3_sub(fq3_one(), r1), o2 = fq3_sub(fq3_one(), r2);
                Fq3Const W[4] = {f3c(c->ring.mul3(o1, o2)), f3c(c->ring.mul3(r1, o2)), f3c(c->ring.mul3(o1, r2)), f3c(c->ring.mul3(r1, r2))};
                size_t q = sharded ? nn / Gw : nn, j0 = sharded ? gr * q : 0;   // this rank's slice of the m/4 entries
                if (use_lut) {
                    // lut[code] = sum_b (t_b - 1) W_b, code = sum_b t_b 3^b
                    std::vector<u64> lut(2 * 81 * 3);   // the 81 values, then their squares
                    for (int code = 0; code < 81; code++) {
                        Fq3 v = fq3_zero();
                        int cc = code;
                        for (int b = 0; b < 4; b++, cc /= 3) {
                            Fq3 wb = fq3_make(W[b].c[0], W[b].c[1], W[b].c[2]);
                            if (cc % 3 == 2) v = fq3_add(v, wb);
                            else if (cc % 3 == 0) v = fq3_sub(v, wb);
                        }
                        lut[3 * code] = v.c[0]; lut[3 * code + 1] = v.c[1]; lut[3 * code + 2] = v.c[2];
                        Fq3 sq = c->ring.mul3(v, v);
                        lut[3 * (81 + code)] = sq.c[0]; lut[3 * (81 + code) + 1] = sq.c[1]; lut[3 * (81 + code) + 2] = sq.c[2];
                    }
                    RET(c->tbuf("fold_lut", 2 * 81 * 3 + 8, &d_lut));
                    RET(c->h2d_small(d_lut, lut.data(), lut.size() * 8));
                    fmode = 3;
                    curF = nullptr; ldF = q;
                } else {
                    launch_fold_materialize2(c->dcrt, S[0].planes, S[1].planes, N, j0, q, K, W, F[0], c->stream());
                    curF = F[0]; ldF = q;
                }
            } else if (round > 3) {
                u64 *fd = F[(round & 1) ? 0 : 1];  // round 4 -> F[1], round 5 -> F[0], ...
                if (use_lut && round == 4) fmode = 4;
                else if (use_r5 && round == 5) fmode = 7;
                else if (fused && ldF >= fuse_min && ldF >= 4) { prevF = curF; prevld = ldF; fmode = 1; }
                else launch_fix_many(c->dcrt, curF, ldF, fd, ldF / 2, ldF, K2 * 3 * 8, r, c->stream());
                curF = fd; ldF = ldF / 2;
            }
            a.eqL = dst; a.eqR = dst + 3 * nn; a.eqB = dst + 6 * nn; a.G1 = dst + 9 * nn; a.G2 = dst + 33 * nn;
            a.ld = nn; a.n = nn;
            flip ^= 1;
        }
        if (sharded && !shard_keep(c, 1, a.n)) sharded = false;   // (round 1 of a tiny instance)
        if (sharded) { a.pcnt = a.n / 2 / Gw; a.p0 = gr * a.pcnt; a.pF0 = a.p0; }
        else { a.p0 = 0; a.pcnt = a.n / 2; a.pF0 = 0; }
    tables_ready:
        // split form of this round's kernel?  (modes 1, 6, 7; c_i and beta_i must be invertible for the host's completion)
        const u64 *Er = nullptr;
        size_t ldEr = 0;
        bool split_now = false;
        // (mode 1, the fused-fix rounds after them, measured slower in this form: 0.55 against 0.51 ms per launch at C4 -- its four reduced products per table dominate)
        if (fr_split && round >= 2 && !sharded && (fmode == 7 || (fmode == 4 && use_r4tab)) && a.pcnt >= c->tn.fold_split_min) {
            sv_c = sv_c_at(round, pt);
            const Fq3 bi = beta[round - 1];
            if ((sv_c.c[0] | sv_c.c[1] | sv_c.c[2]) && (bi.c[0] | bi.c[1] | bi.c[2])) {
                svE_ensure(round);
                Er = svE_ptr(round); ldEr = m >> round;
                split_now = true;
                c->fold_split_mask |= 1u << (round - 1);
            }
        }
        size_t ev = c->ev_begin(0);
        const int svV = 1 << (round - 1);
        if (use_sv && (int)round <= c->tn.sv_rounds && round <= 3 && a.pcnt >= c->tn.sv_min && sv_shape_ok(svV, a.pcnt, K) && (a.p0 * (size_t)svV) % 256 == 0) {
            // rounds 1..3 as exact int8 GEMMs on the matrix cores (lf_sv_rounds.h): G part on the VALU, norm part from the witness planes
            std::vector<Fq3> W((size_t)svV, fq3_one());
            for (int b = 0; b < svV; b++)
                for (u32 j = 0; j + 1 < round; j++) W[b] = c->ring.mul3(W[b], ((b >> j) & 1) ? pt[j] : fq3_sub(fq3_one(), pt[j]));
            std::vector<u64> coef;
            sv_build_coef(c, svV, W.data(), coef);
            u64 *d_coef, *gtmp, *svtp;
            unsigned char *sveb;
            int32_t *svpart, *svtot;
            RET(c->tbuf("sv_coef", coef.size() + 8, &d_coef));
            RET(c->tbuf("sv_gtmp", 128, &gtmp));
            RET(c->tbuf("sv_tp", sv_tp_words(K), &svtp));
            RET(c->tbuf("sv_eb", sv_eb_bytes(a.pcnt), &sveb));
            RET(c->tbuf("sv_part", sv_part_words(svV, a.pcnt, K), &svpart));
            RET(c->tbuf("sv_tot", sv_tot_words(svV, K), &svtot));
            RET(c->h2d_small(d_coef, coef.data(), coef.size() * 8));
            if (!sv_bits[0])   // bit-plane form of the two witnesses, once per step (the fold step builds it ahead on the other lane)
                for (int sd = 0; sd < 2; sd++) {
                    if (S[sd].sv_bits) { sv_bits[sd] = S[sd].sv_bits; continue; }
                    RET(c->tbuf(sd ? "sv_bits_R" : "sv_bits_L", sv_bits_words(N, K), &sv_bits[sd]));
                    launch_sv_bits(S[sd].planes, N, N, K, sv_bits[sd], c->stream());
                }
            hipStream_t sg = round == 1 ? (sv_two_streams && !sharded ? c->st_lane[1] : c->stream()) : sv_g_stream;
            hipEvent_t g_ready = nullptr;
            if (sg != c->stream()) {
                if (!c->ev_prep[0]) { HIPCHK(hipEventCreateWithFlags(&c->ev_prep[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_prep[1], hipEventDisableTiming)); }
                if (round == 1) {   // the tables of round 1 come from fold prepare, whose left chain ran on this lane's stream: the other stream has not seen it
                    HIPCHK(hipEventRecord(c->ev_prep[0], c->stream()));
                    HIPCHK(hipStreamWaitEvent(sg, c->ev_prep[0], 0));
                }
            }
            launch_fold_round_g(c->dcrt, a, partial, gtmp, sg);
            if (sg != c->stream()) {
                HIPCHK(hipEventRecord(c->ev_prep[1], sg));
                g_ready = c->ev_prep[1];
            }
            const u64 *Ei = nullptr;
            size_t ldE = 0;
            Fq3Const w01[2] = {};
            if (sv_split) {
                // (every GEMM round so far ran in order: rounds 1..round-1 are all GEMM rounds when this one is, their challenges are pt[0..round-2])
                sv_c = sv_c_at(round, pt);
                const Fq3 bi = beta[round - 1];
                w01[0] = f3c(c->ring.mul3(sv_c, fq3_sub(fq3_one(), bi)));
                w01[1] = f3c(c->ring.mul3(sv_c, bi));
                svE_ensure(round);
                const size_t ne = m >> round;   // entries of E_round
                Ei = svE[round - 1]; ldE = ne;
            }
            if (launch_sv_round(c->dcrt, svV, sv_bits[0], sv_bits[1], N, a.eqB, a.ld, a.p0, a.pcnt, K, d_mu, d_coef, sveb, svpart, svtot, svtp, gtmp, od, c->stream(), g_ready,
                                Ei, ldE, w01) != 0)
                return LF_ERR_UNSUPPORTED;
            c->sv_round_mask |= 1u << (round - 1);
        } else
        if ((round == 2 || (round == 1 && c->tn.fold_tab_r1)) && a.pcnt >= tab_min) {
            // round 2 (round 1 only on request: its integer kernel is faster than the gathers) as table look-ups: coefficient quadruples of h^3 - h for the 9 / 81 digit codes of a pair (host), times mu_kd (device)
            const int nd = round == 1 ? 2 : 4, ncode = round == 1 ? 9 : 81;
            std::vector<u64> poly((size_t)ncode * 12);
            const Fq3 one = fq3_one(), r1v = round == 2 ? pt[0] : fq3_zero();
            auto small = [&](int v) { return v == 0 ? fq3_zero() : (v > 0 ? (v == 1 ? one : fq3_add(one, one)) : (v == -1 ? fq3_neg(one) : fq3_neg(fq3_add(one, one)))); };
            for (int code = 0; code < ncode; code++) {
                int dg[4] = {0, 0, 0, 0}, cc = code;
                for (int b = 0; b < nd; b++, cc /= 3) dg[b] = cc % 3 - 1;
                Fq3 f0, f1;
                if (round == 1) { f0 = small(dg[0]); f1 = small(dg[1]); }
                else {   // entries d_a + (d_b - d_a) r1
                    f0 = fq3_add(small(dg[0]), c->ring.mul3(small(dg[1] - dg[0]), r1v));
                    f1 = fq3_add(small(dg[2]), c->ring.mul3(small(dg[3] - dg[2]), r1v));
                }
                Fq3 df = fq3_sub(f1, f0), f0s = c->ring.mul3(f0, f0), dfs = c->ring.mul3(df, df);
                Fq3 t1 = c->ring.mul3(f0s, df), t2 = c->ring.mul3(f0, dfs);
                Fq3 q[4] = {fq3_sub(c->ring.mul3(f0s, f0), f0), fq3_sub(fq3_add(fq3_add(t1, t1), t1), df), fq3_add(fq3_add(t2, t2), t2), c->ring.mul3(dfs, df)};
                for (int e = 0; e < 4; e++)
                    for (int w = 0; w < 3; w++) poly[(size_t)code * 12 + 3 * e + w] = q[e].c[w];
            }
            u64 *d_poly, *d_tp;
            RET(c->tbuf("fold_poly", 81 * 12 + 8, &d_poly));
            RET(c->tbuf("fold_tp", (size_t)K2 * 3 * 81 * 12, &d_tp));
            RET(c->h2d_small(d_poly, poly.data(), poly.size() * 8));
            launch_fold_round_tab(c->dcrt, (int)round, a, S[0].planes, S[1].planes, N, K, d_mu, d_poly, d_tp, partial, od, c->stream());
        } else if (round == 1) launch_fold_round1(c->dcrt, a, S[0].planes, S[1].planes, N, K, d_mu, partial, od, c->stream());
        else if (round == 2) launch_fold_round2(c->dcrt, a, S[0].planes, S[1].planes, N, K, d_mu, f3c(pt[0]), partial, od, c->stream());
        else if (fmode == 3 && c->dcrt.nu2p40) {
            u64 *mutab;
            RET(c->tbuf("fold_mutab", (size_t)3 * K2 * 3 * 81 * 4, &mutab));
            launch_fold_round_lut_mu(c->dcrt, a, S[0].planes, S[1].planes, N, d_lut, mutab, K, d_mu, partial, od, c->stream());
        } else if (fmode == 3) launch_fold_round_lut(c->dcrt, a, S[0].planes, S[1].planes, N, d_lut, K, d_mu, partial, od, c->stream());
        else if (fmode == 4 && use_r4tab) {
            u64 *r4sq, *r4mt;
            RET(c->tbuf("fold_r4sq", (size_t)6561 * 4, &r4sq));
            RET(c->tbuf("fold_r4mt", (size_t)K2 * 3 * 162 * 4, &r4mt));
            launch_fold_round_lut_fix_tab(c->dcrt, a, S[0].planes, S[1].planes, N, d_lut, f3c(pt[round - 2]), r4sq, r4mt, use_r5 ? nullptr : (u64 *)curF, ldF, K, d_mu, partial, od,
                                          c->stream(), Er, ldEr);
        } else if (fmode == 7) {
            u64 *r5xx, *r5yy, *r5mt;
            RET(c->tbuf("fold_r5xx", (size_t)6561 * 4, &r5xx));
            RET(c->tbuf("fold_r5yy", (size_t)6561 * 4, &r5yy));
            RET(c->tbuf("fold_r5mt", (size_t)K2 * 3 * 324 * 4, &r5mt));
            launch_fold_round_lut_fix5(c->dcrt, a, S[0].planes, S[1].planes, N, d_lut, f3c(pt[round - 3]), f3c(pt[round - 2]), r5xx, r5yy, r5mt, (u64 *)curF, ldF, K, d_mu, partial, od,
                                       c->stream(), Er, ldEr);
        } else if (fmode == 4) launch_fold_round_lut_fix(c->dcrt, a, S[0].planes, S[1].planes, N, d_lut, f3c(pt[round - 2]), (u64 *)curF, ldF, K, d_mu, partial, od, c->stream());
        else if (fmode == 1) launch_fold_round_fix(c->dcrt, a, prevF, prevld, f3c(pt[round - 2]), (u64 *)curF, ldF, K, d_mu, partial, od, c->stream(), Er, ldEr);
        else launch_fold_round(c->dcrt, a, curF, ldF, K, d_mu, partial, od, c->stream());
        if (split_now) {   // the G part of the message (eqL G1 + eqR G2 at X = 0..4) from its own kernel, behind the three sums of the table kernel
            u64 *partial_g;
            RET(c->tbuf("round_partial_g", round_partial_words(), &partial_g));
            launch_fold_round_g(c->dcrt, a, partial_g, od + 120, c->stream());
        }
        c->ev_end(ev);
        LF_TRACE(c, "fold round");
        u64 *evs = msgs + (size_t)(round - 1) * (deg + 1) * 24;
        if (od == od_shard) {   // sharded step: partial message in device memory -> all-gather + modular sum in stream -> host
            if (sharded) RET(exchange_modsum_dev(c, od, (size_t)(deg + 1) * 24));
            HIPCHK(hipMemcpyAsync(od_host, od, (size_t)(deg + 1) * 24 * 8, hipMemcpyDeviceToHost, c->stream()));
        }
        RET(c->lane_sync());                                  // message is in mapped host memory
        if (split_now) {
            // od_host[e][slot]: e = 0..2 the sums A_e = sum_p E[p] Q_e(p); od_host[120 + X * 24 + ..]: the G part at X = 0..4.  g(X) = c l(X) (A0 + A1 X + A2 X^2 + A3 X^3) + G(X),
            // l(X) = eq(beta_i, X); A3 from g(0) + g(1) = (the previous message at its challenge)
            HostTimer ht2(c);
            const Fq3 bi = beta[round - 1], obi = fq3_sub(fq3_one(), bi), cinv = c->ring.inv3(sv_c), binv = c->ring.inv3(bi);
            const Fq3 x = pt[round - 2];
            Fq3 wS[5];
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
            const u64 *pe = msgs + (size_t)(round - 2) * (deg + 1) * 24;
            auto ld = [&](const u64 *b, u32 e, u32 slot) { return fq3_make(b[e * 24 + 3 * slot], b[e * 24 + 3 * slot + 1], b[e * 24 + 3 * slot + 2]); };
            for (u32 slot = 0; slot < 8; slot++) {
                Fq3 S = fq3_zero();
                for (u32 j = 0; j <= deg; j++) S = fq3_add(S, c->ring.mul3(wS[j], ld(pe, j, slot)));
                const Fq3 A0 = ld(od_host, 0, slot), A1 = ld(od_host, 1, slot), A2 = ld(od_host, 2, slot);
                const u64 *gev = od_host + 120;
                const Fq3 Gsum = fq3_add(ld(gev, 0, slot), ld(gev, 1, slot));                     // G(0) + G(1)
                const Fq3 T1 = c->ring.mul3(fq3_sub(c->ring.mul3(fq3_sub(S, Gsum), cinv), c->ring.mul3(obi, A0)), binv);
                const Fq3 A3 = fq3_sub(fq3_sub(fq3_sub(T1, A0), A1), A2);
                Fq3 l = obi;
                const Fq3 dl = fq3_sub(bi, obi);
                for (u32 X = 0; X <= deg; X++) {
                    const Fq3 xs = fq3_make(X, 0, 0);
                    const Fq3 T = fq3_add(A0, c->ring.mul3(xs, fq3_add(A1, c->ring.mul3(xs, fq3_add(A2, c->ring.mul3(xs, A3))))));
                    const Fq3 g = fq3_add(c->ring.mul3(c->ring.mul3(sv_c, l), T), ld(gev, X, slot));
                    evs[X * 24 + 3 * slot] = g.c[0]; evs[X * 24 + 3 * slot + 1] = g.c[1]; evs[X * 24 + 3 * slot + 2] = g.c[2];
                    l = fq3_add(l, dl);
                }
            }
        } else
        memcpy(evs, od_host, (size_t)(deg + 1) * 24 * 8);
        HostTimer ht(c);
        pt[round - 1] = sc_round_transcript(tr, evs, deg + 1);
        if (round == 1) TL_MARK("  round 1");
        if (round == 2) TL_MARK("  round 2");
        if (round == 3) TL_MARK("  round 3");
        if (round == 6) TL_MARK("  round 6");
        if (round == 10) TL_MARK("  round 10");
    }
    TL_MARK(" fold sumcheck");
    c->ev_end(ph);

    ph = c->ev_begin(15);
    // theta, eta at r_0 (folding.rs:236-256)
    u64 *theta = proof + (size_t)P.s * (deg + 1) * 24, *eta = theta + (size_t)K2 * 72;
    u64 *eq0, *q, *red, *sm, *dpart;
    RET(c->tbuf("fold_eq0", 3 * m, &eq0));
    RET(c->tbuf("dec_q", (size_t)P.t * 24 * n, &q));
    RET(c->tbuf("red_partial", 256 * 4096, &red));
    RET(c->tbuf("dec_small", 32 * 72 + 32 * 4 * 24 + 64, &sm));
    RET(c->tbuf("dot_partial", dot_partial_words(K, P.t), &dpart));
    RET(build_eq_dev(c, pt.data(), P.s, eq0));
    // the helper lane's stream is idle here: every second q_j = M_j^T eq(r_o) is gathered there (the gathers are latency-bound: 3 x 63 us in a row at C4)
    hipStream_t s1f = (t_lane == 0 && c->sh_world == 1 && c->st_lane[1]) ? c->st_lane[1] : c->stream();
    if (s1f != c->stream() && !c->ev_prep[0]) { HIPCHK(hipEventCreateWithFlags(&c->ev_prep[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_prep[1], hipEventDisableTiming)); }
    {
        size_t c0, cnt;
        shard_slice(c, n, &c0, &cnt);   // (sharded: the eta inner products below read this rank's column slice of q_j only)
        if (s1f != c->stream() && P.t > 1) {
            HIPCHK(hipEventRecord(c->ev_prep[0], c->stream()));           // eq(r_o) is built
            HIPCHK(hipStreamWaitEvent(s1f, c->ev_prep[0], 0));
        }
        for (u32 j = 0; j < P.t; j++)
            launch_spmv_t_eq(c->dcrt, c->d_colptr[j], c->d_rowidx[j], c->d_valT[j], eq0, m, q + (size_t)j * 24 * n, n, (j & 1) ? s1f : c->stream(), c0, cnt);
        if (s1f != c->stream() && P.t > 1) {
            HIPCHK(hipEventRecord(c->ev_prep[1], s1f));
            HIPCHK(hipStreamWaitEvent(c->stream(), c->ev_prep[1], 0));
        }
    }
    // theta for both sides first, then eta; the host absorbs theta while the GPU still computes the eta dot products
    u64 *fsm;
    RET(c->tbuf("fold_small", (size_t)K2 * 72 + (size_t)K2 * P.t * 24 + 64, &fsm));
    RET(c->pin((size_t)K2 * 72 + (size_t)K2 * P.t * 24));
    u64 *hp = c->h_pin_ref();
    u64 *d_theta = fsm, *d_eta = fsm + (size_t)K2 * 72;
    // theta = f-hat_{k,d}(r_o): the f-hat tables of the sumcheck, fixed at r_1..r_{s-1}, have two entries left, so one more fix gives
    // the evaluations evaluate_mles would recompute from the witness (exact arithmetic: the same words).  Instances with fewer than
    // 4 variables never materialise the tables, and LF_THETA_EVAL=1 keeps the stand-alone evaluation (masked +-eq sums).
    if (P.s >= 4 && curF && ldF == 2 && !c->tn.theta_eval) launch_fix_final(c->dcrt, curF, K2 * 3 * 8, f3c(pt[P.s - 1]), d_theta, c->stream());
    else
    {
        size_t i0, cnt;
        shard_slice(c, N, &i0, &cnt);
        for (int sd = 0; sd < 2; sd++) RET(coef_eval_dev(c, S[sd].planes + i0, cnt, eq0 + i0, m, K, 1, red, d_theta + (size_t)sd * K * 72, N));
        RET(exchange_modsum_dev(c, d_theta, (size_t)K2 * 72));
    }
    HIPCHK(hipMemcpyAsync(hp, d_theta, (size_t)K2 * 72 * 8, hipMemcpyDeviceToHost, c->stream()));
    if (!c->ev_theta) HIPCHK(hipEventCreateWithFlags(&c->ev_theta, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->ev_theta, c->stream()));
    {
        size_t c0, cnt;
        shard_slice(c, n, &c0, &cnt);
        // the two sides stream their own 0.8 GB of z_k: side by side on the two streams (the helper lane's is idle here)
        hipStream_t s1 = (t_lane == 0 && c->sh_world == 1 && c->st_lane[1]) ? c->st_lane[1] : c->stream();
        if (s1 != c->stream()) {
            if (!c->ev_prep[0]) { HIPCHK(hipEventCreateWithFlags(&c->ev_prep[0], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_prep[1], hipEventDisableTiming)); }
            // the digits of q are the same for both sides: packed once, before the streams part
            unsigned char *ybq = nullptr;
            if (!c->tn.dot_valu && cnt >= c->tn.dot_min && P.t <= 3 && K <= 16 && ((((size_t)(S[0].z + c0)) ^ ((size_t)(S[1].z + c0))) & 15) == 0) {
                RET(c->tbuf("dot_yb", dot_i8_yb_bytes(n + 1), &ybq));
                if (launch_dot_pack_y(S[0].z + c0, q + c0, n, P.t, cnt, ybq, c->stream()) != 0) ybq = nullptr;
            }
            HIPCHK(hipEventRecord(c->ev_prep[0], c->stream()));           // q = M_j^T eq(r_o) is ready (and packed)
            HIPCHK(hipStreamWaitEvent(s1, c->ev_prep[0], 0));
            u64 *dpart1;
            RET(c->tbuf("dot_partial1", dot_partial_words(K, P.t), &dpart1));
            RET(dot_batch_dev(c, S[1].z + c0, n, K, q + c0, n, P.t, cnt, dpart1, d_eta + (size_t)K * P.t * 24, s1, "_1", ybq));
            HIPCHK(hipEventRecord(c->ev_prep[1], s1));
            RET(dot_batch_dev(c, S[0].z + c0, n, K, q + c0, n, P.t, cnt, dpart, d_eta, nullptr, "", ybq));
            HIPCHK(hipStreamWaitEvent(c->stream(), c->ev_prep[1], 0));
        } else
            for (int sd = 0; sd < 2; sd++) RET(dot_batch_dev(c, S[sd].z + c0, n, K, q + c0, n, P.t, cnt, dpart, d_eta + (size_t)sd * K * P.t * 24));
        RET(exchange_modsum_dev(c, d_eta, (size_t)K2 * P.t * 24));
    }
    HIPCHK(hipMemcpyAsync(hp + (size_t)K2 * 72, d_eta, (size_t)K2 * P.t * 24 * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipEventSynchronize(c->ev_theta));
    memcpy(theta, hp, (size_t)K2 * 72 * 8);
    {
        HostTimer ht(c);
        tr.absorb_ring(theta, (size_t)K2 * 3);
    }
    HIPCHK(hipStreamSynchronize(c->stream()));
    memcpy(eta, hp + (size_t)K2 * 72, (size_t)K2 * P.t * 24 * 8);
    TL_MARK(" theta/eta");
    std::vector<u64> rho_c((size_t)K2 * 24, 0), rho((size_t)K2 * 24);
    std::vector<int8_t> rho8((size_t)K2 * 24, 0);
    {
        HostTimer ht(c);
        tr.absorb_ring(eta, (size_t)K2 * P.t);
        // get_rhos (folding/utils.rs:116-131)
        tr.absorb_label("rho_s");
        for (u32 i = 0; i + 1 < K2; i++) tr.get_short_challenge(&rho_c[(size_t)i * 24]);
        rho_c[(size_t)(K2 - 1) * 24] = 1;
        for (u32 i = 0; i < K2; i++) {
            c->ring.crt(&rho_c[(size_t)i * 24], &rho[(size_t)i * 24]);
            for (int q2 = 0; q2 < 24; q2++) {
                u64 v = rho_c[(size_t)i * 24 + q2];
                rho8[(size_t)i * 24 + q2] = (int8_t)(v > LF_P / 2 ? -(int64_t)(LF_P - v) : (int64_t)v);
            }
        }
    }
    // f_0 in the coefficient domain -> new witness
    int8_t *d_rho;
    RET(c->tbuf("c_rho", (size_t)K2 * 24 + 64, &d_rho));
    HIPCHK(hipMemcpyAsync(d_rho, rho8.data(), rho8.size(), hipMemcpyHostToDevice, c->stream()));
    int32_t *npl;
    RET(lf_planes_alloc(c, N * 24 * 4, &npl));
    LF_TRACE(c, "theta/eta");
    launch_fold_witness(S[0].planes, S[1].planes, N, K, d_rho, npl, c->stream());
    // Witness::from_f (arith.rs:299-313): f = CRT(f_coeff) and w_ccs = CRT(recompose(f_coeff, B, L)) of the folded witness, behind compute_f_0 on the same stream
    u64 *nf = nullptr, *nw = nullptr;
    const size_t nf_bytes = N * 24 * 8, nw_bytes = (size_t)P.wit_len * 24 * 8;
    RET(lf_planes_alloc(c, nf_bytes, (int32_t **)&nf));
    RET(lf_planes_alloc(c, nw_bytes, (int32_t **)&nw));
    launch_recompose_crt(c->dcrt, npl, N, (u32)N, 1, P.B, 1, 0, nf, N, 0, c->stream());
    launch_recompose_crt(c->dcrt, npl, N, P.wit_len, P.L, P.B, 1, 0, nw, P.wit_len, 0, c->stream());
    LF_TRACE(c, "fold_witness");
    TL_MARK("  eta absorbed, rho drawn, fold_witness enqueued");

    // compute_v0_u0_x0_cm_0 (folding/utils.rs:460-521) on the host while the GPU folds the witness
    {
    HostTimer ht(c);
    u64 *o = lcccs_out;
    for (u32 i = 0; i < P.s; i++, o += 24) HostRing::from_fq3(pt[i], o);
    {   // v_0 = rot_lin_combination(rho_coeff, theta) (cyclotomic-rings/src/rotation.rs:85-104)
        Fq3 res[24];
        for (int j = 0; j < 24; j++) res[j] = fq3_zero();
        for (u32 i = 0; i < K2; i++) {
            u64 rot[24];
            memcpy(rot, &rho_c[(size_t)i * 24], sizeof(rot));
            const u64 *th = theta + (size_t)i * 72;
            for (int bi = 0; bi < 24; bi++) {
                Fq3 b = fq3_make(th[3 * bi], th[3 * bi + 1], th[3 * bi + 2]);
                for (int j = 0; j < 24; j++)
                    if (rot[j]) res[j] = fq3_add(res[j], fq3_mul_fq(b, rot[j]));
                // multiply by X modulo X^24 - X^12 + 1
                u64 top = rot[23];
                for (int j = 23; j > 0; j--) rot[j] = rot[j - 1];
                rot[0] = fq_neg(top);
                rot[12] = fq_add(rot[12], top);
            }
        }
        for (int j = 0; j < 24; j++) { o[3 * j] = res[j].c[0]; o[3 * j + 1] = res[j].c[1]; o[3 * j + 2] = res[j].c[2]; }
        o += 72;
    }
    u64 tmp[24];
    auto part = [&](u32 i) { return &S[i < K ? 0 : 1].lcccs[(size_t)(i % K) * ll * 24]; };
    for (u32 q2 = 0; q2 < P.kappa; q2++, o += 24) {
        memset(o, 0, 24 * 8);
        for (u32 i = 0; i < K2; i++) { c->ring.mul_ntt(part(i) + ((size_t)P.s + 3 + q2) * 24, &rho[(size_t)i * 24], tmp); HostRing::add(o, tmp, o); }
    }
    for (u32 j = 0; j < P.t; j++, o += 24) {
        memset(o, 0, 24 * 8);
        for (u32 i = 0; i < K2; i++) { c->ring.mul_ntt(&rho[(size_t)i * 24], eta + ((size_t)i * P.t + j) * 24, tmp); HostRing::add(o, tmp, o); }
    }
    for (u32 q2 = 0; q2 < P.l + 1; q2++, o += 24) {
        memset(o, 0, 24 * 8);
        for (u32 i = 0; i < K2; i++) { c->ring.mul_ntt(&rho[(size_t)i * 24], part(i) + ((size_t)P.s + 3 + P.kappa + P.t + q2) * 24, tmp); HostRing::add(o, tmp, o); }
    }
    }
    TL_MARK("  folded instance on the host");
    HIPCHK(hipStreamSynchronize(c->stream()));
    *w_out = new lf_witness{c, npl, N, c->device, N * 24 * 4};
    if (nf) { (*w_out)->f_ntt = nf; (*w_out)->f_bytes = nf_bytes; (*w_out)->w_ccs = nw; (*w_out)->w_bytes = nw_bytes; }
    TL_MARK(" rho + fold_witness");
    c->ev_end(ph);
    return LF_OK;
}

// ---- generic linearization-shaped sumcheck through the ABI (tests / SURVEY 8b) -------------------------------------------------
int lf_sumcheck_lin_begin(lf_ctx *c, const uint64_t *tables, const uint64_t *eq_point) {
    if (LF_XB(c) && tables && eq_point && c->have_ccs_any()) { XB x(c); const lf_params &P = c->params_any(); return lf_sumcheck_lin_begin(c, x.ring_in(tables, (size_t)P.t * c->m_any()), x.ext_in(eq_point, P.s)); }
    if (!c || !tables || !eq_point) return LF_ERR_INVALID;
    if (c->bb) return c->bb->sumcheck_lin_begin(tables, eq_point);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    size_t m = c->m;
    u64 *mz, *eqb;
    RET(c->tbuf("sc_tab0", (size_t)P.t * 24 * m, &mz));
    RET(c->tbuf("sc_eq0", 3 * m, &eqb));
    for (u32 j = 0; j < P.t; j++) RET(up_ring(c, tables + (size_t)j * m * 24, m, mz + (size_t)j * 24 * m));
    std::vector<Fq3> pt(P.s);
    for (u32 i = 0; i < P.s; i++) pt[i] = fq3_make(eq_point[3 * i], eq_point[3 * i + 1], eq_point[3 * i + 2]);
    RET(build_eq_dev(c, pt.data(), P.s, eqb));
    c->sc_round = 0; c->sc_n = m; c->sc_cur = 0;
    return LF_OK;
}
int lf_sumcheck_lin_round(lf_ctx *c, const uint64_t *r_prev, uint64_t *evals_out) {
    if (LF_XB(c) && evals_out) { XB x(c); int rc = lf_sumcheck_lin_round(c, x.ext_in(r_prev, 1), evals_out); if (rc == LF_OK) x.ring_out(evals_out, c->params_any().d + 2); return rc; }
    if (!c || !evals_out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->sumcheck_lin_round(r_prev, evals_out);
    std::lock_guard<std::mutex> g(c->mu);
    if (c->sc_round < 0 || c->sc_round >= (int)c->P.s) return LF_ERR_STATE;  // "Prover is not active"
    if ((c->sc_round == 0) != (r_prev == nullptr)) return LF_ERR_STATE;      // "first round should be prover first" / "verifier message is empty"
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    size_t m = c->m;
    u64 *tab[2], *eq[2], *partial, *od;
    RET(c->tbuf("sc_tab0", (size_t)P.t * 24 * m, &tab[0]));
    RET(c->tbuf("sc_tab1", (size_t)P.t * 24 * (m / 2 ? m / 2 : 1), &tab[1]));
    RET(c->tbuf("sc_eq0", 3 * m, &eq[0]));
    RET(c->tbuf("sc_eq1", 3 * (m / 2 ? m / 2 : 1), &eq[1]));
    RET(c->tbuf("round_partial", round_partial_words(), &partial));
    RET(c->tbuf("round_out", 5 * 24, &od));
    if (r_prev) {
        Fq3Const r; r.c[0] = r_prev[0]; r.c[1] = r_prev[1]; r.c[2] = r_prev[2];
        int src = c->sc_cur, dst = src ^ 1;
        launch_fix_many(c->dcrt, tab[src], c->sc_n, tab[dst], c->sc_n / 2, c->sc_n, P.t * 8, r, c->stream());
        launch_fix_many(c->dcrt, eq[src], c->sc_n, eq[dst], c->sc_n / 2, c->sc_n, 1, r, c->stream());
        c->sc_cur = dst; c->sc_n /= 2;
    }
    launch_lin_round(c->dcrt, c->desc, tab[c->sc_cur], c->sc_n, eq[c->sc_cur], c->sc_n, c->sc_n, P.d + 1, partial, od, c->stream());
    c->sc_round++;
    return down_small(c, od, (size_t)(P.d + 2) * 24, evals_out);
}
int lf_sumcheck_lin_end(lf_ctx *c) {
    if (!c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->sumcheck_lin_end();
    std::lock_guard<std::mutex> g(c->mu);
    c->sc_round = -1;
    return LF_OK;
}

// PoseidonSponge on the device: a script of absorb / squeeze operations on a fresh sponge, one wave.  ops[i] =
// (kind << 24) | count: kind 0 absorbs the next `count` words of absorb_words, kind 1 squeezes `count` words into squeezed_out.
// state_out (optional, 26 words): the 24 state words, the rate index and the mode (1 = squeezing) afterwards.
int lf_device_sponge(lf_ctx *c, const uint32_t *ops, size_t nops, const uint64_t *absorb_words, size_t n_words, uint64_t *squeezed_out,
                     size_t n_out, uint64_t *state_out) {
    if (!c || !ops || !nops || (!absorb_words && n_words) || (!squeezed_out && n_out)) return LF_ERR_INVALID;
    if (c->bb) return LF_ERR_UNSUPPORTED;   // the BabyBear transcript stays on the host
    size_t na = 0, ns = 0;
    for (size_t i = 0; i < nops; i++) {
        if ((ops[i] >> 24) > 1) return LF_ERR_INVALID;
        ((ops[i] >> 24) ? ns : na) += ops[i] & 0xffffff;
    }
    if (na != n_words || ns != n_out) return LF_ERR_INVALID;
    for (size_t i = 0; i < n_words; i++)
        if (absorb_words[i] >= LF_P) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    RET(c->poseidon_setup());
    u64 *dw, *dout;
    u32 *dops;
    RET(c->tbuf("sp_words", n_words + 8, &dw));
    RET(c->tbuf("sp_out", n_out + 32, &dout));
    RET(c->tbuf("sp_ops", nops + 8, &dops));
    if (n_words) HIPCHK(hipMemcpyAsync(dw, absorb_words, n_words * 8, hipMemcpyHostToDevice, c->stream()));
    HIPCHK(hipMemcpyAsync(dops, ops, nops * 4, hipMemcpyHostToDevice, c->stream()));
    launch_sponge_script(c->d_poseidon, c->d_poseidon + 720, dops, (u32)nops, dw, dout, dout + n_out, c->stream());
    std::vector<u64> h(n_out + 26);
    HIPCHK(hipMemcpyAsync(h.data(), dout, (n_out + 26) * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    if (n_out) memcpy(squeezed_out, h.data(), n_out * 8);
    if (state_out) memcpy(state_out, h.data() + n_out, 26 * 8);
    return LF_OK;
}

// ---- the folding sumcheck through the ABI: MLSumcheck::prove_as_subprotocol (utils/sumcheck.rs:53-80) with the comb
// function of nifs/folding/utils.rs:273-325, split at the transcript.  `tables` is the reference's mle list of
// create_sumcheck_polynomial (folding/utils.rs:200-259): [eq(r_L), G_L, eq(r_R), G_R, eq(beta), f-hat_{0,0} .. f-hat_{2K-1,tau-1}],
// P = 5 + 2K*tau tables of m ring elements; the three eq tables must be slot-constant (they are diagonal embeddings in the reference).
int lf_sumcheck_fold_begin(lf_ctx *c, const uint64_t *tables, const uint64_t *mu) {
    if (LF_XB(c) && tables && mu && c->have_ccs_any()) { XB x(c); const lf_params &P = c->params_any(); return lf_sumcheck_fold_begin(c, x.ring_in(tables, (size_t)(5 + 2 * P.K * x.TAU) * c->m_any()), x.ext_in(mu, 2 * P.K)); }
    if (!c || !tables || !mu) return LF_ERR_INVALID;
    if (c->bb) return c->bb->sumcheck_fold_begin(tables, mu);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    const size_t m = c->m;
    const u32 K2 = 2 * P.K;
    static const int eq_idx[3] = {0, 2, 4};
    for (int e = 0; e < 3; e++) {   // slot-constant check of the eq tables
        const u64 *tb = tables + (size_t)eq_idx[e] * m * 24;
        for (size_t i = 0; i < m; i++)
            for (int sl = 1; sl < 8; sl++)
                if (memcmp(tb + i * 24, tb + i * 24 + 3 * sl, 24) != 0) return LF_ERR_UNSUPPORTED;
    }
    u64 *T, *F, *tmp;
    RET(c->tbuf("sf_T0", 57 * m, &T));
    RET(c->tbuf("sf_F0", (size_t)K2 * 3 * 24 * m, &F));
    RET(c->tbuf("sf_tmp", 24 * m, &tmp));
    for (int e = 0; e < 3; e++) {   // eqL, eqR, eqB -> fq3 tables (slot 0 of the ring table)
        RET(up_ring(c, tables + (size_t)eq_idx[e] * m * 24, m, tmp));
        HIPCHK(hipMemcpyAsync(T + (size_t)3 * e * m, tmp, 3 * m * 8, hipMemcpyDeviceToDevice, c->stream()));
    }
    RET(up_ring(c, tables + (size_t)1 * m * 24, m, T + 9 * m));
    RET(up_ring(c, tables + (size_t)3 * m * 24, m, T + 33 * m));
    for (u32 i = 0; i < K2 * 3; i++) RET(up_ring(c, tables + (size_t)(5 + i) * m * 24, m, F + (size_t)i * 24 * m));
    std::vector<Fq3Const> mu_pow((size_t)K2 * 3);
    for (u32 i = 0; i < K2; i++) {
        Fq3 mi = fq3_make(mu[3 * i], mu[3 * i + 1], mu[3 * i + 2]), pm = mi;
        for (u32 d = 0; d < 3; d++) { mu_pow[(size_t)i * 3 + d] = f3c(pm); pm = c->ring.mul3(pm, mi); }
    }
    Fq3Const *d_mu;
    RET(upload_consts(c, "sf_mu", mu_pow, &d_mu));
    c->sf_round = 0; c->sf_n = m; c->sf_cur = 0;
    return LF_OK;
}
int lf_sumcheck_fold_round(lf_ctx *c, const uint64_t *r_prev, uint64_t *evals_out) {
    if (LF_XB(c) && evals_out) { XB x(c); int rc = lf_sumcheck_fold_round(c, x.ext_in(r_prev, 1), evals_out); if (rc == LF_OK) x.ring_out(evals_out, 2 * c->params_any().b + 1); return rc; }
    if (!c || !evals_out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->sumcheck_fold_round(r_prev, evals_out);
    std::lock_guard<std::mutex> g(c->mu);
    if (c->sf_round < 0 || c->sf_round >= (int)c->P.s) return LF_ERR_STATE;   // "Prover is not active" (sumcheck/prover.rs:63)
    if ((c->sf_round == 0) != (r_prev == nullptr)) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    const lf_params &P = c->P;
    const size_t m = c->m;
    const u32 K2 = 2 * P.K;
    u64 *T[2], *F[2], *partial, *od;
    Fq3Const *d_mu;
    RET(c->tbuf("sf_T0", 57 * m, &T[0]));
    RET(c->tbuf("sf_T1", 57 * (m / 2 ? m / 2 : 1), &T[1]));
    RET(c->tbuf("sf_F0", (size_t)K2 * 3 * 24 * m, &F[0]));
    RET(c->tbuf("sf_F1", (size_t)K2 * 3 * 24 * (m / 2 ? m / 2 : 1), &F[1]));
    RET(c->tbuf("sf_mu", (size_t)K2 * 3 + 8, &d_mu));
    RET(c->tbuf("round_partial", round_partial_words(), &partial));
    RET(c->tbuf("round_out", 5 * 24, &od));
    if (r_prev) {
        Fq3Const r; r.c[0] = r_prev[0]; r.c[1] = r_prev[1]; r.c[2] = r_prev[2];
        int src = c->sf_cur, dst = src ^ 1;
        launch_fix_many(c->dcrt, T[src], c->sf_n, T[dst], c->sf_n / 2, c->sf_n, 19, r, c->stream());
        launch_fix_many(c->dcrt, F[src], c->sf_n, F[dst], c->sf_n / 2, c->sf_n, K2 * 3 * 8, r, c->stream());
        c->sf_cur = dst; c->sf_n /= 2;
    }
    const size_t n = c->sf_n;
    const u64 *t5 = T[c->sf_cur];
    FoldRoundArgs a;
    a.eqL = t5; a.eqR = t5 + 3 * n; a.eqB = t5 + 6 * n; a.G1 = t5 + 9 * n; a.G2 = t5 + 33 * n;
    a.ld = n; a.n = n; a.p0 = 0; a.pcnt = n / 2; a.pF0 = 0;
    launch_fold_round(c->dcrt, a, F[c->sf_cur], n, P.K, d_mu, partial, od, c->stream());
    c->sf_round++;
    return down_small(c, od, (size_t)(2 * P.b + 1) * 24, evals_out);
}
int lf_sumcheck_fold_end(lf_ctx *c) {
    if (!c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->sumcheck_fold_end();
    std::lock_guard<std::mutex> g(c->mu);
    c->sf_round = -1;
    return LF_OK;
}

// compute_f_0 (nifs/folding.rs:258-268): out[j] = sum_i coef_i (.) tables_i[j] with ring-element coefficients (8 distinct slots)
int lf_lincomb(lf_ctx *c, const uint64_t *coef, const uint64_t *tables, size_t n_terms, size_t len, uint64_t *out) {
    if (LF_XB(c) && coef && tables && out) { XB x(c); int rc = lf_lincomb(c, x.ring_in(coef, n_terms), x.ring_in(tables, n_terms * len), n_terms, len, out); if (rc == LF_OK) x.ring_out(out, len); return rc; }
    if (!c || !coef || !tables || !out || !n_terms || !len) return LF_ERR_INVALID;
    if (c->bb) return c->bb->lincomb(coef, tables, n_terms, len, out);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *X, *o;
    RET(c->tbuf("io_a", n_terms * len * 24, &X));
    RET(c->tbuf("io_b", len * 24, &o));
    for (size_t i = 0; i < n_terms; i++) RET(up_ring(c, tables + i * len * 24, len, X + i * 24 * len));
    std::vector<Fq3Const> cf(n_terms * 8);
    for (size_t i = 0; i < n_terms; i++)
        for (int sl = 0; sl < 8; sl++)
            for (int q = 0; q < 3; q++) cf[i * 8 + sl].c[q] = coef[i * 24 + 3 * sl + q];
    Fq3Const *d_cf;
    RET(upload_consts(c, "lc_coef", cf, &d_cf));
    launch_lincomb_z(c->dcrt, X, len, (u32)n_terms, d_cf, 1, len, o, c->stream(), 1);
    return down_ring(c, o, len, out);
}
// calculate_challenged_mz_mle (nifs/folding.rs:208-226) and the f-hat half of prepare_g1_and_3_k_mles_list (folding/utils.rs:524-546):
// out[x] = sum_{i<groups} sum_{j<per_group} c_i^{j+1} T_{i,j}[x] (the reference's Horner loop `mle += M; mle *= c_i` over j reversed)
int lf_horner_combine(lf_ctx *c, const uint64_t *tables, size_t groups, size_t per_group, size_t len, const uint64_t *challenges, uint64_t *out) {
    if (LF_XB(c) && tables && challenges && out) { XB x(c); int rc = lf_horner_combine(c, x.ring_in(tables, groups * per_group * len), groups, per_group, len, x.ext_in(challenges, groups), out); if (rc == LF_OK) x.ring_out(out, len); return rc; }
    if (!c || !tables || !challenges || !out || !groups || !per_group || !len) return LF_ERR_INVALID;
    if (c->bb) return c->bb->horner_combine(tables, groups, per_group, len, challenges, out);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    const size_t nt = groups * per_group;
    u64 *X, *o;
    RET(c->tbuf("io_a", nt * len * 24, &X));
    RET(c->tbuf("io_b", len * 24, &o));
    for (size_t i = 0; i < nt; i++) RET(up_ring(c, tables + i * len * 24, len, X + i * 24 * len));
    std::vector<Fq3Const> cf(nt);
    for (size_t i = 0; i < groups; i++) {
        Fq3 ci = fq3_make(challenges[3 * i], challenges[3 * i + 1], challenges[3 * i + 2]), pw = ci;
        for (size_t j = 0; j < per_group; j++) { cf[i * per_group + j] = f3c(pw); pw = c->ring.mul3(pw, ci); }
    }
    Fq3Const *d_cf;
    RET(upload_consts(c, "lc_coef", cf, &d_cf));
    launch_lincomb_z(c->dcrt, X, len, (u32)nt, d_cf, 1, len, o, c->stream(), 0);
    return down_ring(c, o, len, out);
}


// lf_capi.cpp -- the C ABI (include/lfhip.h) of the Goldilocks backend, part 1: context, ring tables, sharding / RCCL set-up, staging, the component entry
// points (CRT, decomposition, Ajtai commitments, eq tables, MLE evaluations, SpMV), constraint-system load, device-resident witnesses, transcripts, timing
// read-outs and the host verifier.  The provers are in lf_prove.cpp (linearization, decomposition, entry points) and lf_fold.cpp (the folding prover).
#include "lf_ctx.h"

const char *lf_strerror(int code) {
    switch (code) {
        case LF_OK: return "ok";
        case LF_ERR_INVALID: return "invalid argument / wrong length";
        case LF_ERR_HIP: return "HIP runtime error (no GPU or out of device memory)";
        case LF_ERR_UNSUPPORTED: return "unsupported parameter";
        case LF_ERR_BAD_TABLES: return "ring tables are not a ring isomorphism";
        case LF_ERR_NORM: return "witness coefficient exceeds the decomposition bound";
        case LF_ERR_SIZE_BOUNDS: return "invalid size bounds (m must be >= wit_len*L, power of two)";
        case LF_ERR_STATE: return "call sequence misuse";
        case LF_ERR_REJECT: return "verifier rejected the proof";
    }
    return "unknown error";
}
const char *lf_phase_name(int i) { return (i >= 0 && i < LF_N_PHASES) ? PHASE_NAMES[i] : ""; }

// ---------------------------------------------------------------------------------------------------------------
static int install_tables(lf_ctx *c, u64 nonres, const u64 *y) {
    CrtTables T;
    if (build_crt_tables(nonres, y, T) != 0) return LF_ERR_BAD_TABLES;
    c->ring.T = T;
    c->dcrt = make_dev_crt(T);
    if (!c->d_icrt) HIPCHK(lf_dev_malloc(&c->d_icrt, 576 * 8));
    HIPCHK(hipMemcpy(c->d_icrt, &T.icrt[0][0], 576 * 8, hipMemcpyHostToDevice));
    // compressed rows for the digit pass of the general commitment (lf_ajtai_i8g.hip k_i8g_cut_ntt): the shipped tables have one entry per slot
    u64 sv[24 * 8];
    u32 sc[24 * 8];
    bool sparse = true;
    for (int r = 0; r < 24 && sparse; r++) {
        int q = 0;
        for (int col = 0; col < 24; col++)
            if (T.icrt[r][col]) {
                if (q == 8) { sparse = false; break; }
                sv[r * 8 + q] = T.icrt[r][col]; sc[r * 8 + q] = (u32)col; q++;
            }
        for (; q < 8; q++) { sv[r * 8 + q] = 0; sc[r * 8 + q] = 0xFFFFFFFFu; }
    }
    if (sparse) {
        if (!c->d_icrt_sp_val) { HIPCHK(lf_dev_malloc(&c->d_icrt_sp_val, sizeof(sv))); HIPCHK(lf_dev_malloc(&c->d_icrt_sp_col, sizeof(sc))); }
        HIPCHK(hipMemcpy(c->d_icrt_sp_val, sv, sizeof(sv), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_icrt_sp_col, sc, sizeof(sc), hipMemcpyHostToDevice));
    } else if (c->d_icrt_sp_val) {
        (void)hipFree(c->d_icrt_sp_val); (void)hipFree(c->d_icrt_sp_col);
        c->d_icrt_sp_val = nullptr; c->d_icrt_sp_col = nullptr;
    }
    return LF_OK;
}

int lf_ctx_create_ring(lf_ctx **out, int device, int ring) {
    if (ring == LF_RING_GOLDILOCKS) return lf_ctx_create(out, device);
    if (!out || ring != LF_RING_BABYBEAR) return LF_ERR_INVALID;
    lf_ctx *c = new lf_ctx();
    c->device = device;
    int rc = lfbb::BbCtx::create(&c->bb, c, device);
    if (rc != LF_OK) { delete c; return rc; }
    *out = c;
    return LF_OK;
}
int lf_ctx_device(const lf_ctx *c) { return c->device; }
int lf_ctx_ring(const lf_ctx *c) { return c && c->bb ? LF_RING_BABYBEAR : LF_RING_GOLDILOCKS; }
int lf_ring_words(int ring) { return ring == LF_RING_BABYBEAR ? 72 : (ring == LF_RING_GOLDILOCKS ? 24 : 0); }
int lf_ring_tau(int ring) { return ring == LF_RING_BABYBEAR ? 9 : (ring == LF_RING_GOLDILOCKS ? 3 : 0); }
uint64_t lf_ring_modulus(int ring) { return ring == LF_RING_BABYBEAR ? (uint64_t)lfbb::BB_P : (ring == LF_RING_GOLDILOCKS ? LF_P : 0); }

int lf_ctx_create(lf_ctx **out, int device) {
    if (!out) return LF_ERR_INVALID;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device < 0 || device >= cnt) return LF_ERR_HIP;
    HIPCHK(hipSetDevice(device));
    lf_ctx *c = new lf_ctx();
    c->device = device;
    {   // lane 1 carries the critical chain of a fold step (two commits back to back); its kernels get dispatch priority over
        // lane 0's latency-bound linearization, which has slack (LF_NO_PRIO=1: equal priorities)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool prio = true;
        const int p0 = least;
        if (hipStreamCreateWithPriority(&c->st_lane[0], hipStreamDefault, prio ? p0 : 0) != hipSuccess ||
            hipStreamCreateWithPriority(&c->st_lane[1], hipStreamDefault, prio ? greatest : 0) != hipSuccess ||
            hipStreamCreateWithPriority(&c->st_io, hipStreamDefault, least) != hipSuccess) { delete c; return LF_ERR_HIP; }
    }
    (void)hipEventCreateWithFlags(&c->ev_block, hipEventBlockingSync | hipEventDisableTiming);
    u64 nr, y[24];
    default_ring(&nr, y);
    int rc = install_tables(c, nr, y);
    if (rc != LF_OK) { delete c; return rc; }
    *out = c;
    return LF_OK;
}
static void free_ccs(lf_ctx *c) {
    for (auto p : c->d_rowptr) (void)hipFree(p);
    for (auto p : c->d_col) (void)hipFree(p);
    for (auto p : c->d_val) (void)hipFree(p);
    for (auto p : c->d_colptr) (void)hipFree(p);
    for (auto p : c->d_rowidx) (void)hipFree(p);
    for (auto p : c->d_valT) (void)hipFree(p);
    c->d_rowptr.clear(); c->d_col.clear(); c->d_val.clear(); c->d_colptr.clear(); c->d_rowidx.clear(); c->d_valT.clear();
    c->have_ccs = false;
}
static void planes_pool_drop(int device);
void lf_ctx_destroy(lf_ctx *c) {
    if (!c) return;
    while (c->io_jobs.load() > 0) std::this_thread::yield();   // ingestion workers still use the context (their handles may be finished later: lf_witness_job_finish needs no context)
    (void)hipSetDevice(c->device);
    planes_pool_drop(c->device);
    if (c->bb) { c->bb->destroy(); delete c; return; }
    (void)hipStreamSynchronize(c->st_lane[0]);
    (void)hipStreamSynchronize(c->st_lane[1]);
    if (c->st_io) (void)hipStreamSynchronize(c->st_io);
    free_ccs(c);
    for (auto &kv : c->bufs) kv.second.release();
    if (c->dAb) (void)hipFree(c->dAb);
    for (int l = 0; l < LF_NLANES; l++) if (c->stage[l]) (void)hipHostFree(c->stage[l]);
    if (c->d_icrt) (void)hipFree(c->d_icrt);
    if (c->d_icrt_sp_val) { (void)hipFree(c->d_icrt_sp_val); (void)hipFree(c->d_icrt_sp_col); }
    for (int l = 0; l < LF_NLANES; l++)
        if (c->h_pin_lane[l]) (void)hipHostFree(c->h_pin_lane[l]);
    if (c->h_pin2) { (void)hipHostFree(c->h_pin2); c->h_pin2 = nullptr; }
    for (auto &e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int l = 0; l < LF_NLANES; l++)
        if (c->h_round[l]) (void)hipHostFree(c->h_round[l]);
    if (c->ev_block) (void)hipEventDestroy(c->ev_block);
    c->comm[0].destroy();
    c->comm[1].destroy();
    if (c->tail_mail) (void)hipHostFree(c->tail_mail);
    if (c->tail_counters) (void)hipFree(c->tail_counters);
    if (c->d_poseidon) (void)hipFree(c->d_poseidon);
    if (c->ev_theta) (void)hipEventDestroy(c->ev_theta);
    if (c->ev_aux) (void)hipEventDestroy(c->ev_aux);
    if (c->st_aux) (void)hipStreamDestroy(c->st_aux);
    if (c->h_aux) (void)hipHostFree(c->h_aux);
    for (int i = 0; i < 2; i++) {
        if (c->ev_prep[i]) (void)hipEventDestroy(c->ev_prep[i]);
        if (c->bits_ev[i]) (void)hipEventDestroy(c->bits_ev[i]);
    }
    (void)hipStreamDestroy(c->st_lane[0]);
    (void)hipStreamDestroy(c->st_lane[1]);
    if (c->st_io) (void)hipStreamDestroy(c->st_io);
    delete c;
}
int lf_set_ring_tables(lf_ctx *c, uint64_t nonres, const uint64_t *y) {
    if (!c || !y) return LF_ERR_INVALID;
    if (c->bb) return c->bb->set_ring_tables(nonres, y);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return install_tables(c, nonres, y);
}
int lf_get_ring_tables(lf_ctx *c, uint64_t *nonres, uint64_t *y) {
    if (!c || !nonres || !y) return LF_ERR_INVALID;
    if (c->bb) return c->bb->get_ring_tables(nonres, y);
    *nonres = c->ring.T.nu;
    for (int k = 0; k < 8; k++)
        for (int q = 0; q < 3; q++) y[3 * k + q] = c->ring.T.y[k].c[q];
    return LF_OK;
}
// ---- external coordinate basis: marshalling of one ABI call ---------------------------------------------------------
// With a non-identity basis every entry point below first re-enters itself on converted copies of its NTT-form inputs (external ->
// internal coordinates), converts its outputs back in place, and -- for the prover entry points -- switches the transcript into
// "absorb internal, speak external" mode for the duration of the call.

int lf_set_digit_mode(lf_ctx *c, int mode) {
    if (!c || (mode != 0 && mode != 1)) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    c->digit_mode = mode;
    if (c->bb) c->bb->set_digit_mode(mode);
    return LF_OK;
}
int lf_set_ext_basis(lf_ctx *c, const uint64_t *T) {
    if (!c || !T) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    const int ring = lf_ctx_ring(c);
    return c->xb.set(T, lf_ring_tau(ring), lf_ring_modulus(ring));
}
int lf_set_sharding(lf_ctx *c, int rank, int world, lf_exchange_fn cb, void *user) {
    if (!c || world < 1 || rank < 0 || rank >= world || (world & (world - 1)) != 0 || (world > 1 && !cb)) return LF_ERR_INVALID;
    if (c->bb) return c->bb->set_sharding(rank, world, cb, user);
    std::lock_guard<std::mutex> g(c->mu);
    if (c->A_loaded) return LF_ERR_STATE;  // choose the sharding before loading/generating the Ajtai matrix
    for (int l = 0; l < 2; l++) {
        c->comm[l].destroy();
        c->comm[l].rank = rank; c->comm[l].world = world; c->comm[l].cb = cb; c->comm[l].user = user; c->comm[l].poisoned = false; c->comm[l].model = false;
    }
    c->sh_rank = rank; c->sh_world = world;
    c->two_lanes_ok = false;                 // one channel: one thread issues every exchange
    c->agreed_two_lanes = -1;
    return LF_OK;
}
// Timing model of ONE rank of a sharded run on a box with one GPU (tools/shard_model.py): rank `rank` of `world` with no peers.  Every kernel and every host
// stage does exactly the share of the work that rank would do, every exchange is enqueued in the lane's stream (the peers' words are zeros), the two-lane
// schedule is the one a passed lf_dist_init self-check selects.  What the step returns is NOT a proof (the peers' partial sums are missing).
int lf_set_sharding_model(lf_ctx *c, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world || (world & (world - 1)) != 0) return LF_ERR_INVALID;
    if (c->bb) return LF_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->A_loaded) return LF_ERR_STATE;
    for (int l = 0; l < 2; l++) {
        c->comm[l].destroy();
        c->comm[l].rank = rank; c->comm[l].world = world; c->comm[l].cb = nullptr; c->comm[l].user = nullptr; c->comm[l].poisoned = false; c->comm[l].model = world > 1;
    }
    c->sh_rank = rank; c->sh_world = world;
    c->two_lanes_ok = world > 1;
    c->agreed_two_lanes = -1;
    return LF_OK;
}
int lf_dist_stats_words(lf_ctx *c, uint64_t *words_sent, int reset) {
    if (!c || !words_sent) return LF_ERR_INVALID;
    if (c->bb) { *words_sent = 0; return LF_OK; }
    *words_sent = c->comm[0].words_sent + c->comm[1].words_sent;
    if (reset) c->comm[0].words_sent = c->comm[1].words_sent = 0;
    return LF_OK;
}
// all-gather `words` canonical words from every rank and add them mod p (RCCL has no modular reduction): device buffer, ordered on the lane's stream (RCCL: no host synchronisation; the reduction is k_modsum)
int exchange_modsum_dev(lf_ctx *c, u64 *inout_dev, size_t words) {
    if (c->sh_world <= 1 && !(c->tn.force_exchange && c->cm().nccl)) return LF_OK;   // LF_DIST_FORCE_EXCHANGE: a 1-rank communicator still runs the collectives (RCCL plumbing test on one GPU)
    u64 *g;
    RET(c->tbuf("sh_gather", (size_t)c->sh_world * words, &g));
    RET(c->cm().allgather_dev(inout_dev, g, words, c->stream()));
    launch_modsum(g, (u32)c->sh_world, words, inout_dev, c->stream());
    return LF_OK;
}
int lf_dist_unique_id(uint8_t *id128) { return lfdist::rccl_unique_id(id128); }
// Start-up self-check of the two-lane schedule (lf_dist_init): the FIRST collectives of both communicators are issued concurrently by the two threads that issue
// them in a fold step -- lane 0 by the caller, lane 1 by the helper thread -- each on its lane's stream, four rounds of all-gathers with rank-, lane- and
// round-dependent words, and every word received is checked.  Passed: the step runs the threaded schedule (LF_SHARD_TWO_LANES unset).  Wrong words: the
// communicators stay usable and one host thread issues every exchange (the conservative schedule).  No completion within the time limit
// (LF_DIST_HANDSHAKE_MS, default 20 s): both communicators are aborted and LF_ERR_STATE is returned -- the launcher makes fresh ids and calls lf_dist_init
// again with LF_DIST_NO_HANDSHAKE=1 (latticefold_amd/dist.py does).
static int dist_handshake(lf_ctx *c) {
    c->two_lanes_ok = false;
    if (getenv("LF_DIST_NO_HANDSHAKE")) return LF_OK;
    const int W = c->sh_world, R = c->sh_rank, ITER = 4;
    const size_t words = 64, per_it = words * (size_t)(W + 1);
    const long limit_ms = 20000;
    u64 *dbuf[2] = {nullptr, nullptr}, *hbuf[2] = {nullptr, nullptr};
    for (int l = 0; l < 2; l++) {
        if (lf_dev_malloc(&dbuf[l], per_it * ITER * 8) != hipSuccess || hipHostMalloc((void **)&hbuf[l], per_it * ITER * 8) != hipSuccess) {
            for (int q = 0; q < 2; q++) { if (dbuf[q]) (void)hipFree(dbuf[q]); if (hbuf[q]) (void)hipHostFree(hbuf[q]); }
            return LF_ERR_HIP;
        }
    }
    auto word = [](int g, int lane, int it, size_t w) { return ((u64)(g + 1) * 0x9E3779B97F4A7C15ull) ^ ((u64)lane << 40) ^ ((u64)it << 32) ^ (u64)w; };
    auto run = [&](int lane) -> int {
        const int keep = t_lane;
        t_lane = lane;
        int rc = hipSetDevice(c->device) == hipSuccess ? LF_OK : LF_ERR_HIP;
        for (int it = 0; it < ITER && rc == LF_OK; it++) {
            u64 *hs = hbuf[lane] + per_it * it, *ds = dbuf[lane] + per_it * it;
            for (size_t w = 0; w < words; w++) hs[w] = word(R, lane, it, w);
            if (hipMemcpyAsync(ds, hs, words * 8, hipMemcpyHostToDevice, c->stream()) != hipSuccess) { rc = LF_ERR_HIP; break; }
            rc = c->cm().allgather_dev(ds, ds + words, words, c->stream());
            if (rc == LF_OK && hipMemcpyAsync(hs + words, ds + words, words * W * 8, hipMemcpyDeviceToHost, c->stream()) != hipSuccess) rc = LF_ERR_HIP;
        }
        t_lane = keep;
        return rc;
    };
    c->lane1.submit([&]() -> int { return run(1); });
    int rc = run(0);
    const int rc1 = c->lane1.wait();
    if (rc == LF_OK) rc = rc1;
    bool timed_out = false;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(limit_ms);
    for (int l = 0; l < 2 && rc == LF_OK && !timed_out; l++)
        for (;;) {
            const hipError_t q = hipStreamQuery(c->st_lane[l]);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { rc = LF_ERR_HIP; break; }
            if (std::chrono::steady_clock::now() > deadline) { timed_out = true; break; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    if (timed_out || rc != LF_OK) {      // the collectives may never complete: tear the communicators down (the buffers stay allocated -- a kernel may still hold them)
        c->comm[0].abort_peers(); c->comm[1].abort_peers();
        return LF_ERR_STATE;
    }
    bool ok = true;
    for (int l = 0; l < 2 && ok; l++)
        for (int it = 0; it < ITER && ok; it++)
            for (int g2 = 0; g2 < W && ok; g2++)
                for (size_t w = 0; w < words && ok; w++) ok = hbuf[l][per_it * it + words + (size_t)g2 * words + w] == word(g2, l, it, w);
    // The ranks must AGREE on the schedule: the threaded one splits the exchanges of a step over comm[0] and comm[1], the one-thread schedule issues all of them
    // on comm[0] -- ranks that chose differently would issue different collective sequences and hang.  So the verdict is exchanged inside the library (one more
    // all-gather on comm[0], one issuing thread: the form that works whatever the check found) and the minimum wins; a rank's own LF_SHARD_TWO_LANES=0 / =1 enters
    // it too (2 = forced on, 1 = check passed, 0 = off / failed: forced-on survives only if every rank forced it), so a C or Rust caller of lf_dist_init needs no
    // agreement of its own (the Python launcher's MIN-reduce is no longer what correctness rests on).
    {
        const char *e = getenv("LF_SHARD_TWO_LANES");
        const u64 mine = e ? (atoi(e) != 0 ? 2 : 0) : (ok ? 1 : 0);
        u64 *hs = hbuf[0], *ds = dbuf[0];
        hs[0] = mine;
        const int keep = t_lane;
        t_lane = 0;
        int rc2 = hipMemcpyAsync(ds, hs, 8, hipMemcpyHostToDevice, c->stream()) == hipSuccess ? LF_OK : LF_ERR_HIP;
        if (rc2 == LF_OK) rc2 = c->cm().allgather_dev(ds, ds + 1, 1, c->stream());
        if (rc2 == LF_OK && hipMemcpyAsync(hs + 1, ds + 1, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream()) != hipSuccess) rc2 = LF_ERR_HIP;
        bool late = false;
        for (; rc2 == LF_OK;) {
            const hipError_t q = hipStreamQuery(c->stream());
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { rc2 = LF_ERR_HIP; break; }
            if (std::chrono::steady_clock::now() > deadline) { late = true; break; }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        t_lane = keep;
        if (late || rc2 != LF_OK) { c->comm[0].abort_peers(); c->comm[1].abort_peers(); return LF_ERR_STATE; }
        u64 mn = 2;
        for (int g2 = 0; g2 < W; g2++) mn = hs[1 + g2] < mn ? hs[1 + g2] : mn;
        ok = mn >= 1;                                          // every rank either passed the check or forces the threaded schedule
        c->agreed_two_lanes = ok ? 1 : 0;                      // what the sharded step follows (lf_fold_step), whatever this rank's environment says later
    }
    for (int l = 0; l < 2; l++) { (void)hipFree(dbuf[l]); (void)hipHostFree(hbuf[l]); c->comm[l].n_exchanges = 0; c->comm[l].us_total = 0; c->comm[l].us_max = 0; }
    c->two_lanes_ok = ok;
    return LF_OK;
}
int lf_dist_init(lf_ctx *c, int rank, int world, const uint8_t *ids) {
    if (!c || !ids || world < 1 || rank < 0 || rank >= world || (world & (world - 1)) != 0) return LF_ERR_INVALID;
    if (c->bb) return c->bb->dist_init(rank, world, ids);
    std::lock_guard<std::mutex> g(c->mu);
    if (c->A_loaded) return LF_ERR_STATE;   // choose the sharding before loading/generating the Ajtai matrix
    HIPCHK(hipSetDevice(c->device));
    for (int l = 0; l < 2; l++) {
        c->comm[l].destroy();
        RET(lfdist::rccl_init(c->comm[l], rank, world, ids + 128 * l));
    }
    c->sh_rank = rank; c->sh_world = world;
    return dist_handshake(c);
}
int lf_dist_two_lanes(lf_ctx *c, int set) {
    if (!c) return LF_ERR_INVALID;
    if (c->bb) return 0;                      // (the BabyBear driver exchanges from one thread only)
    std::lock_guard<std::mutex> g(c->mu);
    if (set == 0 || set == 1) { c->two_lanes_ok = set == 1; c->agreed_two_lanes = set; }   // (the caller sets the same value on every rank)
    return c->two_lanes_ok ? 1 : 0;
}
// per-lane callbacks (host transport): the two lanes of a fold step exchange concurrently, so each needs its own ordered channel
int lf_set_sharding_lanes(lf_ctx *c, int rank, int world, lf_exchange_fn cb0, void *user0, lf_exchange_fn cb1, void *user1) {
    int rc = lf_set_sharding(c, rank, world, cb0, user0);
    if (rc != LF_OK || !cb1) return rc;
    if (c->bb) return LF_OK;   // the BabyBear driver exchanges from one thread only
    std::lock_guard<std::mutex> g(c->mu);
    c->comm[1].cb = cb1; c->comm[1].user = user1;
    c->agreed_two_lanes = -1;
    c->two_lanes_ok = (cb1 != cb0 || user1 != user0);   // two ordered channels supplied by the host language: the threaded schedule is the default
    return LF_OK;
}
int lf_dist_stats(lf_ctx *c, uint64_t *n_exchanges, double *total_us, double *max_us, int reset) {
    if (!c) return LF_ERR_INVALID;
    lfdist::Comm *ms[2] = {c->bb ? c->bb->comm() : &c->comm[0], c->bb ? nullptr : &c->comm[1]};
    uint64_t n = 0;
    double tot = 0, mx = 0;
    for (auto *m : ms)
        if (m) {
            n += m->n_exchanges; tot += m->us_total; mx = m->us_max > mx ? m->us_max : mx;
            if (reset) { m->n_exchanges = 0; m->us_total = 0; m->us_max = 0; }
        }
    if (n_exchanges) *n_exchanges = n;
    if (total_us) *total_us = tot;
    if (max_us) *max_us = mx;
    return LF_OK;
}
int lf_mem_info(lf_ctx *c, size_t *free_bytes, size_t *total_bytes) {
    if (!c || !free_bytes || !total_bytes) return LF_ERR_INVALID;
    if (c->bb) return c->bb->mem_info(free_bytes, total_bytes);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemGetInfo(free_bytes, total_bytes));
    return LF_OK;
}
int lf_device_synchronize(lf_ctx *c) {
    if (!c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->synchronize();
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream()));
    return LF_OK;
}

// ---- host<->device staging of AoS ring-element arrays ----------------------------------------------------------
// upload n ring elements (AoS) into a plane table dst [24][n]
int up_ring(lf_ctx *c, const u64 *host, size_t n, u64 *dst) {
    if (!n) return LF_OK;
    u64 *tmp;
    RET(c->tbuf("stage_aos", n * 24, &tmp));
    HIPCHK(hipMemcpyAsync(tmp, host, n * 24 * 8, hipMemcpyHostToDevice, c->stream()));
    launch_aos_to_soa(tmp, dst, n, c->stream());
    return LF_OK;
}
int down_ring(lf_ctx *c, const u64 *src, size_t n, u64 *host) {
    if (!n) return LF_OK;
    u64 *tmp;
    RET(c->tbuf("stage_aos", n * 24, &tmp));
    launch_soa_to_aos(src, tmp, n, c->stream());
    HIPCHK(hipMemcpyAsync(host, tmp, n * 24 * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    return LF_OK;
}
// small device array -> host (through pinned memory)
int down_small(lf_ctx *c, const u64 *dsrc, size_t words, u64 *host) {
    RET(c->pin(words));
    HIPCHK(hipMemcpyAsync(c->h_pin_ref(), dsrc, words * 8, hipMemcpyDeviceToHost, c->stream()));
    RET(c->lane_sync());
    memcpy(host, c->h_pin_ref(), words * 8);
    return LF_OK;
}
Fq3Const f3c(Fq3 a) { Fq3Const r; r.c[0] = a.c[0]; r.c[1] = a.c[1]; r.c[2] = a.c[2]; return r; }

int lf_selftest_field(lf_ctx *c, uint64_t seed, uint32_t n, uint64_t *mismatches) {
    if (!c || !mismatches) return LF_ERR_INVALID;
    if (c->bb) return c->bb->selftest_field(seed, n, mismatches);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *d;
    RET(c->tbuf("small_dev", 4096, &d));
    launch_selftest_field(seed, n, d, c->stream());
    return down_small(c, d, 1, mismatches);
}

// ---- a1/a2 --------------------------------------------------------------------------------------------------------
int lf_ntt_fwd(lf_ctx *c, const uint64_t *in, uint64_t *out, size_t count) {
    if (LF_XB(c)) { XB x(c); int rc = lf_ntt_fwd(c, in, out, count); if (rc == LF_OK) x.ring_out(out, count); return rc; }
    if (!c || (!in && count) || (!out && count)) return LF_ERR_INVALID;
    if (c->bb) return c->bb->ntt_fwd(in, out, count);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *a, *b;
    RET(c->tbuf("io_a", count * 24, &a));
    RET(c->tbuf("io_b", count * 24, &b));
    RET(up_ring(c, in, count, a));
    launch_crt_fwd(c->dcrt, a, b, count, c->stream());
    return down_ring(c, b, count, out);
}
int lf_ntt_inv(lf_ctx *c, const uint64_t *in, uint64_t *out, size_t count) {
    if (LF_XB(c) && in) { XB x(c); return lf_ntt_inv(c, x.ring_in(in, count), out, count); }
    if (!c || (!in && count) || (!out && count)) return LF_ERR_INVALID;
    if (c->bb) return c->bb->ntt_inv(in, out, count);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *a, *b;
    RET(c->tbuf("io_a", count * 24, &a));
    RET(c->tbuf("io_b", count * 24, &b));
    RET(up_ring(c, in, count, a));
    launch_icrt_dense(c->d_icrt, a, b, count, c->stream());
    return down_ring(c, b, count, out);
}
static bool pow2(u64 b) { return b >= 2 && (b & (b - 1)) == 0; }
int lf_decompose(lf_ctx *c, const uint64_t *in, size_t count, uint64_t base, unsigned digits, int layout, uint64_t *out) {
    if (!c || !in || !out || digits == 0 || digits > 64 || (layout != 0 && layout != 1)) return LF_ERR_INVALID;
    if (c->bb) return c->bb->decompose(in, count, base, digits, layout, out);
    if (!pow2(base)) return LF_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *a, *b;
    RET(c->tbuf("io_a", count * 24, &a));
    RET(c->tbuf("io_b", count * digits * 24, &b));
    RET(up_ring(c, in, count, a));
    launch_decompose(a, count, base, digits, layout, b, c->stream(), c->digit_mode);
    if (layout == 0) return down_ring(c, b, count * digits, out);
    for (unsigned k = 0; k < digits; k++) RET(down_ring(c, b + (size_t)k * 24 * count, count, out + (size_t)k * count * 24));
    return LF_OK;
}
int lf_recompose(lf_ctx *c, const uint64_t *in, size_t count_out, uint64_t base, unsigned digits, uint64_t *out) {
    if (!c || !in || !out || digits == 0) return LF_ERR_INVALID;
    if (c->bb) return c->bb->recompose(in, count_out, base, digits, out);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *a, *b;
    RET(c->tbuf("io_a", count_out * digits * 24, &a));
    RET(c->tbuf("io_b", count_out * 24, &b));
    RET(up_ring(c, in, count_out * digits, a));
    launch_recompose(a, count_out, base, digits, b, c->stream());
    return down_ring(c, b, count_out, out);
}
int lf_linf_check(lf_ctx *c, const uint64_t *f_ntt, size_t count, uint64_t bound, int unsigned_variant, int *ok, uint64_t *max_out) {
    if (LF_XB(c) && f_ntt) { XB x(c); return lf_linf_check(c, x.ring_in(f_ntt, count), count, bound, unsigned_variant, ok, max_out); }
    if (!c || !f_ntt || !ok) return LF_ERR_INVALID;
    if (c->bb) return c->bb->linf_check(f_ntt, count, bound, unsigned_variant, ok, max_out);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *a, *b, *mx;
    RET(c->tbuf("io_a", count * 24, &a));
    RET(c->tbuf("io_b", count * 24, &b));
    RET(c->tbuf("small_dev", 4096, &mx));
    RET(up_ring(c, f_ntt, count, a));
    launch_icrt_dense(c->d_icrt, a, b, count, c->stream());
    if (unsigned_variant) {
        // literal Witness::within_bound: canonical coefficient < bound  <=>  max canonical < bound; reuse the
        // centred kernel on a table where "negative" values are impossible: compare canonical values on host
        // through a max reduction of min(v, p-1-v)?  Not equivalent -- do it exactly: download max canonical.
        std::vector<u64> h(count * 24);
        HIPCHK(hipMemcpyAsync(h.data(), b, count * 24 * 8, hipMemcpyDeviceToHost, c->stream()));
        HIPCHK(hipStreamSynchronize(c->stream()));
        u64 m = 0;
        for (u64 v : h) m = v > m ? v : m;
        if (max_out) *max_out = m;
        *ok = m < bound;
        return LF_OK;
    }
    launch_linf(b, count, mx, c->stream());
    u64 m = 0;
    RET(down_small(c, mx, 1, &m));
    if (max_out) *max_out = m;
    *ok = m < bound;
    return LF_OK;
}

// ---- a5 -----------------------------------------------------------------------------------------------------------
static int shard_columns(lf_ctx *c, size_t n, size_t *col0, size_t *cnt) {
    if (n % (size_t)c->sh_world) return LF_ERR_UNSUPPORTED;
    *cnt = n / c->sh_world;
    *col0 = *cnt * c->sh_rank;
    return LF_OK;
}
// A lives on the device in ONE form: coefficient form, cut into bytes, in MFMA operand order (lf_ajtai_i8.hip) -- what the digit-plane commitments of a fold
// step (k_ajtai_i8s) and the general commitments (lf_ajtai_i8g.hip: commit_ntt, Witness::commit) both stream.  Built once per matrix: rows arrive one at a time
// in NTT form (row_ntt [24][nA] on the device), one fused pass -- inverse CRT map + byte packing -- per row, so the context never holds more than one u64 row.
// (Rounds 2-5 also kept the NTT form, 4.9 GiB at C4, for a 64-bit VALU commit kernel; the int8 general commit retired both.)
static int prep_ajtai_i8_begin(lf_ctx *c) {
    if (c->dAb) { (void)hipFree(c->dAb); c->dAb = nullptr; }
    c->i8_nch = 0;
    const AjtaiI8Ring R = ajtai_i8_goldilocks();
    const u32 maxr = ajtai_i8_max_rows(R), nch = (c->kappa + maxr - 1) / maxr, kc = (c->kappa + nch - 1) / nch;
    const size_t ntiles = (c->nA + 7) / 8;
    const u32 MT = ajtai_i8_row_tiles(R, kc);
    const size_t chunk_bytes = ntiles * (R.RD / 8) * MT * 1024;
    HIPCHK(lf_dev_malloc(&c->dAb, chunk_bytes * nch + ajtai_i8_slack_bytes()));
    HIPCHK(hipMemsetAsync(c->dAb, 0, chunk_bytes * nch + ajtai_i8_slack_bytes(), c->stream()));
    c->i8_nch = nch;
    c->i8_kc = kc;
    return LF_OK;
}
static void prep_ajtai_i8_row(lf_ctx *c, u32 i, const u64 *row_ntt) {
    const AjtaiI8Ring R = ajtai_i8_goldilocks();
    const u32 kc = c->i8_kc, MT = ajtai_i8_row_tiles(R, kc);
    const size_t chunk_bytes = (c->nA + 7) / 8 * (R.RD / 8) * MT * 1024;
    launch_ajtai_icrt_pack_i8(c->d_icrt, row_ntt, c->nA, i % kc, MT, c->dAb + (size_t)(i / kc) * chunk_bytes, c->stream());
}
static int ajtai_install(lf_ctx *c, size_t kappa, size_t n, const uint64_t *A_host, uint64_t seed) {
    size_t col0, cnt;
    RET(shard_columns(c, n, &col0, &cnt));   // a sharded rank keeps only its column slice of the caller's matrix
    c->A_loaded = false;
    c->kappa = (u32)kappa;
    c->nA = cnt; c->nA_total = n; c->A_col0 = col0;
    RET(prep_ajtai_i8_begin(c));
    u64 *row = nullptr;
    RET(c->tbuf("i8_prep_row", 24 * cnt, &row));
    for (size_t i = 0; i < kappa; i++) {
        if (A_host) RET(up_ring(c, A_host + (i * n + col0) * 24, cnt, row));
        else launch_fill_ajtai(row, 1, cnt, n, col0, seed, c->stream(), (u32)i);
        prep_ajtai_i8_row(c, (u32)i, row);
    }
    HIPCHK(hipStreamSynchronize(c->stream()));
    c->drop_buf("i8_prep_row");
    c->drop_buf("stage_aos");
    c->A_loaded = true;
    return LF_OK;
}
// digit planes k0 .. k0+NP-1 of `planes` (this rank's column slice) -> out_dev [NP][kappa][24] NTT form (PARTIAL when sharded)
// wit (optional): the witness `planes` belong to -- if its bit-plane form is at hand (built at the start of the fold step for the GEMM rounds) the
// kernel cuts the digits from it
int commit_planes_i8(lf_ctx *c, const int32_t *planes, size_t ld, u32 k0, u32 NP, u64 *out_dev, const lf_witness *wit) {
    const AjtaiI8Ring R = ajtai_i8_goldilocks();
    const u32 nch = c->i8_nch, kc = c->i8_kc, MT = ajtai_i8_row_tiles(R, kc), maxp = ajtai_i8_max_planes(R);
    const size_t ntiles = (c->nA + 7) / 8, chunk_bytes = ntiles * (R.RD / 8) * MT * 1024;
    // One persistent workgroup per CU fills its LDS (157 KB): on a fully occupied chip the latency-bound round kernels of the other lane
    // cannot be placed until a commit workgroup retires.  7/8 of the CUs (28 of 32 per XCD) leaves them room: C4 26.1 -> 25.0 ms/step
    // (measured 256 / 240 / 224 / 192 / 160 / 128 workgroups: 26.1 / 26.3 / 25.0 / 25.1 / 26.1 / 28.2 ms).
    u32 nwg = c->tn.i8_wgs > 0 ? (u32)c->tn.i8_wgs : 224;
    if (nwg > ntiles) nwg = (u32)ntiles;
    const u32 nslots = nwg < 16 ? 16 : nwg;    // (two plane groups run as 2 x 8 chunks at least: launch_ajtai_i8)
    int32_t *part, *dsum;
    long long *sum;
    u64 *coef, *ntt;
    const u32 NTmax = ajtai_i8_col_tiles(R, maxp);
    RET(c->tbuf("i8_part", ajtai_i8_part_words(nslots, MT, NTmax), &part));
    RET(c->tbuf("i8_dsum", (size_t)nslots * maxp * R.RD, &dsum));
    RET(c->tbuf("i8_sum", ajtai_i8_sum_words(R, MT, NTmax, maxp), &sum));
    const size_t side_words = (size_t)24 * NP * c->kappa;
    RET(c->tbuf("i8_coef", side_words, &coef));
    RET(c->tbuf("i8_ntt", side_words, &ntt));
    const u32 *bits = nullptr;
    if (wit && c->A_col0 == 0 && planes == wit->planes && c->nA == c->N)
        for (int sd = 0; sd < 2; sd++)
            if (c->bits_wit[sd] == wit && c->bits_ptr[sd]) {
                bits = c->bits_ptr[sd];
                if (c->stream() != c->st_lane[1]) HIPCHK(hipStreamWaitEvent(c->stream(), c->bits_ev[sd], 0));
                break;
            }
    const size_t bits_nw = (c->N + 511) / 512 * 16;          // words per row of the bit-plane form (positions padded to 512)
    const u32 bits_rows = 16 * ((c->P.K + 15) / 16) + 1;
    for (u32 p0 = 0; p0 < NP; p0 += maxp) {
        const u32 np = NP - p0 < maxp ? NP - p0 : maxp;
        u64 *cf = coef + (size_t)24 * p0 * c->kappa;   // SoA block of this plane group: [24][np*kappa]
        for (u32 ch = 0; ch < nch; ch++) {
            const u32 row0 = ch * kc, kn = c->kappa - row0 < kc ? c->kappa - row0 : kc;
            size_t ev = c->ev_begin(1);
            int g = launch_ajtai_i8(R, c->dAb + (size_t)ch * chunk_bytes, MT, planes, ld, c->nA, kn, row0, c->kappa, k0 + p0, np, nwg, part, dsum, sum, cf, c->stream(),
                                    bits, bits_nw, bits_rows);
            c->ev_end(ev);
            if (g < 0) return LF_ERR_UNSUPPORTED;
        }
        const size_t ne = (size_t)np * c->kappa;
        launch_crt_fwd(c->dcrt, cf, ntt, ne, c->stream());
        launch_soa_to_aos(ntt, out_dev + (size_t)p0 * c->kappa * 24, ne, c->stream());
    }
    return LF_OK;
}
int lf_ajtai_load(lf_ctx *c, const uint64_t *A, size_t kappa, size_t n) {
    if (LF_XB(c) && A && kappa <= 128) { XB x(c); return lf_ajtai_load(c, x.ring_in(A, kappa * n), kappa, n); }
    if (!c || !A || !kappa || !n || kappa > 128) return LF_ERR_INVALID;
    if (c->bb) return c->bb->ajtai_load(A, kappa, n);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return ajtai_install(c, kappa, n, A, 0);
}
int lf_ajtai_generate(lf_ctx *c, uint64_t seed, size_t kappa, size_t n) {
    if (!c || !kappa || !n || kappa > 128) return LF_ERR_INVALID;
    if (c->bb) return c->bb->ajtai_generate(seed, kappa, n);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    return ajtai_install(c, kappa, n, nullptr, seed);
}
int lf_device_memory(lf_ctx *c, size_t *free_bytes, size_t *total_bytes) {
    if (!c || !free_bytes || !total_bytes) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemGetInfo(free_bytes, total_bytes));
    return LF_OK;
}
// General commitments from the resident byte planes of A (lf_ajtai_i8g.hip): AjtaiCommitmentScheme::commit_ntt (commitment_scheme.rs:37-54,75-77) for
// `batch` vectors F [batch][24][ldF] in NTT form (pointing at this rank's first column), or Witness::commit (arith.rs:357-362) for the centred int32
// coefficient planes of a witness handle (F null, batch 1).  out_dev: [batch][kappa][24] NTT form, AoS (PARTIAL when sharded).
static int commit_dev_i8g(lf_ctx *c, const u64 *F, size_t ldF, u32 batch, const int32_t *planes, size_t ldp, u64 *out_dev, bool timed) {
    if (!c->A_loaded || !c->i8_nch || !c->dAb) return LF_ERR_STATE;
    const AjtaiI8Ring R = ajtai_i8_goldilocks();
    const u32 nch = c->i8_nch, kc = c->i8_kc, MT = ajtai_i8_row_tiles(R, kc);
    const size_t ntiles = (c->nA + 7) / 8, chunk_bytes = ntiles * (R.RD / 8) * MT * 1024;
    const u32 NP = planes ? ajtai_i8g_planes_i32() : ajtai_i8g_planes_general(R);
    const char *e_wgs = getenv("LF_I8G_WGS");           // (test hook: workgroups of the general commit kernel; default one per CU)
    const u32 nwg = e_wgs && atoi(e_wgs) > 0 ? (u32)atoi(e_wgs) : 256;
    size_t pw, dw, sw;
    if (ajtai_i8g_scratch(R, MT, c->nA, NP, nwg, &pw, &dw, &sw) != 0) return LF_ERR_UNSUPPORTED;
    unsigned long long *pre;
    int32_t *part, *dsum;
    long long *sum;
    u64 *coef, *ntt;
    RET(c->tbuf("i8g_pre", (size_t)NP * 24 * ntiles, &pre));
    RET(c->tbuf("i8g_part", pw, &part));
    RET(c->tbuf("i8g_dsum", dw, &dsum));
    RET(c->tbuf("i8g_sum", sw, &sum));
    RET(c->tbuf("i8g_coef", (size_t)24 * c->kappa, &coef));
    RET(c->tbuf("i8g_ntt", (size_t)24 * c->kappa, &ntt));
    for (u32 b = 0; b < batch; b++) {
        const size_t ev = timed ? c->ev_begin(1) : 0;   // the whole device side of one commitment: digit pass, contraction, recombination, CRT
        if (planes) launch_i8g_cut_i32(planes, ldp, c->nA, 24, NP, pre, ntiles, c->stream());
        else launch_i8g_cut_ntt(c->d_icrt, c->d_icrt_sp_val, c->d_icrt_sp_col, F + (size_t)b * 24 * ldF, ldF, c->nA, NP, pre, ntiles, c->stream());
        for (u32 ch = 0; ch < nch; ch++) {
            const u32 row0 = ch * kc, kn = c->kappa - row0 < kc ? c->kappa - row0 : kc;
            const int g = launch_ajtai_i8g(R, c->dAb + (size_t)ch * chunk_bytes, MT, pre, ntiles, c->nA, kn, row0, c->kappa, NP, nwg, part, dsum, sum, coef, c->stream());
            if (g < 0) return LF_ERR_UNSUPPORTED;
        }
        launch_crt_fwd(c->dcrt, coef, ntt, c->kappa, c->stream());
        launch_soa_to_aos(ntt, out_dev + (size_t)b * c->kappa * 24, c->kappa, c->stream());
        if (timed) c->ev_end(ev);
    }
    return LF_OK;
}
// F: [batch][24][ldF] device, pointing at this rank's first column; out_dev: [batch][kappa][24] device AoS (PARTIAL when sharded)
static int commit_dev(lf_ctx *c, const u64 *F, size_t ldF, u32 batch, u64 *out_dev, bool timed) { return commit_dev_i8g(c, F, ldF, batch, nullptr, 0, out_dev, timed); }
// download a (partial) commitment and, when sharded, all-gather + add the partials mod p
int commit_download(lf_ctx *c, const u64 *dev, size_t words, u64 *host) {
    RET(exchange_modsum_dev(c, (u64 *)dev, words));   // sharded: ncclAllGather of the partial commitments + k_modsum, in stream
    return down_small(c, dev, words, host);
}
// index slice of this rank: [*i0, *i0 + *cnt) of n items (the last rank takes the remainder)
void shard_slice(const lf_ctx *c, size_t n, size_t *i0, size_t *cnt) {
    size_t per = (n + (size_t)c->sh_world - 1) / (size_t)c->sh_world;
    size_t lo = per * (size_t)c->sh_rank;
    if (lo > n) lo = n;
    *i0 = lo;
    *cnt = lo + per > n ? n - lo : per;
}
// Sharded sumchecks: tables of `n` entries stay sharded while every rank keeps at least 64 pairs AND the tables are larger than the hand-over size of the
// sumcheck (kind 0 linearization, 1 folding; Tunables::shard_lin_min / shard_fold_min, never above m / 16 so that small instances still exercise the sharded
// rounds).  Every rank evaluates the same predicate on the same numbers: the ranks leave the sharded form in the same round.
bool shard_keep(const lf_ctx *c, int kind, size_t n) {
    const size_t Gw = (size_t)c->sh_world;
    if (Gw <= 1 || n / 2 < Gw * 64) return false;
    size_t thr = kind ? c->tn.shard_fold_min : c->tn.shard_lin_min;
    if (thr > (c->m >> 4)) thr = c->m >> 4;
    return n > thr;
}
// all-gather the ranks' column slices of `planes` tables stored with GLOBAL layout [plane][n] (rank g holds entries
// [g*n/G, (g+1)*n/G) of every plane) and fill in the others' slices
int gather_slices(lf_ctx *c, u64 *buf, size_t planes, size_t n) {
    const size_t Gw = (size_t)c->sh_world, lcl = n / Gw, words = planes * lcl;
    u64 *gall, *gtmp;
    RET(c->tbuf("sh_gather_tab", words * Gw, &gall));
    RET(c->tbuf("sh_gather_tmp", words, &gtmp));
    HIPCHK(hipMemcpy2DAsync(gtmp, lcl * 8, buf + (size_t)c->sh_rank * lcl, n * 8, lcl * 8, planes, hipMemcpyDeviceToDevice, c->stream()));
    RET(c->cm().allgather_dev(gtmp, gall, words, c->stream()));
    launch_gather_relayout(gall, (u32)Gw, planes, lcl, buf, c->stream());
    return LF_OK;
}
// Several table sets in ONE exchange (the hand-over of a sharded sumcheck to its replicated rounds): part i is this rank's `lcl` entries of `planes` rows
// at src (row stride src_ld) and becomes the full tables dst [planes][G lcl] on every rank.  src may lie inside dst (the payload is staged first).
int gather_parts(lf_ctx *c, const GatherPart *parts, int np, size_t lcl) {
    const size_t Gw = (size_t)c->sh_world;
    size_t ptot = 0;
    for (int i = 0; i < np; i++) ptot += parts[i].planes;
    const size_t words = ptot * lcl;
    u64 *gall, *gtmp;
    RET(c->tbuf("sh_gather_tab", words * Gw, &gall));
    RET(c->tbuf("sh_gather_tmp", words, &gtmp));
    size_t p0 = 0;
    for (int i = 0; i < np; i++) {
        HIPCHK(hipMemcpy2DAsync(gtmp + p0 * lcl, lcl * 8, parts[i].src, parts[i].src_ld * 8, lcl * 8, parts[i].planes, hipMemcpyDeviceToDevice, c->stream()));
        p0 += parts[i].planes;
    }
    RET(c->cm().allgather_dev(gtmp, gall, words, c->stream()));
    p0 = 0;
    for (int i = 0; i < np; i++) {
        launch_gather_relayout_part(gall, (u32)Gw, ptot, p0, parts[i].planes, lcl, parts[i].dst, c->stream());
        p0 += parts[i].planes;
    }
    return LF_OK;
}
int lf_ajtai_commit(lf_ctx *c, const uint64_t *f, size_t n, size_t batch, uint64_t *out) {
    if (LF_XB(c) && f && out) {
        XB x(c);
        u32 kap = c->bb ? 0 : c->kappa;
        int rc = lf_ajtai_commit(c, x.ring_in(f, n * batch), n, batch, out);
        if (rc == LF_OK) x.ring_out(out, batch * (c->bb ? c->bb->kappa() : kap));
        return rc;
    }
    if (!c || !f || !out || !batch) return LF_ERR_INVALID;
    if (c->bb) return c->bb->ajtai_commit(f, n, batch, out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->A_loaded) return LF_ERR_STATE;
    if (n != c->nA_total) return LF_ERR_INVALID;  // CommitmentError::WrongWitnessLength(n, width)
    HIPCHK(hipSetDevice(c->device));
    u64 *F, *o;
    RET(c->tbuf("io_a", batch * n * 24, &F));
    RET(c->tbuf("io_b", batch * c->kappa * 24, &o));
    for (size_t b = 0; b < batch; b++) RET(up_ring(c, f + b * n * 24, n, F + b * 24 * n));
    c->tn = Tunables::read((size_t)1 << 14);
    c->ev_reset();
    RET(commit_dev(c, F + c->A_col0, n, (u32)batch, o, true));   // timed: lf_last_kernel_stats reports the stand-alone kernel
    c->ev_collect();
    return commit_download(c, o, batch * c->kappa * 24, out);
}

// column-sharded commit: the context holds only columns [col0, col0+n_local) of A (loaded with lf_ajtai_load on
// that slice); f is the matching slice of each witness.  The result is the PARTIAL commitment of this shard; the caller
// exchanges partials (all-gather) and adds them mod p -- lf_modsum -- because RCCL has no modular reduction.
int lf_modsum(const uint64_t *parts, size_t nparts, size_t words, uint64_t *out) {
    if (!parts || !out || !nparts) return LF_ERR_INVALID;
    for (size_t w = 0; w < words; w++) {
        u64 acc = 0;
        for (size_t g = 0; g < nparts; g++) {
            u64 v = parts[g * words + w];
            if (v >= LF_P) return LF_ERR_INVALID;
            acc = fq_add(acc, v);
        }
        out[w] = acc;
    }
    return LF_OK;
}
int lf_modsum_ring(const uint64_t *parts, size_t nparts, size_t words, uint64_t *out, int ring) {
    if (ring == LF_RING_GOLDILOCKS) return lf_modsum(parts, nparts, words, out);
    if (ring != LF_RING_BABYBEAR || !parts || !out || !nparts) return LF_ERR_INVALID;
    for (size_t w = 0; w < words; w++) {
        u64 acc = 0;
        for (size_t g = 0; g < nparts; g++) {
            u64 v = parts[g * words + w];
            if (v >= lfbb::BB_P) return LF_ERR_INVALID;
            acc += v;                         // nparts * p < 2^64 for any realistic rank count
        }
        out[w] = acc % lfbb::BB_P;
    }
    return LF_OK;
}

// ---- a8/a9/a11 ------------------------------------------------------------------------------------------------------
int build_eq_dev(lf_ctx *c, const Fq3 *pt, u32 nv, u64 *eq_dev) {
    Fq3Const *rd;
    RET(c->tbuf("eq_point", 64, &rd));
    std::vector<Fq3Const> h(nv);
    for (u32 i = 0; i < nv; i++) h[i] = f3c(pt[i]);
    RET(c->h2d_small(rd, h.data(), nv * sizeof(Fq3Const)));
    if (nv >= 6) {   // two-level: one product per entry
        u64 *scr;
        RET(c->tbuf("eq_scratch", build_eq_scratch_words(nv), &scr));
        launch_build_eq2(c->dcrt, rd, nv, scr, eq_dev, c->stream());
    } else launch_build_eq(c->dcrt, rd, nv, eq_dev, c->stream());
    return LF_OK;
}
int lf_build_eq(lf_ctx *c, const uint64_t *point, unsigned nv, uint64_t *out) {
    if (LF_XB(c) && point && out && nv && nv <= 40) { XB x(c); int rc = lf_build_eq(c, x.ext_in(point, nv), nv, out); if (rc == LF_OK) x.ext_out(out, (size_t)1 << nv); return rc; }
    if (!c || !point || !out || nv == 0 || nv > 40) return LF_ERR_INVALID;
    if (c->bb) return c->bb->build_eq(point, nv, out);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    size_t n = (size_t)1 << nv;
    u64 *eq;
    RET(c->tbuf("io_a", 3 * n, &eq));
    std::vector<Fq3> pt(nv);
    for (unsigned i = 0; i < nv; i++) pt[i] = fq3_make(point[3 * i], point[3 * i + 1], point[3 * i + 2]);
    RET(build_eq_dev(c, pt.data(), nv, eq));
    std::vector<u64> h(3 * n);
    HIPCHK(hipMemcpyAsync(h.data(), eq, 3 * n * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    for (size_t i = 0; i < n; i++)
        for (int q = 0; q < 3; q++) out[3 * i + q] = h[(size_t)q * n + i];
    return LF_OK;
}
int lf_mle_eval_batch(lf_ctx *c, const uint64_t *tables, size_t ntables, size_t len, const uint64_t *point, unsigned nv, uint64_t *out) {
    if (LF_XB(c) && tables && point && out) {
        XB x(c);
        int rc = lf_mle_eval_batch(c, x.ring_in(tables, ntables * len), ntables, len, x.ext_in(point, nv), nv, out);
        if (rc == LF_OK) x.ring_out(out, ntables);
        return rc;
    }
    if (!c || !tables || !point || !out || !ntables || nv == 0 || nv > 40) return LF_ERR_INVALID;
    if (c->bb) return c->bb->mle_eval_batch(tables, ntables, len, point, nv, out);
    size_t n = (size_t)1 << nv;
    if (len > n || len == 0) return LF_ERR_INVALID;  // MleEvaluationError::IncorrectLength
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *eq, *X, *partial, *o;
    RET(c->tbuf("io_eq", 3 * n, &eq));
    RET(c->tbuf("io_a", ntables * len * 24, &X));
    RET(c->tbuf("red_partial", 256 * (ntables * 24 > 4096 ? ntables * 24 : 4096), &partial));
    RET(c->tbuf("io_b", ntables * 24, &o));
    std::vector<Fq3> pt(nv);
    for (unsigned i = 0; i < nv; i++) pt[i] = fq3_make(point[3 * i], point[3 * i + 1], point[3 * i + 2]);
    RET(build_eq_dev(c, pt.data(), nv, eq));
    for (size_t a = 0; a < ntables; a++) RET(up_ring(c, tables + a * len * 24, len, X + a * 24 * len));
    launch_dot_eq(c->dcrt, X, len, (u32)ntables, eq, n, len, partial, o, c->stream());
    return down_small(c, o, ntables * 24, out);
}

// ---- CCS -------------------------------------------------------------------------------------------------------------
size_t lf_lcccs_len_ring(const lf_params *p, int ring) { return ring == LF_RING_BABYBEAR ? lfbb::bb_lcccs_len(p) : lf_lcccs_len(p); }
size_t lf_cccs_len_ring(const lf_params *p, int ring) { return ring == LF_RING_BABYBEAR ? lfbb::bb_cccs_len(p) : lf_cccs_len(p); }
size_t lf_proof_len_ring(const lf_params *p, int ring) { return ring == LF_RING_BABYBEAR ? lfbb::bb_proof_len(p) : lf_proof_len(p); }
size_t lf_lcccs_len(const lf_params *p) { return (size_t)p->s + 3 + p->kappa + p->t + p->l + 1; }
size_t lf_cccs_len(const lf_params *p) { return (size_t)p->kappa + p->l; }
size_t lin_proof_len(const lf_params *p) { return (size_t)p->s * (p->d + 2) + 3 + p->t; }
size_t dec_proof_len(const lf_params *p) { return (size_t)p->K * (p->t + 3 + p->l + 1 + p->kappa); }
static size_t fold_proof_len(const lf_params *p) { return (size_t)p->s * (2 * p->b + 1) + 2 * (size_t)p->K * (3 + p->t); }
size_t lf_proof_len(const lf_params *p) { return lin_proof_len(p) + 2 * dec_proof_len(p) + fold_proof_len(p); }

int lf_ccs_load(lf_ctx *c, const lf_params *p, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val,
                const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *cc) {
    if (LF_XB(c) && p && rowptr && col && val && S_off && S_idx && cc && p->t >= 1 && p->t <= 4 && p->s <= 30 && p->q <= 8) {
        XB x(c);
        const size_t m = (size_t)1 << p->s;
        const uint64_t *v2[4];
        for (u32 j = 0; j < p->t; j++) {
            if (!rowptr[j] || !val[j]) return LF_ERR_INVALID;
            v2[j] = x.ring_in(val[j], rowptr[j][m]);
        }
        return lf_ccs_load(c, p, rowptr, col, v2, S_off, S_idx, x.ring_in(cc, p->q));
    }
    if (!c || !p || !rowptr || !col || !val || !S_off || !S_idx || !cc) return LF_ERR_INVALID;
    if (c->bb) return c->bb->ccs_load(p, rowptr, col, val, S_off, S_idx, cc);
    if (p->s < 3 || p->s > 30 || p->t == 0 || p->t > 4 || p->q == 0 || p->q > 8 || p->K == 0 || p->K > 32 || p->L == 0 || p->L > 8 ||
        p->d + 1 > 4 || p->wit_len == 0)
        return LF_ERR_UNSUPPORTED;
    if (p->b != 2) return LF_ERR_UNSUPPORTED;  // folding comb is specialised to b = 2 (all reference Goldilocks rows)
    // B = 2^32 (config.toml:158): balanced digits lie in [-2^31, 2^31]; the int32 planes hold all of them but +2^31 exactly, which the ingest
    // rejects (LF_ERR_UNSUPPORTED) -- one value in 2^32 per digit
    if (!pow2(p->B) || p->B > (1ULL << 32)) return LF_ERR_UNSUPPORTED;
    {   // K base-2 digits must cover |coeff| <= B/2
        u64 half = p->B / 2;
        u32 need = 0;
        while ((half >> need) != 0) need++;
        if (need > p->K) return LF_ERR_UNSUPPORTED;
    }
    size_t m = (size_t)1 << p->s, N = (size_t)p->wit_len * p->L, n = (size_t)p->l + 1 + p->wit_len;
    if (N > m) return LF_ERR_SIZE_BOUNDS;  // sanity_check, nifs.rs:165-173
    // the reference indexes comb values by matrix index: multisets must concatenate to 0..t-1
    {
        u32 next = 0;
        for (u32 i = 0; i < p->q; i++)
            for (u32 k = S_off[i]; k < S_off[i + 1]; k++)
                if (S_idx[k] != next++) return LF_ERR_UNSUPPORTED;
        if (next != p->t || S_off[p->q] > 16) return LF_ERR_UNSUPPORTED;
    }
    RET(lf_validate_csr(p->t, m, n, rowptr, col, val, 24, LF_P));   // before any context state is touched
    for (size_t k = 0; k < (size_t)p->q * 24; k++)
        if (cc[k] >= LF_P) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    free_ccs(c);
    c->P = *p; c->N = N; c->m = m; c->n = n;
    memset(&c->desc, 0, sizeof(c->desc));
    c->desc.t = p->t; c->desc.q = p->q;
    for (u32 i = 0; i <= p->q; i++) c->desc.S_off[i] = S_off[i];
    for (u32 k = 0; k < S_off[p->q]; k++) c->desc.S_idx[k] = S_idx[k];
    for (u32 i = 0; i < p->q; i++)
        for (u32 k = S_off[i]; k < S_off[i + 1]; k++) { c->desc.ms[k] = i; c->desc.first[k] = (k == S_off[i]); }
    for (u32 i = 0; i < p->q; i++) {
        memcpy(c->desc.c[i], cc + (size_t)i * 24, 24 * 8);
        u64 one[24], mone[24];
        HostRing::from_u64(1, one);
        HostRing::from_u64(LF_P - 1, mone);
        c->desc.c_unit[i] = !memcmp(c->desc.c[i], one, sizeof(one)) ? 1 : (!memcmp(c->desc.c[i], mone, sizeof(mone)) ? -1 : 0);
    }
    // every device array is registered in the context as soon as it exists, so a failure half-way leaks nothing (free_ccs frees them)
    auto dalloc = [](auto &vec, size_t bytes) -> void * {
        void *ptr = nullptr;
        if (lf_dev_malloc(&ptr, bytes) != hipSuccess) return nullptr;
        vec.push_back((typename std::remove_reference<decltype(vec)>::type::value_type)ptr);
        return ptr;
    };
    for (u32 j = 0; j < p->t; j++) {
        size_t nnz = rowptr[j][m];
        void *drp = dalloc(c->d_rowptr, (m + 1) * 4), *dci = dalloc(c->d_col, (nnz + 1) * 4), *dv = dalloc(c->d_val, (nnz + 1) * 24 * 8);
        void *dcp = dalloc(c->d_colptr, (n + 1) * 4), *dri = dalloc(c->d_rowidx, (nnz + 1) * 4), *dvT = dalloc(c->d_valT, (nnz + 1) * 24 * 8);
        if (!drp || !dci || !dv || !dcp || !dri || !dvT) return LF_ERR_HIP;
        HIPCHK(hipMemcpy(drp, rowptr[j], (m + 1) * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dci, col[j], nnz * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dv, val[j], nnz * 24 * 8, hipMemcpyHostToDevice));
        // CSC
        std::vector<u32> cp(n + 1, 0), ri(nnz);
        std::vector<u64> vT(nnz * 24);
        for (size_t k = 0; k < nnz; k++) cp[col[j][k] + 1]++;
        for (size_t i = 0; i < n; i++) cp[i + 1] += cp[i];
        std::vector<u32> fill(cp.begin(), cp.end() - 1);
        for (size_t r = 0; r < m; r++)
            for (u32 k = rowptr[j][r]; k < rowptr[j][r + 1]; k++) {
                u32 pos = fill[col[j][k]]++;
                ri[pos] = (u32)r;
                memcpy(&vT[(size_t)pos * 24], val[j] + (size_t)k * 24, 24 * 8);
            }
        HIPCHK(hipMemcpy(dcp, cp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dri, ri.data(), nnz * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dvT, vT.data(), nnz * 24 * 8, hipMemcpyHostToDevice));
    }
    {
        const size_t rows_used = n < m ? n : m;
        c->ccs_general = false;
        for (u32 jj = 0; jj < p->t; jj++)
            if ((size_t)rowptr[jj][m] * 2 > rows_used * 3) c->ccs_general = true;
    }
    c->have_ccs = true;
    c->shc_r0 = (size_t)-1;
    return LF_OK;
}
int lf_spmv(lf_ctx *c, unsigned j, const uint64_t *z, uint64_t *out) {
    if (LF_XB(c) && z && out && c->have_ccs_any()) { XB x(c); int rc = lf_spmv(c, j, x.ring_in(z, c->n_any()), out); if (rc == LF_OK) x.ring_out(out, c->m_any()); return rc; }
    if (!c || !z || !out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->spmv(j, z, out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    if (j >= c->P.t) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    u64 *zd, *od;
    RET(c->tbuf("io_a", c->n * 24, &zd));
    RET(c->tbuf("io_b", c->m * 24, &od));
    RET(up_ring(c, z, c->n, zd));
    if (c->ccs_general) {
        u64 *zaos;
        RET(c->tbuf("spmv_zaos", c->n * 24, &zaos));
        launch_spmv_rows(c->dcrt, 1, &c->d_rowptr[j], &c->d_col[j], &c->d_val[j], zd, 0, c->n, zaos, od, c->m, 0, c->stream());
    } else
    launch_spmv(c->dcrt, c->d_rowptr[j], c->d_col[j], c->d_val[j], zd, c->n, od, c->m, 0, c->stream());
    return down_ring(c, od, c->m, out);
}

// ---- witnesses ---------------------------------------------------------------------------------------------------------
static int witness_from_coef_table(lf_ctx *c, const u64 *coef_dev /* [24][N] canonical */, lf_witness **out) {
    int32_t *pl;
    HIPCHK(lf_dev_malloc(&pl, c->N * 24 * 4));
    int *viol;
    if (c->tbuf("small_dev", 4096, (u64 **)&viol) != LF_OK) { (void)hipFree(pl); return LF_ERR_HIP; }
    (void)hipMemsetAsync(viol, 0, 4, c->stream());
    launch_coef_to_i32(coef_dev, pl, c->N, (u32)(c->P.B / 2), viol, c->stream());
    int hv = 0;
    if (hipMemcpyAsync(&hv, viol, 4, hipMemcpyDeviceToHost, c->stream()) != hipSuccess || hipStreamSynchronize(c->stream()) != hipSuccess) {
        (void)hipFree(pl);
        return LF_ERR_HIP;
    }
    if (hv) { (void)hipFree(pl); return (hv & 1) ? LF_ERR_NORM : LF_ERR_UNSUPPORTED; }
    lf_witness *w = new lf_witness{c, pl, c->N, c->device, c->N * 24 * 4};
    *out = w;
    return LF_OK;
}
// Witness::from_w_ccs, arith.rs:230-248: ICRT -> gadget_decompose(B, L); on the calling thread's lane (its stream, its buffers)
static int witness_from_w_ccs_lane(lf_ctx *c, const uint64_t *w_ccs, lf_witness **out) {
    u64 *a, *b, *d;
    RET(c->tbuf("io_a", (size_t)c->P.wit_len * 24, &a));
    RET(c->tbuf("io_b", (size_t)c->P.wit_len * 24, &b));
    RET(c->tbuf("io_c", c->N * 24, &d));
    RET(up_ring(c, w_ccs, c->P.wit_len, a));
    launch_icrt_dense(c->d_icrt, a, b, c->P.wit_len, c->stream());
    launch_decompose(b, c->P.wit_len, c->P.B, c->P.L, 0, d, c->stream(), c->digit_mode);
    return witness_from_coef_table(c, d, out);
}
// ---- ingestion next to a running fold step (a chain's next witness: upload over PCIe, ICRT and gadget decomposition on the lowest-priority stream while the
// step before it folds).  Goldilocks contexts in the default basis run it on lane 2 (own stream, own buffers, c->io_mu instead of c->mu: the constraint system
// must not be reloaded meanwhile); every other configuration runs the blocking call on the worker thread -- the same witness, no overlap promised.
struct lf_witness_job {
    std::future<int> fut;
    lf_witness *w = nullptr;
};
int lf_witness_from_w_ccs_begin(lf_ctx *c, const uint64_t *w_ccs, lf_witness_job **job) {
    if (!c || !w_ccs || !job) return LF_ERR_INVALID;
    if (!c->have_ccs_any()) return LF_ERR_STATE;
    lf_witness_job *j = new lf_witness_job();
    c->io_jobs.fetch_add(1);
    j->fut = std::async(std::launch::async, [c, w_ccs, j]() -> int {
        struct Done { lf_ctx *c; ~Done() { c->io_jobs.fetch_sub(1); } } done{c};
        if (c->bb || c->xb.on) return lf_witness_from_w_ccs(c, w_ccs, &j->w);
        std::lock_guard<std::mutex> g(c->io_mu);
        if (hipSetDevice(c->device) != hipSuccess) return LF_ERR_HIP;
        t_lane = 2;
        return witness_from_w_ccs_lane(c, w_ccs, &j->w);
    });
    *job = j;
    return LF_OK;
}
int lf_witness_job_finish(lf_witness_job *job, lf_witness **out) {
    if (!job) return LF_ERR_INVALID;
    const int rc = job->fut.valid() ? job->fut.get() : LF_ERR_STATE;
    if (rc == LF_OK && out) *out = job->w;
    else if (job->w) lf_witness_free(job->w);      // (a caller that abandons the job passes out = NULL)
    delete job;
    return rc;
}
int lf_witness_from_w_ccs(lf_ctx *c, const uint64_t *w_ccs, lf_witness **out) {
    if (LF_XB(c) && w_ccs && c->have_ccs_any()) { XB x(c); return lf_witness_from_w_ccs(c, x.ring_in(w_ccs, c->params_any().wit_len), out); }
    if (!c || !w_ccs || !out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_from_w_ccs(w_ccs, out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    return witness_from_w_ccs_lane(c, w_ccs, out);
}
int lf_witness_from_f_coeff(lf_ctx *c, const uint64_t *f_coeff, lf_witness **out) {
    if (!c || !f_coeff || !out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_from_f_coeff(f_coeff, out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    u64 *d;
    RET(c->tbuf("io_c", c->N * 24, &d));
    RET(up_ring(c, f_coeff, c->N, d));
    return witness_from_coef_table(c, d, out);
}
int lf_witness_from_f(lf_ctx *c, const uint64_t *f_ntt, lf_witness **out) {
    if (LF_XB(c) && f_ntt && c->have_ccs_any()) { XB x(c); return lf_witness_from_f(c, x.ring_in(f_ntt, c->N_any()), out); }
    if (!c || !f_ntt || !out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_from_f(f_ntt, out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    u64 *a, *d;
    RET(c->tbuf("io_a", c->N * 24, &a));
    RET(c->tbuf("io_c", c->N * 24, &d));
    RET(up_ring(c, f_ntt, c->N, a));
    launch_icrt_dense(c->d_icrt, a, d, c->N, c->stream());
    return witness_from_coef_table(c, d, out);
}
int lf_witness_get_f_coeff(lf_ctx *c, const lf_witness *w, uint64_t *out) {
    if (!c || !w || !out || w->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_get_f_coeff(w, out);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    u64 *d;
    RET(c->tbuf("io_c", w->N * 24, &d));
    launch_i32_to_coef(w->planes, d, w->N, c->stream());
    return down_ring(c, d, w->N, out);
}
int lf_witness_get_f(lf_ctx *c, const lf_witness *w, uint64_t *out) {
    if (LF_XB(c) && w && out) { XB x(c); int rc = lf_witness_get_f(c, w, out); if (rc == LF_OK) x.ring_out(out, w->N); return rc; }
    if (!c || !w || !out || w->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_get_f(w, out);
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (w->f_ntt) return down_ring(c, w->f_ntt, w->N, out);      // built inside the fold step that produced this witness
    u64 *d, *e;
    RET(c->tbuf("io_c", w->N * 24, &d));
    RET(c->tbuf("io_b", w->N * 24, &e));
    launch_i32_to_coef(w->planes, d, w->N, c->stream());
    launch_crt_fwd(c->dcrt, d, e, w->N, c->stream());
    return down_ring(c, e, w->N, out);
}
int lf_witness_get_w_ccs(lf_ctx *c, const lf_witness *w, uint64_t *out) {
    if (LF_XB(c) && w && out && c->have_ccs_any()) { XB x(c); int rc = lf_witness_get_w_ccs(c, w, out); if (rc == LF_OK) x.ring_out(out, c->params_any().wit_len); return rc; }
    if (!c || !w || !out || w->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_get_w_ccs(w, out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    if (w->w_ccs && w->w_bytes == (size_t)c->P.wit_len * 24 * 8) return down_ring(c, w->w_ccs, c->P.wit_len, out);
    u64 *e;
    RET(c->tbuf("io_b", (size_t)c->P.wit_len * 24, &e));
    launch_recompose_crt(c->dcrt, w->planes, w->N, c->P.wit_len, c->P.L, c->P.B, 1, 0, e, c->P.wit_len, 0, c->stream());
    return down_ring(c, e, c->P.wit_len, out);
}
int lf_witness_commit(lf_ctx *c, const lf_witness *w, uint64_t *cm_out) {
    if (LF_XB(c) && w && cm_out) { XB x(c); int rc = lf_witness_commit(c, w, cm_out); if (rc == LF_OK) x.ring_out(cm_out, c->bb ? c->bb->kappa() : c->kappa); return rc; }
    if (!c || !w || !cm_out || w->ctx != c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_commit(w, cm_out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->A_loaded) return LF_ERR_STATE;
    if (w->N != c->nA_total) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    u64 *o;
    RET(c->tbuf("io_o", (size_t)c->kappa * 24, &o));
    // the int32 planes of the handle are the operand: five base-128 digit planes, no NTT of the witness
    c->ev_reset();
    RET(commit_dev_i8g(c, nullptr, 0, 1, w->planes + c->A_col0, w->N, o, true));   // timed: lf_last_kernel_stats reports the stand-alone kernel
    c->ev_collect();
    return commit_download(c, o, (size_t)c->kappa * 24, cm_out);
}
// pool of recycled witness-plane buffers: process-wide (a witness may be freed after its context), keyed by device and size
namespace {
struct PoolEnt { int device; size_t bytes; int32_t *p; };
std::mutex g_pool_mu;
std::vector<PoolEnt> g_pool;
}  // namespace
int lf_planes_alloc(lf_ctx *c, size_t bytes, int32_t **out) {
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i].device == c->device && g_pool[i].bytes == bytes) {
                *out = g_pool[i].p;
                g_pool.erase(g_pool.begin() + (long)i);
                return LF_OK;
            }
    }
    return lf_dev_malloc(out, bytes) == hipSuccess ? LF_OK : LF_ERR_HIP;
}
static void planes_release_dev(int device, size_t bytes, int32_t *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        if (g_pool.size() < 9) { g_pool.push_back({device, bytes, p}); return; }   // planes, f and w_ccs of up to three witnesses
    }
    (void)hipFree(p);
}
void lf_planes_release(lf_ctx *c, size_t bytes, int32_t *p) { planes_release_dev(c->device, bytes, p); }
static void planes_pool_drop(int device) {
    std::lock_guard<std::mutex> g(g_pool_mu);
    for (size_t i = 0; i < g_pool.size();)
        if (g_pool[i].device == device) { (void)hipFree(g_pool[i].p); g_pool.erase(g_pool.begin() + (long)i); }
        else i++;
}
void lf_witness_free(lf_witness *w) {
    if (!w) return;
    // the context may be gone already (callers close contexts before their witnesses): use only what the handle itself carries
    (void)hipSetDevice(w->device);
    planes_release_dev(w->device, w->plane_bytes, w->planes);
    planes_release_dev(w->device, w->f_bytes, (int32_t *)w->f_ntt);
    planes_release_dev(w->device, w->w_bytes, (int32_t *)w->w_ccs);
    delete w;
}

// ---- transcript ------------------------------------------------------------------------------------------------------------
lf_transcript *lf_transcript_new(void) { return new lf_transcript(); }
lf_transcript *lf_transcript_new_ring(int ring) {
    if (ring == LF_RING_GOLDILOCKS) return new lf_transcript();
    if (ring != LF_RING_BABYBEAR) return nullptr;
    lf_transcript *t = new lf_transcript();
    t->bb = new lfbb::BbTranscript();
    return t;
}
lf_transcript *lf_transcript_clone(const lf_transcript *t) { return t ? new lf_transcript(*t) : nullptr; }
void lf_transcript_free(lf_transcript *t) { delete t; }
void lf_transcript_absorb_fq(lf_transcript *t, const uint64_t *x, size_t n) {
    if (t->bb) t->bb->absorb_fq(x, n);
    else t->t.absorb_fq(x, n);
}
void lf_transcript_absorb_ring(lf_transcript *t, const uint64_t *e, size_t n) {
    if (t->bb) t->bb->absorb_ring(e, n);
    else t->t.absorb_ring(e, n);
}
void lf_transcript_get_challenge(lf_transcript *t, uint64_t *o) {
    if (t->bb) {
        lfbb::H9 c = t->bb->get_challenge();
        memcpy(o, c.c, sizeof(c.c));
        return;
    }
    Fq3 c = t->t.get_challenge();
    o[0] = c.c[0]; o[1] = c.c[1]; o[2] = c.c[2];
}
void lf_transcript_get_short_challenge(lf_transcript *t, uint64_t *o) {
    if (t->bb) t->bb->get_short_challenge(o);
    else t->t.get_short_challenge(o);
}
void lf_transcript_squeeze_bytes(lf_transcript *t, uint8_t *out, size_t n) {
    // CryptographicSponge::squeeze_bytes of the arkworks-0.4 PoseidonSponge (Transcript::squeeze_bytes, transcript/poseidon.rs:62-64): ceil(n / usable) field
    // elements, usable = (modulus bits - 1) / 8 low little-endian bytes of each (7 Goldilocks, 3 BabyBear), truncated to n
    if (!t || !out || !n) return;
    const size_t usable = t->bb ? 3 : 7, ne = (n + usable - 1) / usable;
    std::vector<u64> e(ne);
    if (t->bb) t->bb->squeeze(e.data(), ne);
    else t->t.squeeze(e.data(), ne);
    for (size_t i = 0, o = 0; i < ne && o < n; i++)
        for (size_t j = 0; j < usable && o < n; j++) out[o++] = (uint8_t)(e[i] >> (8 * j));
}
void lf_poseidon_permute(uint64_t *state, int plain) {
    if (plain == 2) Transcript::permute_scalar(state);
    else if (plain) Transcript::permute_plain(state);
    else Transcript::permute(state);
}
void lf_poseidon_permute_ring(uint64_t *state, int plain, int ring) {
    if (ring == LF_RING_BABYBEAR) {
        if (plain == 2) lfbb::BbTranscript::permute_scalar(state);
        else if (plain) lfbb::BbTranscript::permute_plain(state);
        else lfbb::BbTranscript::permute(state);
    } else lf_poseidon_permute(state, plain);
}
void lf_poseidon_params(uint64_t *ark, uint64_t *mds) {
    const u64 *a, *m;
    Transcript::params(&a, &m);
    memcpy(ark, a, 720 * 8);
    memcpy(mds, m, 576 * 8);
}
void lf_poseidon_params_ring(uint64_t *ark, uint64_t *mds, int ring) {
    const u64 *a, *m;
    if (ring == LF_RING_BABYBEAR) lfbb::BbTranscript::params(&a, &m);
    else Transcript::params(&a, &m);
    memcpy(ark, a, 720 * 8);
    memcpy(mds, m, 576 * 8);
}

int lf_last_phase_ms(lf_ctx *c, float *out) {
    if (!c || !out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->last_phase_ms(out);
    for (int i = 0; i < LF_N_PHASES; i++) out[i] = c->phase_ms[i];
    return LF_OK;
}
// wall-clock marks of the caller thread during the last lf_fold_step (Goldilocks driver): name i (NUL-terminated, at most 31 characters) at
// names + 32 i, ms[i] = milliseconds since the start of the step.  Returns the number of marks written (<= max_marks), < 0 on error.
int lf_last_timeline(lf_ctx *c, char *names, double *ms, int max_marks) {
    if (!c || !names || !ms || max_marks < 0) return LF_ERR_INVALID;
    if (c->bb) return 0;
    int n = 0;
    for (auto &m : c->tl_marks) {
        if (n >= max_marks) break;
        const char *w = m.first;
        while (*w == ' ') w++;
        snprintf(names + 32 * n, 32, "%s", w);
        ms[n++] = m.second;
    }
    return n;
}
// measurement hook of tools/gpu_i8prof.sh (not part of the prover interface, not declared in lfhip.h): per-phase clock totals of the last commit
// launch made with LF_I8_PROF set
int lf_abi_version(void) { return LFHIP_ABI_VERSION; }
int lf_debug_i8_prof(uint64_t *out64) {   // (LF_I8G_PROF set: the table of the general-commit kernel instead)
    if (!out64) return LF_ERR_INVALID;
    return getenv("LF_I8G_PROF") ? ajtai_i8g_read_prof((unsigned long long *)out64) : ajtai_i8_read_prof((unsigned long long *)out64);
}
// (not part of the ABI: tools/i8g_prof.py) per-workgroup loop durations of the last profiled general commit
extern "C" int lfdbg_i8g_wg(unsigned int *out512) { return out512 ? ajtai_i8g_read_wg(out512) : -1; }
int lf_last_fold_paths(lf_ctx *c, unsigned *sv_round_mask) {
    if (!c || !sv_round_mask) return LF_ERR_INVALID;
    *sv_round_mask = c->bb ? c->bb->fold_paths() : c->sv_round_mask;
    return LF_OK;
}
int lf_last_fold_split_rounds(lf_ctx *c, unsigned *round_mask) {
    if (!c || !round_mask) return LF_ERR_INVALID;
    *round_mask = c->bb ? c->bb->fold_split_rounds() : c->fold_split_mask;
    return LF_OK;
}
int lf_last_lin_split_rounds(lf_ctx *c, unsigned *rounds) {
    if (!c || !rounds) return LF_ERR_INVALID;
    *rounds = c->bb ? 0u : c->lin_split_rounds;
    return LF_OK;
}
int lf_last_kernel_stats(lf_ctx *c, float *fold_ms, int *fold_n, float *aj_ms, int *aj_n) {
    if (!c) return LF_ERR_INVALID;
    if (c->bb) return c->bb->last_kernel_stats(fold_ms, fold_n, aj_ms, aj_n);
    if (fold_ms) *fold_ms = c->k_fold_ms;
    if (fold_n) *fold_n = c->k_fold_n;
    if (aj_ms) *aj_ms = c->k_ajtai_ms;
    if (aj_n) *aj_n = c->k_ajtai_n;
    return LF_OK;
}

// ---- host-side verifier ---------------------------------------------------------------------------------------
namespace {
struct GoldV {
    static constexpr int RE = 24, TAU = 3;
    static u64 modulus() { return LF_P; }
    typedef Fq3 Ext;
    typedef Transcript Tr;
    HostRing ring;
    void mul(const u64 *a, const u64 *b, u64 *o) const { ring.mul_ntt(a, b, o); }
    void mul_ext(const u64 *a, Ext s, u64 *o) const { ring.mul_fq3(a, s, o); }
    static void add(const u64 *a, const u64 *b, u64 *o) { HostRing::add(a, b, o); }
    static void sub(const u64 *a, const u64 *b, u64 *o) { HostRing::sub(a, b, o); }
    static void from_u64(u64 v, u64 *o) { HostRing::from_u64(v, o); }
    static void from_ext(Ext e, u64 *o) { HostRing::from_fq3(e, o); }
    static Ext ext_of(const u64 *e) { return fq3_make(e[0], e[1], e[2]); }
    static Ext ext_from_u64(u64 v) { return fq3_make(v % LF_P, 0, 0); }
    Ext ext_mul(Ext a, Ext b) const { return ring.mul3(a, b); }
    static Ext ext_add(Ext a, Ext b) { return fq3_add(a, b); }
    static Ext ext_sub(Ext a, Ext b) { return fq3_sub(a, b); }
    Ext ext_inv(Ext a) const { return ring.inv3(a); }
    static void absorb_ext(Tr &tr, Ext e) { tr.absorb_fq3_as_ring(e); }
    void crt(const u64 *c, u64 *o) const { ring.crt(c, o); }
    static u64 fmul(u64 a, u64 b) { return fq_mul(a % LF_P, b % LF_P); }
    static u64 fadd(u64 a, u64 b) { return fq_add(a, b); }
    static void rot_x(u64 *a) {   // multiply by X modulo X^24 - X^12 + 1
        u64 top = a[23];
        for (int j = 23; j > 0; j--) a[j] = a[j - 1];
        a[0] = fq_neg(top);
        a[12] = fq_add(a[12], top);
    }
};
}  // namespace

int lf_verify_host(int ring, const lf_params *p, const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *c, lf_transcript *t,
                   const uint64_t *acc, const uint64_t *cm_i, const uint64_t *proof, uint64_t *lcccs_out, int *failed_stage) {
    if (!p || !S_off || !S_idx || !c || !t || !acc || !cm_i || !proof || !lcccs_out) return LF_ERR_INVALID;
    if (p->s == 0 || p->s > 40 || p->K == 0 || p->K > 32 || p->q == 0 || p->q > 8 || p->t == 0 || p->t > 16) return LF_ERR_UNSUPPORTED;
    if (((size_t)1 << p->s) < (size_t)p->wit_len * p->L) return LF_ERR_SIZE_BOUNDS;   // sanity_check, nifs.rs:165-173
    if (failed_stage) *failed_stage = 0;
    if (ring == LF_RING_BABYBEAR) return t->bb ? lfbb::bb_verify_host(p, S_off, S_idx, c, *t->bb, acc, cm_i, proof, lcccs_out, failed_stage) : LF_ERR_INVALID;
    if (ring != LF_RING_GOLDILOCKS || t->bb) return LF_ERR_INVALID;
    static const GoldV *gv = [] {
        GoldV *g = new GoldV();
        u64 nr, y[24];
        default_ring(&nr, y);
        build_crt_tables(nr, y, g->ring.T);
        return g;
    }();
    lfv::Verifier<GoldV> V(*gv, *p, S_off, S_idx, c);
    int rc = V.verify(t->t, acc, cm_i, proof, lcccs_out);
    if (failed_stage) *failed_stage = V.stage;
    return rc;
}


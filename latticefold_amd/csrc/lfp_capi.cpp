// lfp_capi.cpp -- host driver + C ABI (include/lfplus.h) of the LatticeFold+ double commitment on the Frog ring.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "../../include/lfplus.h"
#include "lfp_kernels.h"
#include "lfp_ctx.h"


extern "C" int lfplus_ctx_create(int device, lfplus_ctx **out) {
    if (!out) return LFPLUS_E_ARG;
    *out = nullptr;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || device < 0 || device >= cnt) return LFPLUS_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return LFPLUS_E_NO_DEVICE;
    lfplus_ctx *c = new lfplus_ctx;
    c->device = device;
    c->pool.device = device;
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess || lfp_dev_malloc(&c->err_d, 4) != hipSuccess) {
        delete c;
        return LFPLUS_E_HIP;
    }
    {   // the second stream (lfplus_rg_from_f_async) exists from the start -- creating a stream costs ~2 ms, which would land inside the first prove -- and has
        // the lowest priority: its pass fills the gaps of the latency-bound sumcheck rounds on `st`, it must not delay their kernels
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&c->st2, hipStreamNonBlocking, least) != hipSuccess || hipEventCreateWithFlags(&c->ev_ff, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (c->st2) (void)hipStreamDestroy(c->st2);
            c->st2 = nullptr; c->ev_ff = nullptr;      // (no second stream: the hint is ignored)
        }
    }
    *out = c;
    return LFPLUS_OK;
}
extern "C" void lfplus_ctx_destroy(lfplus_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st);
    if (c->st2) { (void)hipStreamSynchronize(c->st2); (void)hipStreamDestroy(c->st2); }
    if (c->ev_ff) (void)hipEventDestroy(c->ev_ff);
    c->A = nullptr;
    c->A_ref.reset();      // frees the matrix unless another context still shares it
    c->drop_mats();
    for (void *p : {(void *)c->f, (void *)c->Df, (void *)c->mtau, (void *)c->comMf, (void *)c->tau, (void *)c->coms, (void *)c->part, (void *)c->g}) c->own_free(p);
    c->pool.clear();       // (every block is idle now: they go to the process-wide cache)
    if (c->err_d) (void)hipFree(c->err_d);
    if (c->hpin) (void)hipHostFree(c->hpin);
    (void)hipStreamDestroy(c->st);
    delete c;
}
// Releases the scratch blocks that destroyed contexts left in the process-wide cache (lfp_ctx.h::LfpDevCache); device < 0: on every device
extern "C" void lfplus_scratch_trim(int device) { LfpDevCache::inst().trim(device); }
extern "C" size_t lfplus_scratch_bytes(int device) { return LfpDevCache::inst().held(device); }
extern "C" const char *lfplus_last_error(const lfplus_ctx *c) { return c ? c->err.c_str() : "null context"; }

// ---- column sharding over `world` ranks, one GPU each (lfplus.h) ------------------------------------------------------------------------
static int shard_geometry_ok(lfplus_ctx *c, int rank, int world) {
    if (!c) return LFPLUS_E_ARG;
    if (world < 1 || world > 64 || (world & (world - 1)) || rank < 0 || rank >= world) return fail(c, LFPLUS_E_ARG, "sharding: world must be a power of two <= 64, 0 <= rank < world");
    if (c->A || c->f) return fail(c, LFPLUS_E_ARG, "sharding must be set before the matrix and the witness (a rank keeps only its columns of A)");
    return LFPLUS_OK;
}
extern "C" int lfplus_set_sharding(lfplus_ctx *c, int rank, int world, lfplus_exchange_fn cb, void *user) {
    int rc = shard_geometry_ok(c, rank, world);
    if (rc) return rc;
    if (world > 1 && !cb) return fail(c, LFPLUS_E_ARG, "lfplus_set_sharding: no exchange callback");
    c->sh = std::make_shared<LfpShard>();
    c->sh->comm.rank = rank; c->sh->comm.world = world; c->sh->comm.cb = (lf_exchange_fn)cb; c->sh->comm.user = user;
    c->rank = rank; c->world = world;
    return LFPLUS_OK;
}
// TIMING MODEL (as lf_set_sharding_model, include/lfhip.h): the context is rank `rank` of `world` with no peers -- every kernel and host stage does that rank's
// share, every exchange gets zeros for the peers' words.  What such a prover returns is not a proof; tools/shard_model.py --lfplus measures a rank's share with it.
extern "C" int lfplus_set_sharding_model(lfplus_ctx *c, int rank, int world) {
    int rc = shard_geometry_ok(c, rank, world);
    if (rc) return rc;
    c->sh = std::make_shared<LfpShard>();
    c->sh->comm.rank = rank; c->sh->comm.world = world; c->sh->comm.model = world > 1;
    c->rank = rank; c->world = world;
    return LFPLUS_OK;
}
extern "C" int lfplus_dist_stats_words(lfplus_ctx *c, uint64_t *words_sent, int reset) {
    if (!c || !words_sent) return LFPLUS_E_ARG;
    *words_sent = c->sh ? c->sh->comm.words_sent : 0;
    if (reset && c->sh) c->sh->comm.words_sent = 0;
    return LFPLUS_OK;
}
extern "C" int lfplus_dist_unique_id(uint8_t *id128) { return lfdist::rccl_unique_id(id128) == 0 ? LFPLUS_OK : LFPLUS_E_HIP; }
extern "C" int lfplus_dist_init(lfplus_ctx *c, int rank, int world, const uint8_t *id128) {
    int rc = shard_geometry_ok(c, rank, world);
    if (rc) return rc;
    if (!id128) return fail(c, LFPLUS_E_ARG, "lfplus_dist_init: null id");
    HIPCHK(c, hipSetDevice(c->device));
    c->sh = std::make_shared<LfpShard>();
    if (lfdist::rccl_init(c->sh->comm, rank, world, id128) != 0) { c->sh.reset(); return fail(c, LFPLUS_E_HIP, "lfplus_dist_init: RCCL not loadable / ncclCommInitRank failed"); }
    c->rank = rank; c->world = world;
    return LFPLUS_OK;
}
extern "C" int lfplus_dist_stats(lfplus_ctx *c, uint64_t *n_exchanges, double *total_us, double *max_us, int reset) {
    if (!c) return LFPLUS_E_ARG;
    lfdist::Comm zero, &cm = c->sh ? c->sh->comm : zero;
    if (n_exchanges) *n_exchanges = cm.n_exchanges;
    if (total_us) *total_us = cm.us_total;
    if (max_us) *max_us = cm.us_max;
    if (reset) { cm.n_exchanges = 0; cm.us_total = cm.us_max = 0; }
    return LFPLUS_OK;
}

static int upload(lfplus_ctx *c, u64 **dst, const u64 *src, size_t words) {
    if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
    HIPCHK(c, lfp_dev_malloc(dst, words * 8));
    HIPCHK(c, hipMemcpyAsync(*dst, src, words * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return LFPLUS_OK;
}
static int shape_buffers(lfplus_ctx *c, u32 kappa, u64 n) {
    c->kappa = kappa;
    c->n = n;
    for (u64 **p : {&c->tau, &c->coms})
        if (*p) { c->own_free(*p); *p = nullptr; }
    if (c->mtau) { c->own_free(c->mtau); c->mtau = nullptr; }
    HIPCHK(c, c->own_alloc(&c->tau, n * 8));
    HIPCHK(c, c->own_alloc(&c->mtau, n));
    HIPCHK(c, c->own_alloc(&c->coms, (size_t)3 * kappa * 16 * 8));
    return LFPLUS_OK;
}
static void ff_join(lfplus_ctx *c);
extern "C" int lfplus_set_matrix(lfplus_ctx *c, const uint64_t *A, uint32_t kappa, uint64_t n) {
    if (!c || !A || !kappa || kappa > 64 || !n || n * (uint64_t)(c ? c->world : 1) > (1ull << 32)) return fail(c, LFPLUS_E_ARG, "lfplus_set_matrix: bad shape");
    if (c->sharded() && (n & (n - 1) || n < 4 * (u64)c->world)) return fail(c, LFPLUS_E_ARG, "lfplus_set_matrix: a sharded context takes the rank's n / world columns, a power of two >= 4 world");
    if (!canonical(A, (size_t)kappa * n * 16)) return fail(c, LFPLUS_E_ARG, "lfplus_set_matrix: non-canonical word");
    HIPCHK(c, hipSetDevice(c->device));
    ff_join(c);
    c->have = false;
    u64 *fresh = nullptr;   // a new allocation: contexts sharing the previous matrix keep it alive through their own reference
    int rc = upload(c, &fresh, A, (size_t)kappa * n * 16);
    if (rc) return rc;
    c->A = fresh;
    c->A_ref = std::shared_ptr<void>(fresh, [](void *p) { (void)hipFree(p); });
    c->nloc = n;
    c->row0 = (u64)c->rank * n;
    return shape_buffers(c, kappa, n * (u64)c->world);
}
// The commitment matrix of `from` (same device), not copied: PlusProver keeps one context per accumulated / fresh instance and one Ajtai matrix.
// The allocation is reference-counted: it lives until the last context holding it re-uploads or is destroyed.
extern "C" int lfplus_share_matrix(lfplus_ctx *c, lfplus_ctx *from) {
    if (!c || !from || c == from || !from->A || c->device != from->device) return fail(c, LFPLUS_E_ARG, "lfplus_share_matrix: bad arguments");
    if (c->f && from->sharded()) return fail(c, LFPLUS_E_ARG, "lfplus_share_matrix: share a sharded matrix before the witness is set");
    HIPCHK(c, hipSetDevice(c->device));
    c->have = false;
    c->A = from->A;
    c->A_ref = from->A_ref;
    c->sh = from->sh;          // the column slice of A fixes the shard geometry: sharers exchange over the same transport
    c->rank = from->rank; c->world = from->world; c->nloc = from->nloc; c->row0 = from->row0;
    return shape_buffers(c, from->kappa, from->n);
}
// The words are checked on the DEVICE, behind the upload (a host scan of a 2^20-row witness costs 5 ms, twice its upload): a non-canonical word fails the call
// and leaves the context WITHOUT a resident witness.
extern "C" int lfplus_set_witness(lfplus_ctx *c, const uint64_t *f, uint64_t n) {
    if (!c || !f || !n) return fail(c, LFPLUS_E_ARG, "lfplus_set_witness: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    ff_join(c);     // (an asynchronous from_f may still read the witness that is being replaced)
    c->have = false;
    if (!(c->f && c->nf == n)) {   // (same length as the resident witness: overwrite it, no hipFree / hipMalloc round trip per instance)
        c->own_free(c->f);
        c->f = nullptr; c->nf = 0;
        HIPCHK(c, c->own_alloc(&c->f, (size_t)n * 16 * 8));
    }
    c->nf = 0;
    u32 flag = 0;
    HIPCHK(c, hipMemsetAsync(c->err_d, 0, 4, c->st));
    HIPCHK(c, hipMemcpyAsync(c->f, f, (size_t)n * 16 * 8, hipMemcpyHostToDevice, c->st));
    lfp::launch_check_canonical(c->f, (size_t)n * 16, c->err_d, 4u, c->st);
    HIPCHK(c, hipMemcpyAsync(&flag, c->err_d, 4, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    if (flag & 4u) {
        c->own_free(c->f);
        c->f = nullptr;
        return fail(c, LFPLUS_E_ARG, "lfplus_set_witness: non-canonical word");
    }
    c->nf = n;
    return LFPLUS_OK;
}

static int log2_exact(u64 b) { return (b && !(b & (b - 1))) ? __builtin_ctzll(b) : -1; }
struct Plan {
    u32 J, nblk;      // phase 1: rows per block, blocks
    u32 J2, nblk2;    // phase 2
    size_t nout_m, nout_f;
};
static Plan plan_for(u64 n, u32 kappa, u32 k) {
    Plan p;
    // phase 1 is register-bound at 2 waves / SIMD (176 VGPRs for 2 rows x 4 digit planes): two blocks per CU = 512 blocks (measured
    // best of 512 / 768 / 1024, LFP_BLOCKS1 overrides), and not below 128 rows per block (a block's partial sums are k*kappa*256 words:
    // fewer, longer blocks keep that traffic well under the input's)
    constexpr long env1 = 0, env2 = 0;   // (blocks of the two phases: the defaults below)
    u64 J = (n + 511) / 512;
    if (J < 128) J = 128;
    if (env1 > 0) J = (n + env1 - 1) / env1;
    J = (J + lfp::TJ - 1) / lfp::TJ * lfp::TJ;
    p.J = (u32)J;
    p.nblk = (u32)((n + J - 1) / J);
    u64 J2 = (n + (env2 > 0 ? env2 : 1024) - 1) / (env2 > 0 ? env2 : 1024);     // phase 2 is a streaming pass: up to 1024 blocks, rows per block a multiple of the unrolled step
    J2 = (J2 + 127) / 128 * 128;
    p.J2 = (u32)J2;
    p.nblk2 = (u32)((n + J2 - 1) / J2);
    p.nout_m = (size_t)k * kappa * 256;
    p.nout_f = (size_t)kappa * 16;
    return p;
}
static int ensure_part(lfplus_ctx *c, size_t words) {
    if (c->part_cap >= words) return LFPLUS_OK;
    c->own_free(c->part);
    c->part = nullptr;
    c->part_cap = 0;
    HIPCHK(c, c->own_alloc(&c->part, words * 8));
    c->part_cap = words;
    return LFPLUS_OK;
}
// phase 1 over an arbitrary vector v (v == c->f for from_f); k == 0: only A v
static void enqueue_phase1(lfplus_ctx *c, const u64 *v, u64 b, u32 k, const Plan &p, u64 *part) {
    for (u32 i0 = 0; i0 < c->kappa;) {
        u32 icnt = lfp::group_size(c->kappa - i0), k0 = 0;
        do {
            lfp::Phase1Args a;
            a.f = v; a.A = c->A; a.n = c->nloc; a.kappa = c->kappa;   // v: the rank's rows (all of them unsharded)
            a.i0 = i0; a.icnt = icnt;
            a.k = k; a.k0 = k0; a.kcnt = k ? lfp::group_size(k - k0) : 0;
            a.b = b; a.sh = log2_exact(b); a.J = p.J;
            a.Df = c->Df; a.part = part;
            a.err = c->err_d;
            a.write_df = i0 == 0; a.do_f = k0 == 0;
            lfp::launch_phase1(a, p.nblk, c->st);
            k0 += a.kcnt;
        } while (k0 < k);
        i0 += icnt;
    }
}
static int check_params(lfplus_ctx *c, u64 b, u32 k, u32 l) {
    if (!c) return LFPLUS_E_ARG;
    if (!c->A || !c->f) return fail(c, LFPLUS_E_ARG, "lfplus_rg_from_f: matrix / witness not set");
    if (c->nf != c->n) return fail(c, LFPLUS_E_ARG, "lfplus_rg_from_f: witness length differs from the matrix width");
    if (b < 2 || b > (1ull << 31) || !k || k > 16 || !l || l > 64) return fail(c, LFPLUS_E_ARG, "lfplus_rg_from_f: parameters outside the envelope");
    unsigned __int128 need = (unsigned __int128)c->kappa * k * 16 * l * 16;
    if (need >= c->n) return fail(c, LFPLUS_E_SMALL_N, "lfplus_rg_from_f: small n unsupported, must be > kappa*k*d*l*d (utils.rs:33-39)");
    return LFPLUS_OK;
}
static int prepare(lfplus_ctx *c, u32 k, const Plan &p) {
    size_t dfb = (size_t)k * c->nloc * 16, cmw = p.nout_m + p.nout_f;
    if (c->Df_cap < dfb) {
        c->own_free(c->Df);
        c->Df = nullptr; c->Df_cap = 0;
        HIPCHK(c, c->own_alloc(&c->Df, dfb));
        c->Df_cap = dfb;
    }
    if (c->comMf_cap < cmw) {
        c->own_free(c->comMf);
        c->comMf = nullptr; c->comMf_cap = 0;
        HIPCHK(c, c->own_alloc(&c->comMf, cmw * 8));
        c->comMf_cap = cmw;
    }
    return ensure_part(c, (size_t)p.nblk * (p.nout_m + p.nout_f) + (size_t)p.nblk2 * 2 * p.nout_f);
}
// c->comMf: [comM_f (k, kappa, 16, 16) | cm_f (kappa, 16)];  c->coms: [C_Mf | cm_mtau]
// device vector of `words` canonical words: sum over the ranks mod p, in place (a sharded prover's partial commitments; drains the stream)
static int xsum_dev(lfplus_ctx *c, u64 *d, size_t words) {
    if (!c->sharded()) return LFPLUS_OK;
    std::vector<u64> h(words);
    HIPCHK(c, hipMemcpyAsync(h.data(), d, words * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    int rc = lfp_xsum(c, h.data(), words);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(d, h.data(), words * 8, hipMemcpyHostToDevice, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));   // h is a local buffer
    return LFPLUS_OK;
}
// Sharded: both passes run over the rank's columns of A and rows of f / tau; the partial comM_f | cm_f and C_Mf | cm_mtau are exchanged (one all-gather +
// modular sum each: 2 per double commitment), and tau -- the gadget digits of the COMPLETE comM_f, a short non-zero prefix -- is rebuilt whole on every rank.
static int enqueue_from_f(lfplus_ctx *c, u64 b, u32 k, u32 l, const Plan &p) {
    u64 *part1 = c->part, *part2 = part1 + (size_t)p.nblk * (p.nout_m + p.nout_f);
    size_t need = (size_t)c->kappa * k * 16 * l * 16;
    const u32 nout1 = (u32)(p.nout_m + p.nout_f);
    (void)hipMemsetAsync(c->err_d, 0, 4, c->st);
    (void)hipMemsetAsync(c->tau + need, 0, (c->n - need) * 8, c->st);
    enqueue_phase1(c, c->f + c->row0 * 16, b, k, p, part1);
    if (!c->sharded()) {
        lfp::launch_reduce(part1, p.nblk, nout1, c->comMf, (u32)p.nout_m, c->kappa, k, lfp::D / 2, l, c->tau, c->st);
    } else {
        lfp::launch_reduce(part1, p.nblk, nout1, c->comMf, 0, c->kappa, k, 2, 0, nullptr, c->st);
        int rc = xsum_dev(c, c->comMf, nout1);
        if (rc) return rc;
        lfp::launch_reduce(c->comMf, 1, nout1, c->comMf, (u32)p.nout_m, c->kappa, k, lfp::D / 2, l, c->tau, c->st);   // one "block": the split of the summed comM_f
        lfp::launch_mtau_all(c->tau, c->n, c->mtau, c->err_d, c->st);
    }
    for (u32 i0 = 0; i0 < c->kappa;) {
        lfp::Phase2Args a;
        a.A = c->A; a.tau = c->tau + c->row0; a.n = c->nloc; a.kappa = c->kappa;
        a.i0 = i0; a.icnt = lfp::group_size(c->kappa - i0); a.J = p.J2;
        a.mtau = c->mtau + c->row0;
        a.part = part2;
        a.err = c->err_d;
        lfp::launch_phase2(a, p.nblk2, c->st);
        i0 += a.icnt;
    }
    lfp::launch_reduce(part2, p.nblk2, (u32)(2 * p.nout_f), c->coms, 0, c->kappa, k, 2, 0, nullptr, c->st);
    return xsum_dev(c, c->coms, 2 * p.nout_f);
}
static int finish(lfplus_ctx *c) {
    u32 flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, c->err_d, 4, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    HIPCHK(c, hipGetLastError());
    if (c->sharded()) {     // every rank must take the same branch: a digit outside the domain on ANY rank fails the call on all of them
        std::vector<u64> all;
        u64 mine = flag;
        int rc = lfp_allgather(c, &mine, 1, all);
        if (rc) return rc;
        for (u64 x : all) flag |= (u32)x;
    }
    if (flag) return fail(c, LFPLUS_E_EXP_DOMAIN, flag & 1 ? "lfplus_rg_from_f: a digit of f is outside (-d/2, d/2)" : "lfplus_rg_from_f: tau outside (-d/2, d/2)");
    return LFPLUS_OK;
}
// An asynchronous from_f in flight (lfplus_rg_from_f_async): wait for it and publish its result -- or, if it failed, leave the context without one (the
// caller that needs a result runs the synchronous pass and gets the error there).
static void ff_join(lfplus_ctx *c) {
    if (!c || !c->ff_pending) return;
    c->ff_pending = false;
    if (hipStreamSynchronize(c->st2) != hipSuccess) { (void)hipGetLastError(); return; }
    const std::string keep = c->err;
    if (finish(c) == LFPLUS_OK) { c->k = c->ff_k; c->l = c->ff_l; c->have = true; }
    else c->err = keep;
}
extern "C" int lfplus_join_async(lfplus_ctx *c) { if (!c) return LFPLUS_E_ARG; ff_join(c); return LFPLUS_OK; }
// RgInstance::from_f of the resident witness, enqueued on the context's SECOND stream; returns at once.  The pass needs the witness, the matrix and the
// parameters only -- no challenge -- so a prover issues it as soon as the witness is resident and lets it run next to the linearization's latency-bound
// sumcheck rounds (rgchk.rs:260-331 is called from Mlin::mlin, mlin.rs:52-60, after every linearization: the order of the results is not observable).
// lfplus_rg_from_f with the same parameters collects the result (errors of the pass are reported there); any other call that touches the witness or the
// from_f buffers waits for it first.  A sharded context ignores the hint (its pass exchanges through the host: the synchronous call does the work).
extern "C" int lfplus_rg_from_f_async(lfplus_ctx *c, uint64_t b, uint32_t k, uint32_t l) {
    int rc = check_params(c, b, k, l);
    if (rc) return rc;
    if (c->sharded() || !c->st2 || !c->ev_ff) return LFPLUS_OK;
    HIPCHK(c, hipSetDevice(c->device));
    ff_join(c);
    c->have = false;
    Plan p = plan_for(c->nloc, c->kappa, k);
    if ((rc = prepare(c, k, p))) return rc;
    HIPCHK(c, hipEventRecord(c->ev_ff, c->st));            // behind whatever made the witness resident
    HIPCHK(c, hipStreamWaitEvent(c->st2, c->ev_ff, 0));
    std::swap(c->st, c->st2);
    rc = enqueue_from_f(c, b, k, l, p);
    std::swap(c->st, c->st2);
    if (rc) { (void)hipStreamSynchronize(c->st2); return rc; }
    c->ff_pending = true; c->ff_b = b; c->ff_k = k; c->ff_l = l;
    return LFPLUS_OK;
}
extern "C" int lfplus_rg_from_f(lfplus_ctx *c, uint64_t b, uint32_t k, uint32_t l) {
    int rc = check_params(c, b, k, l);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->ff_pending && c->ff_b == b && c->ff_k == k && c->ff_l == l) {      // issued ahead (lfplus_rg_from_f_async): collect it
        c->ff_pending = false;
        HIPCHK(c, hipStreamSynchronize(c->st2));
        if ((rc = finish(c))) return rc;
        c->k = k; c->l = l; c->have = true;
        return LFPLUS_OK;
    }
    ff_join(c);
    c->have = false;
    Plan p = plan_for(c->nloc, c->kappa, k);
    if ((rc = prepare(c, k, p))) return rc;
    if ((rc = enqueue_from_f(c, b, k, l, p))) return rc;
    if ((rc = finish(c))) return rc;
    c->k = k;
    c->l = l;
    c->have = true;
    return LFPLUS_OK;
}
extern "C" int lfplus_rg_from_f_timed(lfplus_ctx *c, uint64_t b, uint32_t k, uint32_t l, uint32_t iters, double *ms_avg) {
    if (!ms_avg || !iters) return fail(c, LFPLUS_E_ARG, "lfplus_rg_from_f_timed: bad arguments");
    int rc = lfplus_rg_from_f(c, b, k, l);   // warm-up + validation
    if (rc) return rc;
    if (c->sharded()) return fail(c, LFPLUS_E_ARG, "lfplus_rg_from_f_timed: unsharded contexts only (a sharded pass drains the stream at its two exchanges)");
    Plan p = plan_for(c->nloc, c->kappa, k);
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipEventRecord(e0, c->st));
    for (u32 it = 0; it < iters; it++) (void)enqueue_from_f(c, b, k, l, p);
    HIPCHK(c, hipEventRecord(e1, c->st));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_avg = (double)ms / iters;
    return finish(c);
}
extern "C" int lfplus_rg_read(lfplus_ctx *c, int8_t *Df, uint64_t *comMf, uint64_t *tau, int8_t *mtau, uint64_t *cm_f, uint64_t *C_Mf, uint64_t *cm_mtau) {
    if (!c) return LFPLUS_E_ARG;
    ff_join(c);
    if (!c->have) return fail(c, LFPLUS_E_ARG, "lfplus_rg_read: no result (run lfplus_rg_from_f first)");
    if (Df && c->sharded()) return fail(c, LFPLUS_E_ARG, "lfplus_rg_read: D_f of a sharded context exists per rank only (pass NULL)");
    HIPCHK(c, hipSetDevice(c->device));
    size_t cw = (size_t)c->kappa * 16;
    if (Df) HIPCHK(c, hipMemcpyAsync(Df, c->Df, (size_t)c->k * c->n * 16, hipMemcpyDeviceToHost, c->st));
    if (comMf) HIPCHK(c, hipMemcpyAsync(comMf, c->comMf, (size_t)c->k * c->kappa * 256 * 8, hipMemcpyDeviceToHost, c->st));
    if (tau) HIPCHK(c, hipMemcpyAsync(tau, c->tau, c->n * 8, hipMemcpyDeviceToHost, c->st));
    if (mtau) HIPCHK(c, hipMemcpyAsync(mtau, c->mtau, c->n, hipMemcpyDeviceToHost, c->st));
    if (cm_f) HIPCHK(c, hipMemcpyAsync(cm_f, c->comMf + (size_t)c->k * c->kappa * 256, cw * 8, hipMemcpyDeviceToHost, c->st));
    if (C_Mf) HIPCHK(c, hipMemcpyAsync(C_Mf, c->coms, cw * 8, hipMemcpyDeviceToHost, c->st));
    if (cm_mtau) HIPCHK(c, hipMemcpyAsync(cm_mtau, c->coms + cw, cw * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return LFPLUS_OK;
}
extern "C" int lfplus_commit(lfplus_ctx *c, const uint64_t *v, uint64_t n, uint64_t *out) {
    if (!c || !v || !out) return fail(c, LFPLUS_E_ARG, "lfplus_commit: null argument");
    if (!c->A || n != c->n) return fail(c, LFPLUS_E_ARG, "lfplus_commit: matrix not set / length mismatch");
    if (!canonical(v, (size_t)n * 16)) return fail(c, LFPLUS_E_ARG, "lfplus_commit: non-canonical word");
    HIPCHK(c, hipSetDevice(c->device));
    ff_join(c);
    Plan p = plan_for(c->nloc, c->kappa, 0);
    int rc = ensure_part(c, (size_t)p.nblk * p.nout_f + p.nout_f);
    if (rc) return rc;
    u64 *dv = nullptr;      // the rank's rows of v (all of them unsharded): partial commitment, summed over the ranks below
    HIPCHK(c, lfp_dev_malloc(&dv, (size_t)c->nloc * 16 * 8));
    (void)hipMemcpyAsync(dv, v + c->row0 * 16, (size_t)c->nloc * 16 * 8, hipMemcpyHostToDevice, c->st);
    u64 *res = c->part + (size_t)p.nblk * p.nout_f;
    enqueue_phase1(c, dv, 2, 0, p, c->part);
    lfp::launch_reduce(c->part, p.nblk, (u32)p.nout_f, res, 0, c->kappa, 0, 2, 0, nullptr, c->st);
    (void)hipMemcpyAsync(out, res, p.nout_f * 8, hipMemcpyDeviceToHost, c->st);
    hipError_t e = hipStreamSynchronize(c->st);
    (void)hipFree(dv);
    HIPCHK(c, e);
    HIPCHK(c, hipGetLastError());
    return lfp_xsum(c, out, p.nout_f);
}
// r 2^64 mod p (Montgomery form) on the host
static u64 to_mont(u64 a) { return (u64)((((unsigned __int128)a) << 64) % lfp::P); }
// dst0 / dst1 (optional): contexts that receive F0 / F1 as their resident witnesses (device-to-device; `c` itself is allowed: its witness is consumed first)
static int decompose_impl(lfplus_ctx *c, uint64_t B, const uint64_t *r_a, const uint64_t *r_b, uint32_t nm, const uint32_t *const *rowptr,
                          const uint32_t *const *col, const uint64_t *const *val, uint64_t *F0, uint64_t *F1, uint64_t *C0, uint64_t *C1, uint64_t *v0,
                          uint64_t *v1, lfplus_ctx *dst0, lfplus_ctx *dst1) {
    if (!c || !r_a || !r_b || (nm && rowptr && (!col || !val))) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: null argument");
    for (lfplus_ctx *d : {dst0, dst1})
        if (d && (d->device != c->device || d->n != c->n)) return fail(c, LFPLUS_E_ARG, "lfplus_decompose_resident: the receiving context is on another device or has another width");
    if (dst0 && dst0 == dst1) return fail(c, LFPLUS_E_ARG, "lfplus_decompose_resident: F0 and F1 need two contexts");
    for (lfplus_ctx *d : {c, dst0, dst1}) ff_join(d);
    const bool resident = nm && !rowptr;   // the matrices lfplus_set_matrices left in the context
    if (resident && (c->mats.size() != nm || c->mats_n != c->n)) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: no resident matrices of this shape");
    if (!c->A || !c->f || c->nf != c->n) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: matrix / witness not set or of different length");
    const u64 n = c->n;
    if (n & (n - 1)) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: n must be a power of two (nvars = log2(A.ncols))");
    if (B < 2 || B > (1ull << 62) || nm > 64) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: parameters outside the envelope");
    u32 nvars = 0;
    while (((u64)1 << nvars) < n) nvars++;
    if (!canonical(r_a, (size_t)nvars * 16) || !canonical(r_b, (size_t)nvars * 16)) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: non-canonical point");
    for (u32 j = 0; j < nm && !resident; j++) {
        if (!rowptr[j] || !col[j] || !val[j] || rowptr[j][0] != 0) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: bad CSR");
        for (u64 r = 0; r < n; r++)
            if (rowptr[j][r + 1] < rowptr[j][r]) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: rowptr not monotone");
        const u32 nnz = rowptr[j][n];
        for (u32 k = 0; k < nnz; k++)
            if (col[j][k] >= n) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: column index out of range");
        if (!canonical(val[j], (size_t)nnz * 16)) return fail(c, LFPLUS_E_ARG, "lfplus_decompose: non-canonical coefficient");
    }
    HIPCHK(c, hipSetDevice(c->device));
    // Sharded: the witness is whole on every rank (after lfplus_mlin: all-gathered), F0 / F1 are cut whole (element-wise, cheap) because the M_j F_i rows
    // read arbitrary columns; the tables -- the rank's rows of F_i and of M_j F_i -- are fixed locally for the first log2(n / world) variables, gathered (one
    // entry per table and rank) and finished replicated; the two commitments are partial sums over the rank's columns (one exchange).
    const u32 E = 2 * (1 + nm), T = 2 * E;        // vectors (F0, M_j F0 .., F1, M_j F1 ..), tables = one per vector and point
    const size_t vw = (size_t)n * 16, nl = c->nloc, lw = nl * 16, r0w = c->row0 * 16;
    u64 *buf = nullptr;
    // F0 | F1 | tables T*nloc | ping T*nloc/2 | rM nvars*32 | gathered tables T*world
    const size_t words = 2 * vw + (size_t)T * lw + (size_t)T * lw / 2 + (size_t)nvars * 32 + 64 + (size_t)T * c->world * 16;
    buf = (u64 *)c->pool.get(words * 8);
    if (!buf) return fail(c, LFPLUS_E_HIP, "hipMalloc (decompose tables)");
    u64 *dF0 = buf, *dF1 = dF0 + vw, *tab = dF1 + vw, *ping = tab + (size_t)T * lw, *drM = ping + (size_t)T * lw / 2, *gath = drM + (size_t)nvars * 32 + 64;
    int rc = LFPLUS_OK;
    std::vector<void *> tofree;
    auto cleanup = [&]() { for (void *q : tofree) c->pool.put(q); c->pool.put(buf); };
#define HIPCHK2(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { c->err = std::string(#x) + ": " + hipGetErrorString(e_); cleanup(); return LFPLUS_E_HIP; } } while (0)
    {
        std::vector<u64> rM((size_t)nvars * 32);
        for (u32 k = 0; k < nvars; k++)
            for (int w = 0; w < 16; w++) {
                rM[(size_t)k * 32 + w] = to_mont(r_a[(size_t)k * 16 + w]);
                rM[(size_t)k * 32 + 16 + w] = to_mont(r_b[(size_t)k * 16 + w]);
            }
        HIPCHK2(hipMemcpyAsync(drM, rM.data(), rM.size() * 8, hipMemcpyHostToDevice, c->st));
        HIPCHK2(hipStreamSynchronize(c->st));
    }
    lfp::launch_decompose2(c->f, vw, B, dF0, dF1, c->st);
    for (int s = 0; s < 2; s++) {
        const u64 *Fi = s ? dF1 : dF0;
        lfp::launch_replicate(Fi + r0w, lw, 2, tab + (size_t)(s * (1 + nm)) * 2 * lw, c->st);
    }
    for (u32 j = 0; j < nm; j++) {
        u32 *drp = nullptr, *dci = nullptr;
        u64 *dv = nullptr, *dy = nullptr;
        if (resident) {
            dy = (u64 *)c->pool.get(lw * 8); if (!dy) { cleanup(); return fail(c, LFPLUS_E_HIP, "hipMalloc"); } tofree.push_back(dy);
            const LfpMatrix &mj = c->mats[j];
            for (int s = 0; s < 2; s++) {
                lfp::launch_spmv_ring(mj.rowptr + c->row0, mj.col, mj.spmv_vals(), s ? dF1 : dF0, nl, dy, c->st, mj.const_coef);
                lfp::launch_replicate(dy, lw, 2, tab + (size_t)(s * (1 + nm) + 1 + j) * 2 * lw, c->st);
            }
            continue;
        }
        const u32 nnz = rowptr[j][n];
        HIPCHK2(lfp_dev_malloc(&drp, (n + 1) * 4)); tofree.push_back(drp);
        HIPCHK2(lfp_dev_malloc(&dci, (size_t)(nnz ? nnz : 1) * 4)); tofree.push_back(dci);
        HIPCHK2(lfp_dev_malloc(&dv, (size_t)(nnz ? nnz : 1) * 16 * 8)); tofree.push_back(dv);
        HIPCHK2(lfp_dev_malloc(&dy, lw * 8)); tofree.push_back(dy);
        std::vector<u64> vM((size_t)nnz * 16);
        for (size_t i = 0; i < vM.size(); i++) vM[i] = to_mont(val[j][i]);
        HIPCHK2(hipMemcpyAsync(drp, rowptr[j], (n + 1) * 4, hipMemcpyHostToDevice, c->st));
        HIPCHK2(hipMemcpyAsync(dci, col[j], (size_t)nnz * 4, hipMemcpyHostToDevice, c->st));
        HIPCHK2(hipMemcpyAsync(dv, vM.data(), vM.size() * 8, hipMemcpyHostToDevice, c->st));
        HIPCHK2(hipStreamSynchronize(c->st));   // vM is a local buffer
        for (int s = 0; s < 2; s++) {
            lfp::launch_spmv_ring(drp + c->row0, dci, dv, s ? dF1 : dF0, nl, dy, c->st);
            lfp::launch_replicate(dy, lw, 2, tab + (size_t)(s * (1 + nm) + 1 + j) * 2 * lw, c->st);
        }
    }
    // fix_variables, variable 0 first, all T tables at once
    {
        u64 *cur = tab, *nxt = ping;
        size_t len = nl;
        for (u32 k = 0; k < nvars; k++) {
            if (c->sharded() && len == 1) {       // one entry per table left on every rank: gather them (entry index = rank: the high bits) and finish replicated
                std::vector<u64> mine((size_t)T * 16), all, re((size_t)T * c->world * 16);
                HIPCHK2(hipMemcpyAsync(mine.data(), cur, mine.size() * 8, hipMemcpyDeviceToHost, c->st));
                HIPCHK2(hipStreamSynchronize(c->st));
                if ((rc = lfp_allgather(c, mine.data(), mine.size(), all))) { cleanup(); return rc; }
                for (u32 t = 0; t < T; t++)
                    for (int g = 0; g < c->world; g++) memcpy(&re[((size_t)t * c->world + g) * 16], &all[((size_t)g * T + t) * 16], 16 * 8);
                HIPCHK2(hipMemcpyAsync(gath, re.data(), re.size() * 8, hipMemcpyHostToDevice, c->st));
                HIPCHK2(hipStreamSynchronize(c->st));
                cur = gath; nxt = tab;
                len = (size_t)c->world;
            }
            bool const_r = true;      // this variable's two coordinates are constants (always, for PlusProver's points): one product per word instead of 16
            for (int w = 1; w < 16 && const_r; w++) const_r = !r_a[(size_t)k * 16 + w] && !r_b[(size_t)k * 16 + w];
            lfp::launch_ring_fix(cur, nxt, T, len, drM + (size_t)k * 32, c->st, const_r);
            std::swap(cur, nxt);
            if (nxt == gath) nxt = ping;
            len /= 2;
        }
        // cur: T ring elements, table index = (s*(1+nm) + j)*2 + point
        if (v0) HIPCHK2(hipMemcpyAsync(v0, cur, (size_t)(1 + nm) * 2 * 16 * 8, hipMemcpyDeviceToHost, c->st));
        if (v1) HIPCHK2(hipMemcpyAsync(v1, cur + (size_t)(1 + nm) * 2 * 16, (size_t)(1 + nm) * 2 * 16 * 8, hipMemcpyDeviceToHost, c->st));
    }
    // commitments of the two parts
    {
        Plan p = plan_for(c->nloc, c->kappa, 0);
        rc = ensure_part(c, (size_t)p.nblk * p.nout_f + 2 * p.nout_f);
        if (rc) { cleanup(); return rc; }
        u64 *res = c->part + (size_t)p.nblk * p.nout_f;
        for (int s = 0; s < 2; s++) {
            enqueue_phase1(c, (s ? dF1 : dF0) + r0w, 2, 0, p, c->part);
            lfp::launch_reduce(c->part, p.nblk, (u32)p.nout_f, res + (size_t)s * p.nout_f, 0, c->kappa, 0, 2, 0, nullptr, c->st);
        }
        if ((rc = xsum_dev(c, res, 2 * p.nout_f))) { cleanup(); return rc; }
        if (C0) HIPCHK2(hipMemcpyAsync(C0, res, p.nout_f * 8, hipMemcpyDeviceToHost, c->st));
        if (C1) HIPCHK2(hipMemcpyAsync(C1, res + p.nout_f, p.nout_f * 8, hipMemcpyDeviceToHost, c->st));
    }
    if (F0) HIPCHK2(hipMemcpyAsync(F0, dF0, vw * 8, hipMemcpyDeviceToHost, c->st));
    if (F1) HIPCHK2(hipMemcpyAsync(F1, dF1, vw * 8, hipMemcpyDeviceToHost, c->st));
    lfplus_ctx *publish[2] = {nullptr, nullptr};
    for (int s2 = 0; s2 < 2; s2++) {     // the parts as resident witnesses of the receiving contexts (all on c's stream: the source of F0 / F1 -- c->f -- was read above)
        lfplus_ctx *d = s2 ? dst1 : dst0;
        if (!d) continue;
        const bool reuse = d->f && (d->nf == n || (s2 && d == dst0));
        if (!reuse) {
            if (d->f) { d->own_free(d->f); d->f = nullptr; }
            d->nf = 0;
            HIPCHK2(d->own_alloc(&d->f, vw * 8));
        }
        d->have = false;
        d->nf = 0;                        // no resident witness until the copy has COMPLETED: a failure below must not leave a length-n witness of undefined content
        HIPCHK2(hipMemcpyAsync(d->f, s2 ? dF1 : dF0, vw * 8, hipMemcpyDeviceToDevice, c->st));
        publish[s2] = d;
    }
    HIPCHK2(hipStreamSynchronize(c->st));
    HIPCHK2(hipGetLastError());
    for (lfplus_ctx *d : publish) if (d) d->nf = n;
#undef HIPCHK2
    cleanup();
    return LFPLUS_OK;
}
extern "C" int lfplus_decompose(lfplus_ctx *c, uint64_t B, const uint64_t *r_a, const uint64_t *r_b, uint32_t nm, const uint32_t *const *rowptr,
                                const uint32_t *const *col, const uint64_t *const *val, uint64_t *F0, uint64_t *F1, uint64_t *C0, uint64_t *C1, uint64_t *v0,
                                uint64_t *v1) {
    return decompose_impl(c, B, r_a, r_b, nm, rowptr, col, val, F0, F1, C0, C1, v0, v1, nullptr, nullptr);
}
extern "C" int lfplus_decompose_resident(lfplus_ctx *c, uint64_t B, const uint64_t *r_a, const uint64_t *r_b, uint32_t nm, const uint32_t *const *rowptr,
                                         const uint32_t *const *col, const uint64_t *const *val, lfplus_ctx *dst0, lfplus_ctx *dst1, uint64_t *C0, uint64_t *C1,
                                         uint64_t *v0, uint64_t *v1) {
    if (!dst0 || !dst1) return fail(c, LFPLUS_E_ARG, "lfplus_decompose_resident: null receiving context");
    return decompose_impl(c, B, r_a, r_b, nm, rowptr, col, val, nullptr, nullptr, C0, C1, v0, v1, dst0, dst1);
}
extern "C" int lfplus_get_witness(lfplus_ctx *c, uint64_t *f_out, uint64_t n) {
    if (!c || !f_out) return fail(c, LFPLUS_E_ARG, "lfplus_get_witness: null argument");
    if (!c->f || c->nf != n) return fail(c, LFPLUS_E_ARG, "lfplus_get_witness: no resident witness of this length");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(f_out, c->f, (size_t)n * 16 * 8, hipMemcpyDeviceToHost, c->st));
    HIPCHK(c, hipStreamSynchronize(c->st));
    return LFPLUS_OK;
}
extern "C" int lfplus_tensor(lfplus_ctx *c, const uint64_t *r, uint32_t n, uint64_t *out) {
    if (!c || !out || (n && !r) || n > 28) return fail(c, LFPLUS_E_ARG, "lfplus_tensor: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    u64 *buf = nullptr;
    size_t len = (size_t)1 << n;
    HIPCHK(c, lfp_dev_malloc(&buf, 2 * len * 8));
    u64 *cur = buf, *nxt = buf + len, one = 1;
    (void)hipMemcpyAsync(cur, &one, 8, hipMemcpyHostToDevice, c->st);
    for (u32 i = 0; i < n; i++) {
        lfp::launch_tensor_level(cur, (u64)1 << i, r[i] % lfp::P, nxt, c->st);
        std::swap(cur, nxt);
    }
    (void)hipMemcpyAsync(out, cur, len * 8, hipMemcpyDeviceToHost, c->st);
    hipError_t e = hipStreamSynchronize(c->st);
    (void)hipFree(buf);
    HIPCHK(c, e);
    return LFPLUS_OK;
}
extern "C" int lfplus_tensor_product(lfplus_ctx *c, const uint64_t *a, uint64_t m, const uint64_t *b, uint64_t n, uint64_t *out) {
    if (!c || !out || (m && !a) || (n && !b)) return fail(c, LFPLUS_E_ARG, "lfplus_tensor_product: null argument");
    if (!m || !n) {   // an empty side returns the other (utils.rs:52-57)
        const u64 *src = m ? a : b;
        for (u64 i = 0; i < m + n; i++) out[i] = src[i];
        return LFPLUS_OK;
    }
    if (m * n > (1ull << 30)) return fail(c, LFPLUS_E_ARG, "lfplus_tensor_product: too large");
    HIPCHK(c, hipSetDevice(c->device));
    u64 *buf = nullptr;
    HIPCHK(c, lfp_dev_malloc(&buf, (m + n + m * n) * 8));
    std::vector<u64> ha(a, a + m), hb(b, b + n);
    for (auto &x : ha) x %= lfp::P;
    for (auto &x : hb) x %= lfp::P;
    (void)hipMemcpyAsync(buf, ha.data(), m * 8, hipMemcpyHostToDevice, c->st);
    (void)hipMemcpyAsync(buf + m, hb.data(), n * 8, hipMemcpyHostToDevice, c->st);
    lfp::launch_tensor_product(buf, m, buf + m, n, buf + m + n, c->st);
    (void)hipMemcpyAsync(out, buf + m + n, m * n * 8, hipMemcpyDeviceToHost, c->st);
    hipError_t e = hipStreamSynchronize(c->st);
    (void)hipFree(buf);
    HIPCHK(c, e);
    return LFPLUS_OK;
}

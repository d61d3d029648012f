// lfp_ctx.h -- the context of the LatticeFold+ slice (include/lfplus.h), shared by lfp_capi.cpp and lfp_protocol.cpp
#pragma once
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <memory>
#include <string>
#include <vector>
#include "../../include/lfplus.h"
#include "lf_dist.h"
#include "lfp_kernels.h"

using lfp::u32;
using lfp::u64;

// one n x n sparse matrix with ring coefficients, resident in both orientations
struct LfpMatrix {
    u32 *rowptr = nullptr, *col = nullptr;      // CSR
    u64 *valM = nullptr;                        //   Montgomery coefficients (launch_spmv_ring)
    u32 *colptr = nullptr, *rowidx = nullptr;   // CSC
    u64 *valT = nullptr;                        //   canonical coefficients (launch_spmvT_eq)
    size_t nnz = 0;
    bool const_coef = false;                    // every coefficient is a constant polynomial (launch_spmv_ring's one-product path)
    u64 *valMc = nullptr, *valTc = nullptr;     //   const_coef: the constant terms alone, one word per non-zero (valM / valT hold a 128-byte ring element per non-zero,
                                                //   of which the one-product kernels would fetch a whole line for 8 bytes)
    const u64 *spmv_vals() const { return const_coef ? valMc : valM; }
    void release() {
        for (void *p : {(void *)rowptr, (void *)col, (void *)valM, (void *)colptr, (void *)rowidx, (void *)valT, (void *)valMc, (void *)valTc})
            if (p) (void)hipFree(p);
        rowptr = col = colptr = rowidx = nullptr; valM = valT = valMc = valTc = nullptr; nnz = 0;
    }
};

// Idle scratch blocks of DESTROYED contexts, per device, for the whole process: a prover is a handful of contexts, and a caller that builds one prover per proof (the
// reference's benches do; so does bench.py) would otherwise pay hipMalloc for every table of every prove -- ~30 allocations, 17 GB at 2^20 rows, several milliseconds
// inside the timed call.  A context's pool looks here before it asks the driver, and hands its blocks over when the context is destroyed.  Bounded PER DEVICE:
// LFPLUS_CACHE_GB when set (0 switches the cache off), otherwise a quarter of that device's memory (hipMemGetInfo total) and never more than 32 GB -- the memory is
// invisible to every other allocator in the process, so it must stay a minority share on any part.  lfplus_scratch_trim() releases everything; the main path's
// allocators (lf_common.h lf_dev_malloc) and this slice's own (lfp_dev_malloc below) call it before they report out-of-memory.  Blocks are never freed at process
// exit (the runtime may be gone by then).
struct LfpDevCache {
    struct Blk { void *p; size_t bytes; int device; };
    std::mutex mu;
    std::vector<Blk> blks;
    std::vector<size_t> total, cap;                          // per device; cap[d] == SIZE_MAX: not asked yet
    long env_gb = -1;
    LfpDevCache() {
        const char *e = getenv("LFPLUS_CACHE_GB");
        if (e) env_gb = atol(e) > 0 ? atol(e) : 0;
    }
    size_t cap_of(int device) {                              // (mu held)
        if (device < 0) return 0;
        if ((size_t)device >= cap.size()) { cap.resize((size_t)device + 1, SIZE_MAX); total.resize((size_t)device + 1, 0); }
        if (cap[(size_t)device] == SIZE_MAX) {
            if (env_gb >= 0) cap[(size_t)device] = (size_t)env_gb << 30;
            else {
                size_t fr = 0, tot = 0;
                int cur = 0;
                (void)hipGetDevice(&cur);
                const bool ok = hipSetDevice(device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess;
                (void)hipSetDevice(cur);
                const size_t quarter = ok ? tot / 4 : 0, lim = (size_t)32 << 30;
                cap[(size_t)device] = quarter < lim ? quarter : lim;
            }
        }
        return cap[(size_t)device];
    }
    size_t held(int device) { std::lock_guard<std::mutex> g(mu); return device >= 0 && (size_t)device < total.size() ? total[(size_t)device] : 0; }
    static LfpDevCache &inst() { static LfpDevCache *c = new LfpDevCache; return *c; }
    void *take(int device, size_t bytes, size_t *got) {      // best fit, at most twice the request (as the pools)
        std::lock_guard<std::mutex> g(mu);
        int best = -1;
        for (size_t i = 0; i < blks.size(); i++)
            if (blks[i].device == device && blks[i].bytes >= bytes && blks[i].bytes <= 2 * bytes + (1u << 16) && (best < 0 || blks[i].bytes < blks[(size_t)best].bytes)) best = (int)i;
        if (best < 0) return nullptr;
        void *p = blks[(size_t)best].p;
        *got = blks[(size_t)best].bytes;
        total[(size_t)device] -= *got;
        blks.erase(blks.begin() + best);
        return p;
    }
    void give(int device, void *p, size_t bytes) {
        {
            std::lock_guard<std::mutex> g(mu);
            const size_t lim = cap_of(device);
            if (device >= 0 && total[(size_t)device] + bytes <= lim) { blks.push_back({p, bytes, device}); total[(size_t)device] += bytes; return; }
        }
        (void)hipFree(p);
    }
    void trim(int device) {                                  // device < 0: every device
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < blks.size();)
                if (device < 0 || blks[i].device == device) { drop.push_back(blks[i]); total[(size_t)blks[i].device] -= blks[i].bytes; blks.erase(blks.begin() + (long)i); } else i++;
        }
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (Blk &b : drop) { (void)hipSetDevice(b.device); (void)hipFree(b.p); }
        (void)hipSetDevice(cur);
    }
};

// hipMalloc of this slice outside the pools (matrices, per-call tables): out of memory => release the idle cache of the device and retry once
template <class T> static inline hipError_t lfp_dev_malloc(T **p, size_t bytes) {
    hipError_t e = hipMalloc((void **)p, bytes);
    if (e != hipErrorOutOfMemory) return e;
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return e;
    LfpDevCache::inst().trim(dev);
    return hipMalloc((void **)p, bytes);
}

// Scratch pool of a context: the protocol stages allocate their tables (up to ~1 GB at n = 2^18) anew in every call, and hipMalloc / hipFree of such blocks
// cost milliseconds each; freed blocks are kept and handed out again (best fit, at most twice the request).  One thread per context at a time.
struct LfpPool {
    struct Blk { void *p; size_t bytes; bool busy; };
    std::vector<Blk> blks;
    int device = 0;
    void *get(size_t bytes) {
        if (!bytes) bytes = 8;
        int best = -1;
        for (size_t i = 0; i < blks.size(); i++)
            if (!blks[i].busy && blks[i].bytes >= bytes && blks[i].bytes <= 2 * bytes + (1u << 16) && (best < 0 || blks[i].bytes < blks[(size_t)best].bytes)) best = (int)i;
        if (best >= 0) { blks[(size_t)best].busy = true; return blks[(size_t)best].p; }
        size_t got = 0;
        if (void *q = LfpDevCache::inst().take(device, bytes, &got)) { blks.push_back({q, got, true}); return q; }
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {      // out of memory: drop the idle blocks (the process-wide cache first) and retry once
            LfpDevCache::inst().trim(device);
            if (hipMalloc(&p, bytes) != hipSuccess) {
                for (size_t i = 0; i < blks.size();)
                    if (!blks[i].busy) { (void)hipFree(blks[i].p); blks.erase(blks.begin() + (long)i); } else i++;
                if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
            }
        }
        blks.push_back({p, bytes, true});
        return p;
    }
    void put(void *p) {
        for (Blk &b : blks) if (b.p == p) { b.busy = false; return; }
        if (p) (void)hipFree(p);
    }
    void clear() {                                           // (the context is being destroyed, its stream drained: the blocks go to the process-wide cache)
        for (Blk &b : blks) LfpDevCache::inst().give(device, b.p, b.bytes);
        blks.clear();
    }
};

// Column sharding of one prover over `world` ranks, one GPU each (BASELINE configs[4]): rank g owns rows [g n / world, (g + 1) n / world) of every
// n-indexed object -- the high index bits, so sumcheck pairs (2j, 2j + 1) stay local for the first log2(n / world) rounds.  The transport is the main path's
// (lf_dist.h: RCCL communicator or a host callback); the contexts of one PlusProver share it together with the commitment matrix.
struct LfpShard {
    lfdist::Comm comm;
    ~LfpShard() { comm.destroy(); }
};

struct lfplus_ctx {
    int device = 0;
    hipStream_t st = nullptr;
    std::string err;
    u64 *A = nullptr, *f = nullptr;   // sharded: A holds the rank's columns (kappa x nloc); f is the WHOLE witness (the host hands it to every rank)
    u32 kappa = 0;
    u64 n = 0, nf = 0;                // n: the GLOBAL width
    std::shared_ptr<LfpShard> sh;
    int rank = 0, world = 1;
    u64 nloc = 0, row0 = 0;           // this rank's rows [row0, row0 + nloc); = (n, 0) unsharded.  Local in a sharded context: A, Df ([k][nloc][16]), g; whole: f, tau, mtau
    bool sharded() const { return world > 1; }
    // results of the last from_f
    int8_t *Df = nullptr, *mtau = nullptr;
    u64 *comMf = nullptr, *tau = nullptr, *coms = nullptr;   // comMf: comM_f (k, kappa, 16, 16) | cm_f; coms: C_Mf | cm_mtau (kappa*16 words each)
    u32 k = 0, l = 0;
    size_t Df_cap = 0, comMf_cap = 0;
    // partial sums
    u64 *part = nullptr;
    size_t part_cap = 0;
    u32 *err_d = nullptr;
    bool have = false;
    // lfplus_rg_from_f_async: the double commitment of the resident witness in flight on a second stream (it needs no challenge: PlusProver issues it while the
    // linearization's latency-bound rounds run on `st`); lfplus_rg_from_f with the same parameters collects it, anything else that touches the buffers joins it first
    hipStream_t st2 = nullptr;
    hipEvent_t ev_ff = nullptr;
    bool ff_pending = false;
    u64 ff_b = 0;
    u32 ff_k = 0, ff_l = 0;
    // the folded witness of the last lfplus_cm_prove (cm.rs:164-181): n ring elements
    u64 *g = nullptr;
    u64 g_n = 0;
    bool g_valid = false;   // false once lfplus_mlin has summed the instances' g into ctxs[0]->g (it is then the resident witness f, not g_0)
    LfpPool pool;
    // the context's own long-lived buffers (the from_f results, the witness, g) come from the pool too and stay busy until they are replaced or the context dies:
    // with the pool they end up in the process-wide cache instead of going back to the driver (a prover per proof allocated ~10 of them inside every prove)
    template <class T> hipError_t own_alloc(T **p, size_t bytes) { *p = (T *)pool.get(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
    void own_free(void *p) { if (p) pool.put(p); }
    // pinned host staging of the per-round partial sums and other small downloads (a copy into pageable memory is staged and synchronised by
    // the runtime: tens of microseconds per sumcheck round)
    // -- and the round kernels write their block partials straight into it (mapped: no copy command between the kernel and the host's read)
    u64 *hpin = nullptr, *hpin_dev = nullptr;
    size_t hpin_words = 0;
    u64 *pin(size_t words) {
        if (words <= hpin_words) return hpin;
        if (hpin) (void)hipHostFree(hpin);
        hpin = nullptr; hpin_dev = nullptr; hpin_words = 0;
        if (words < 65536) words = 65536;
        if (hipHostMalloc((void **)&hpin, words * 8, hipHostMallocMapped) != hipSuccess) { hpin = nullptr; return nullptr; }
        if (hipHostGetDevicePointer((void **)&hpin_dev, hpin, 0) != hipSuccess) { (void)hipHostFree(hpin); hpin = nullptr; hpin_dev = nullptr; return nullptr; }
        hpin_words = words;
        return hpin;
    }
    // The commitment matrix and the constraint-system matrices may be shared between contexts (lfplus_share_matrix / lfplus_share_matrices: PlusProver keeps
    // one context per instance and one copy of each): the device allocations are reference-counted, so a sharer stays valid when the context that uploaded
    // them re-uploads or is destroyed -- the memory goes back to the driver with the last holder.  `A` and `mats` are the raw views the kernels take.
    std::shared_ptr<void> A_ref;
    struct MatsOwner {
        std::vector<LfpMatrix> m;
        ~MatsOwner() { for (LfpMatrix &x : m) x.release(); }
    };
    std::shared_ptr<MatsOwner> mats_ref;
    std::vector<LfpMatrix> mats;   // lfplus_set_matrices: the constraint-system matrices, uploaded once
    u64 mats_n = 0;
    void drop_mats() {
        mats_ref.reset();
        mats.clear();
        mats_n = 0;
    }
};

#define HIPCHK(c, x)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) {                                                             \
            (c)->err = std::string(#x) + ": " + hipGetErrorString(e_);                      \
            return LFPLUS_E_HIP;                                                            \
        }                                                                                   \
    } while (0)

static inline int fail(lfplus_ctx *c, int rc, const std::string &m) {
    if (c) c->err = m;
    return rc;
}
static inline bool canonical(const u64 *w, size_t n) {
    for (size_t i = 0; i < n; i++)
        if (w[i] >= lfp::P) return false;
    return true;
}
// ---- exchanges of a sharded prover (no-ops when world == 1).  Every small exchange is "all-gather `words` u64 per rank, add the world vectors mod p" on host
// buffers (the round messages and evaluations are summed on the host anyway); the two large ones (h and the folded g, n ring elements) are device all-gathers.
static inline u64 lfp_addp(u64 a, u64 b) { unsigned __int128 s = (unsigned __int128)a + b; return (u64)(s >= lfp::P ? s - lfp::P : s); }
static inline int lfp_xsum(lfplus_ctx *c, u64 *v, size_t words) {
    if (!c->sharded() || !words) return LFPLUS_OK;
    std::vector<u64> all((size_t)c->world * words);
    if (c->sh->comm.allgather_host(v, all.data(), words, c->st) != 0) return fail(c, LFPLUS_E_HIP, "sharded prover: exchange failed");
    for (size_t i = 0; i < words; i++) {
        u64 s = 0;
        for (int g = 0; g < c->world; g++) s = lfp_addp(s, all[(size_t)g * words + i]);
        v[i] = s;
    }
    return LFPLUS_OK;
}
// every rank's `words` (host) -> all[world][words]
static inline int lfp_allgather(lfplus_ctx *c, const u64 *mine, size_t words, std::vector<u64> &all) {
    all.assign((size_t)c->world * words, 0);
    if (c->sh->comm.allgather_host(mine, all.data(), words, c->st) != 0) return fail(c, LFPLUS_E_HIP, "sharded prover: exchange failed");
    return LFPLUS_OK;
}
// device all-gather of the ranks' row slices (words per rank) into the whole vector, in row order
static inline int lfp_allgather_dev(lfplus_ctx *c, const u64 *mine_dev, u64 *whole_dev, size_t words) {
    if (c->sh->comm.allgather_dev(mine_dev, whole_dev, words, c->st) != 0) return fail(c, LFPLUS_E_HIP, "sharded prover: device all-gather failed");
    return LFPLUS_OK;
}


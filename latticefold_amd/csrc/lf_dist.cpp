// lf_dist.cpp -- see lf_dist.h
#include "lf_dist.h"

#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <vector>

namespace lfdist {

namespace {
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    bool ok = false;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // prefer a copy that is already mapped (PyTorch brings its own librccl.so): one RCCL runtime per process
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.h) break;
        }
        if (!r.h)
            for (const char *name : {"librccl.so.1", "librccl.so"}) {
                r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.h) break;
            }
        if (!r.h) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
        r.AllGather = (decltype(r.AllGather))dlsym(r.h, "ncclAllGather");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.h, "ncclCommAbort");
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
    });
    return r;
}
struct Stopwatch {
    Comm &c;
    std::chrono::steady_clock::time_point t0;
    explicit Stopwatch(Comm &cc) : c(cc), t0(std::chrono::steady_clock::now()) {}
    ~Stopwatch() {
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        c.n_exchanges++; c.us_total += us; if (us > c.us_max) c.us_max = us;
    }
};
}  // namespace

int rccl_unique_id(uint8_t *id128) {
    Rccl &r = rccl();
    if (!r.ok || !id128) return LF_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return LF_ERR_HIP;
    static_assert(sizeof(id.internal) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, id.internal, 128);
    return LF_OK;
}
int rccl_init(Comm &c, int rank, int world, const uint8_t *id128) {
    Rccl &r = rccl();
    if (!r.ok) return LF_ERR_UNSUPPORTED;
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclComm_t comm = nullptr;
    if (r.CommInitRank(&comm, world, id, rank) != ncclSuccess) return LF_ERR_HIP;
    c.rank = rank; c.world = world; c.cb = nullptr; c.user = nullptr; c.nccl = comm; c.poisoned = false; c.model = false;
    return LF_OK;
}
int Comm::ensure_stage(size_t words) {
    if (words > d_stage_words) {
        if (d_stage) (void)hipFree(d_stage);
        d_stage = nullptr; d_stage_words = 0;
        if (hipMalloc((void **)&d_stage, words * 8) != hipSuccess) return LF_ERR_HIP;
        d_stage_words = words;
    }
    if (words > h_stage_words) {
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr; h_stage_words = 0;
        if (hipHostMalloc((void **)&h_stage, words * 8) != hipSuccess) return LF_ERR_HIP;
        h_stage_words = words;
    }
    return LF_OK;
}
int Comm::allgather_dev(const uint64_t *send_dev, uint64_t *recv_all_dev, size_t words, hipStream_t s) {
    if (poisoned) return LF_ERR_STATE;
    if (world <= 1 && !nccl) return hipMemcpyAsync(recv_all_dev, send_dev, words * 8, hipMemcpyDeviceToDevice, s) == hipSuccess ? LF_OK : LF_ERR_HIP;
    Stopwatch sw(*this);
    words_sent += words;
    if (model) {   // timing model: zeros from the absent peers, this rank's words in its slot -- device-side and in-stream like the RCCL path
        if (hipMemsetAsync(recv_all_dev, 0, (size_t)world * words * 8, s) != hipSuccess) return LF_ERR_HIP;
        return hipMemcpyAsync(recv_all_dev + (size_t)rank * words, send_dev, words * 8, hipMemcpyDeviceToDevice, s) == hipSuccess ? LF_OK : LF_ERR_HIP;
    }
    if (nccl) {   // in-stream: no host synchronisation at all
        return rccl().AllGather(send_dev, recv_all_dev, words, ncclUint64, (ncclComm_t)nccl, s) == ncclSuccess ? LF_OK : LF_ERR_HIP;
    }
    if (!cb) return LF_ERR_STATE;
    const size_t tot = (size_t)world * words;
    int rc = ensure_stage(tot + words);
    if (rc != LF_OK) return rc;
    uint64_t *hs = h_stage, *hr = h_stage + words;
    if (hipMemcpyAsync(hs, send_dev, words * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return LF_ERR_HIP;
    if (cb(user, hs, hr, words) != 0) return LF_ERR_HIP;
    if (hipMemcpyAsync(recv_all_dev, hr, tot * 8, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return LF_ERR_HIP;
    return LF_OK;
}
int Comm::allgather_host(const uint64_t *send, uint64_t *recv_all, size_t words, hipStream_t s) {
    if (poisoned) return LF_ERR_STATE;
    if (world <= 1) { memcpy(recv_all, send, words * 8); return LF_OK; }
    Stopwatch sw(*this);
    words_sent += words;
    if (model) {
        memset(recv_all, 0, (size_t)world * words * 8);
        memcpy(recv_all + (size_t)rank * words, send, words * 8);
        return LF_OK;
    }
    if (nccl) {
        const size_t tot = (size_t)world * words;
        int rc = ensure_stage(tot + words);
        if (rc != LF_OK) return rc;
        uint64_t *ds = d_stage, *dr = d_stage + words;
        memcpy(h_stage, send, words * 8);
        if (hipMemcpyAsync(ds, h_stage, words * 8, hipMemcpyHostToDevice, s) != hipSuccess) return LF_ERR_HIP;
        if (rccl().AllGather(ds, dr, words, ncclUint64, (ncclComm_t)nccl, s) != ncclSuccess) return LF_ERR_HIP;
        if (hipMemcpyAsync(h_stage + words, dr, tot * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return LF_ERR_HIP;
        memcpy(recv_all, h_stage + words, tot * 8);
        return LF_OK;
    }
    if (!cb) return LF_ERR_STATE;
    return cb(user, send, recv_all, words) == 0 ? LF_OK : LF_ERR_HIP;
}
void Comm::abort_peers() {
    // the context keeps its shard geometry (sh_world / sh_rank), so the communicator keeps world / rank too and refuses further exchanges;
    // with the host-callback transport the peers are the host language's to release (the callback has no abort message)
    if (world > 1) poisoned = true;
    if (nccl && rccl().CommAbort) { (void)rccl().CommAbort((ncclComm_t)nccl); nccl = nullptr; }
}
void Comm::destroy() {
    if (nccl) { (void)rccl().CommDestroy((ncclComm_t)nccl); nccl = nullptr; }
    if (d_stage) (void)hipFree(d_stage);
    if (h_stage) (void)hipHostFree(h_stage);
    d_stage = nullptr; h_stage = nullptr; d_stage_words = h_stage_words = 0;
}

}  // namespace lfdist

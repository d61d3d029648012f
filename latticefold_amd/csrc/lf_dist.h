// lf_dist.h -- the exchange layer of the intra-step sharding: one RCCL communicator per context (one rank per GPU,
// xGMI), used with DEVICE buffers on the context's own stream; or a host callback supplied by the host language (tests: gloo).
//
// Every exchange on the path is "all-gather `words` u64 from each rank, then add the `world` vectors mod p" -- RCCL has no modular
// reduction and ncclSum on canonical residues would wrap mod 2^64 -- so the collective is ncclAllGather and the
// reduction a device kernel (lf_kernels.hip: launch_modsum / bb: host).  RCCL is loaded with dlopen at lf_dist_init, so the library
// has no link-time dependency on it and reuses the copy PyTorch may already have mapped.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/lfhip.h"

namespace lfdist {

struct Comm {
    int rank = 0, world = 1;
    lf_exchange_fn cb = nullptr;     // host transport (lf_set_sharding)
    void *user = nullptr;
    void *nccl = nullptr;            // ncclComm_t (lf_dist_init)
    bool model = false;              // lf_set_sharding_model: no peers -- an all-gather puts this rank's words into its slot and zeros into the others (a TIMING
                                     // model of rank `rank` of `world` alone on a GPU: every kernel does a rank's share of the work, the results are not a proof)
    bool poisoned = false;           // set by abort_peers(): every later exchange returns LF_ERR_STATE (the step that failed is lost; shard a fresh context)
    uint64_t *d_stage = nullptr;     // device staging for host-buffer exchanges over RCCL / device-buffer exchanges over the callback
    size_t d_stage_words = 0;
    uint64_t *h_stage = nullptr;     // pinned
    size_t h_stage_words = 0;
    // per-exchange latency log (host wall clock around enqueue + completion)
    uint64_t n_exchanges = 0;
    uint64_t words_sent = 0;         // u64 words this rank contributed, summed over its exchanges (an all-gather delivers (world - 1) x as many to it)
    double us_total = 0, us_max = 0;

    bool active() const { return world > 1; }
    // all-gather `words` u64 per rank: device buffers, ordered on `s`.  recv_all_dev holds world*words words in rank order.
    int allgather_dev(const uint64_t *send_dev, uint64_t *recv_all_dev, size_t words, hipStream_t s);
    // the same for host buffers (blocking)
    int allgather_host(const uint64_t *send, uint64_t *recv_all, size_t words, hipStream_t s);
    void abort_peers();              // a rank that fails mid-step tears the communicator down so that its peers error out instead of hanging
    void destroy();
    int ensure_stage(size_t words);
};

int rccl_unique_id(uint8_t *id128);
int rccl_init(Comm &c, int rank, int world, const uint8_t *id128);

}  // namespace lfdist

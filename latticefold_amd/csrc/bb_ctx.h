// bb_ctx.h -- internal to the BabyBear backend's host side (bb_capi.cpp, bb_prove.cpp): the context (stream, device arena, event timeline, resident matrices
// and tables) and the helpers the two translation units share.  Not part of the C ABI.
#pragma once
#include "bb_capi.h"
#include "lf_sv_rounds.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "bb_kernels.h"
#include "lf_ajtai_i8.h"
#include "lf_common.h"
#include "lf_dist.h"
#include "lf_verify.h"

namespace lfbb {

static const int NPH = LF_N_PHASES;

struct EvPair { hipEvent_t a, b; };

struct BbCtxImpl {
    lf_ctx *owner = nullptr;
    int device = 0;
    hipStream_t st_lane[2] = {nullptr, nullptr};
    int lane = 0;   // 0 = main work, 1 = left decomposition running concurrently (own stream, "lane1:" buffers, own pinned arena)
    std::mutex mu;
    BbHostRing ring;
    DevBb dev;
    fe *d_icrt = nullptr;
    fe *d_icrt_sp_val = nullptr;    // the rows of the inverse CRT map in compressed form ([72][8] values / columns), null when a row has more than 8 entries
    u32 *d_icrt_sp_col = nullptr;
    fe *dA = nullptr;               // the matrix in NTT form while it is being installed (freed once the byte planes are packed)
    unsigned char *dAb = nullptr;   // the same matrix in coefficient form, bytes in int8-MFMA operand order (lf_ajtai_i8.hip); row chunks of <= 16
    u32 i8_nch = 0, i8_kc = 0;
    u32 kappa = 0;
    size_t nA = 0, nA_total = 0, A_col0 = 0;   // columns held by this rank / of the whole matrix / first held column
    // intra-step sharding, same scheme as the Goldilocks backend
    int sh_rank = 0, sh_world = 1;   // mirror comm.rank / comm.world
    lfdist::Comm comm;               // exchange layer (lf_dist.h): RCCL communicator or host callback
    bool have_ccs = false;
    lf_params P{};
    size_t N = 0, m = 0, n = 0;
    std::vector<u32 *> d_rowptr, d_col, d_colptr, d_rowidx;
    std::vector<fe *> d_val, d_valT;
    LinDesc desc{};
    std::map<std::string, DevBuf> bufs;
    u64 *h_pin = nullptr;
    size_t h_pin_words = 0;
    u64 *arena[2] = {nullptr, nullptr};   // pinned staging for the asynchronous decompositions (bump-allocated per step)
    size_t arena_words = 0, arena_used[2] = {0, 0};
    hipEvent_t ev_side[2] = {nullptr, nullptr};
    hipEvent_t ev_prep[2] = {nullptr, nullptr};   // fold prepare: fork / join of the right side's chain on the other stream
    hipEvent_t ev_dec[4] = {nullptr, nullptr, nullptr, nullptr};   // decomposition milestones: [2*side + (0 commit, 1 evaluations)]
    int digit_mode = 0;   // balanced-digit rule (lf_set_digit_mode)
    unsigned fold_split_mask = 0;   // table rounds of the last folding sumcheck in the split eq form (lf_last_fold_split_rounds)
    unsigned sv_round_mask = 0;   // rounds of the last folding sumcheck that ran as int8 GEMMs (bit i-1 = round i; lf_last_fold_paths)
    Tunables tn;          // environment switches, re-read at the start of every linearize / fold_step
    // v_s of the linearized instance computed inside the linearization (v = sum_k 2^k v_s[k]); reused by the right decomposition of the same fold step
    const lf_witness *vs_wit = nullptr;
    const fe *vs_eq = nullptr;
    u64 *vs_dev = nullptr;
    bool vs_keep = false;
    u32 lin_blocks = 0;   // grid bound of the linearization rounds while the commit chain runs on the other lane (0 = none)
    u64 *h_round = nullptr;   // pinned + device-mapped: sumcheck round kernels write their message straight to the host
    u64 *round_out() {
        if (!h_round && hipHostMalloc((void **)&h_round, 5 * RE * 8 * 2, hipHostMallocMapped) != hipSuccess) h_round = nullptr;
        return h_round;
    }
    int sc_round = -1;
    size_t sc_n = 0;
    int sc_cur = 0;
    int sf_round = -1;   // folding-sumcheck ABI state
    size_t sf_n = 0;
    int sf_cur = 0;
    // measurement
    float phase_ms[NPH] = {0};
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<int, size_t>> ev_tags;
    float k_fold_ms = 0, k_ajtai_ms = 0;
    int k_fold_n = 0, k_ajtai_n = 0;
    double host_tr_ms = 0;

    hipStream_t stream() const { return st_lane[lane]; }
    u64 *arena_alloc(size_t words) {   // nullptr when exhausted
        if (arena_used[lane] + words > arena_words) return nullptr;
        u64 *r = arena[lane] + arena_used[lane];
        arena_used[lane] += words;
        return r;
    }
    int buf(const std::string &name, size_t bytes, void **out) {
        DevBuf &b = bufs[lane ? "lane1:" + name : name];
        int rc = b.ensure(bytes);
        *out = b.p;
        return rc;
    }
    template <class T>
    int tbuf(const std::string &name, size_t count, T **out) {
        void *q;
        int rc = buf(name, count * sizeof(T), &q);
        *out = (T *)q;
        return rc;
    }
    int pin(size_t words) {
        if (words <= h_pin_words) return LF_OK;
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        if (words < 16384) words = 16384;
        if (hipHostMalloc((void **)&h_pin, words * 8) != hipSuccess) return LF_ERR_HIP;
        h_pin_words = words;
        return LF_OK;
    }
    size_t ev_begin(int tag) {
        if (ev_used == ev_pool.size()) {
            EvPair e;
            (void)hipEventCreate(&e.a);
            (void)hipEventCreate(&e.b);
            ev_pool.push_back(e);
        }
        size_t i = ev_used++;
        (void)hipEventRecord(ev_pool[i].a, stream());
        ev_tags.push_back({tag, i});
        return i;
    }
    void ev_end(size_t i) { (void)hipEventRecord(ev_pool[i].b, stream()); }
    void ev_reset() { ev_used = 0; ev_tags.clear(); }
    void ev_collect() {
        (void)hipStreamSynchronize(st_lane[0]);
        (void)hipStreamSynchronize(st_lane[1]);
        k_fold_ms = k_ajtai_ms = 0;
        k_fold_n = k_ajtai_n = 0;
        for (int i = 0; i < NPH; i++) phase_ms[i] = 0;
        for (auto &tg : ev_tags) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev_pool[tg.second].a, ev_pool[tg.second].b);
            if (tg.first == 0) { k_fold_ms += ms; k_fold_n++; }
            else if (tg.first == 1) { k_ajtai_ms += ms; k_ajtai_n++; }
            else if (tg.first >= 10 && tg.first < 10 + NPH) phase_ms[tg.first - 10] += ms;
        }
        phase_ms[6] = (float)host_tr_ms;
    }
};
typedef BbCtxImpl C;

struct HostTimer {
    C *c;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(C *cc) : c(cc), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { c->host_tr_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};


// ---- shared between bb_capi.cpp / bb_prove.cpp (hidden: not part of the ABI) -----------------------------------------------------------------------
#pragma GCC visibility push(hidden)
struct BbMarks {
    bool on = false;
    std::chrono::steady_clock::time_point t0;
    double last = 0;
    void start() { on = getenv("LF_TIMELINE") != nullptr; t0 = std::chrono::steady_clock::now(); last = 0; }
    void mark(const char *what) {
        if (!on) return;
        const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[bb timeline] %-44s at %8.3f ms  (+%7.3f)\n", what, t, t - last);
        last = t;
    }
};
inline BbMarks g_marks;
#define BB_MARK(x) g_marks.mark(x)
static inline H9 h9_load(const u64 *w) { H9 r; for (int i = 0; i < TAU; i++) r.c[i] = w[i] % BB_P; return r; }
static inline H9 h9_one() { H9 r; memset(&r, 0, sizeof(r)); r.c[0] = 1; return r; }
static inline H9 h9_sub(const H9 &a, const H9 &b) { H9 r; for (int i = 0; i < TAU; i++) r.c[i] = hsub(a.c[i], b.c[i]); return r; }
static inline H9 h9_add(const H9 &a, const H9 &b) { H9 r; for (int i = 0; i < TAU; i++) r.c[i] = hadd(a.c[i], b.c[i]); return r; }
static inline H9 h9_scale(const H9 &a, u64 k) { H9 r; for (int i = 0; i < TAU; i++) r.c[i] = hmul(a.c[i], k % BB_P); return r; }
// inverse in F_p[Y]/(Y^9 - nu): solve (multiplication by a) x = 1 by Gaussian elimination on the 9 x 9 matrix M[i][j] = [Y^i](a Y^j); false if a = 0
int build_eq_dev(C *c, const H9 *pt, u32 nv, fe *eq_dev);
int down_small(C *c, const u64 *dsrc, size_t words, u64 *host);
bool h9_inv(const H9 &a, u64 nu, H9 *out);
int exchange_modsum(C *c, u64 *inout, size_t words);
bool is_diag(const u64 *e, H9 *out);
int commit_planes_i8(C *c, const int32_t *planes, size_t ld, u32 k0, u32 NP, u64 *out_dev);
int build_eq_async(C *c, const H9 *pt, u32 nv, fe *eq_dev);
int up_ring(C *c, const u64 *host, size_t n, fe *dst);
int down_ring(C *c, const fe *src, size_t n, u64 *host);
size_t dec_proof_len(const lf_params *p);
size_t lin_proof_len(const lf_params *p);
#pragma GCC visibility pop

}  // namespace lfbb

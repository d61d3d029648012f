// lf_ctx.h -- internal to the Goldilocks backend's host side (lf_capi.cpp, lf_prove.cpp, lf_fold.cpp): the context (streams, lane worker, device arena, event
// timeline, resident matrices and tables), the transcript handle and the helpers the three translation units share.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <thread>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lfhip.h"
#include "bb_capi.h"
#include "lf_common.h"
#include "lf_dist.h"
#include "lf_kernels.h"
#include "lf_verify.h"

using namespace lf;

namespace lf {
void launch_fix_many(const DevCrt &t, const u64 *in, size_t ld_in, u64 *out, size_t ld_out, size_t n_in, u32 rows3, Fq3Const r, hipStream_t s);
}

#include <stdio.h>
#include <stdlib.h>
static bool lf_trace_on() { static int v = -1; if (v < 0) v = getenv("LF_TRACE") ? 1 : 0; return v == 1; }
#define LF_TRACE(c, msg)                                                              \
    do {                                                                              \
        if (lf_trace_on()) {                                                          \
            hipError_t e_ = hipStreamSynchronize((c)->stream());                            \
            fprintf(stderr, "[lf] %s:%d %s -> %s\n", __func__, __LINE__, msg, hipGetErrorString(e_)); \
            fflush(stderr);                                                           \
        }                                                                             \
    } while (0)

inline thread_local int t_lane = 0;  // 0 = caller thread, 1 = helper thread running the left decomposition
constexpr int LF_NLANES = 2;

struct lf_transcript {
    Transcript t;
    lfbb::BbTranscript *bb = nullptr;   // BabyBear transcripts live here (ring 1); t is unused then
    lf_transcript() {}
    lf_transcript(const lf_transcript &o) : t(o.t), bb(o.bb ? new lfbb::BbTranscript(*o.bb) : nullptr) {}
    ~lf_transcript() { delete bb; }
};

// wall-clock timeline of the calling thread (LF_TIMELINE=1): printed at the end of lf_fold_step
struct Timeline {
    bool on;
    std::chrono::steady_clock::time_point t0;
    std::vector<std::pair<const char *, double>> marks, marks1;   // marks1: the helper lane's thread ("L1: ..."), merged by time at the end of the step
    Timeline() : on(getenv("LF_TIMELINE") != nullptr), t0(std::chrono::steady_clock::now()) { marks1.reserve(32); }
    void mark(const char *what) {   // always recorded (lf_last_timeline); printed only with LF_TIMELINE
        marks.push_back({what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()});
    }
    void mark1(const char *what) { marks1.push_back({what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()}); }
    void merge() {
        marks.insert(marks.end(), marks1.begin(), marks1.end());
        marks1.clear();
        std::stable_sort(marks.begin(), marks.end(), [](const std::pair<const char *, double> &a, const std::pair<const char *, double> &b) { return a.second < b.second; });
    }
    void dump() {
        if (!on) return;
        double prev = 0;
        for (auto &m : marks) { fprintf(stderr, "[timeline] %-28s at %8.3f ms  (+%7.3f)\n", m.first, m.second, m.second - prev); prev = m.second; }
    }
};
inline thread_local Timeline *t_tl = nullptr;
#define TL_MARK(x) do { if (t_tl) t_tl->mark(x); } while (0)

static const char *PHASE_NAMES[LF_N_PHASES] = {"linearization", "decomp_crt_commit", "decomp_evals", "fold_prepare",
                                                "fold_sumcheck", "fold_finish", "host_transcript", "total"};

struct EvPair { hipEvent_t a, b; };

// The helper lane of a fold step: ONE thread per context, created at the first step and parked on a condition variable between steps
// (a std::async thread per step cost a thread creation + join every 7-30 ms).
struct LaneWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, stop = false;
    int rc = 0;
    void loop() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || stop; });
            if (stop) return;
            std::function<int()> j = std::move(job);
            has_job = false;
            lk.unlock();
            int r = j();
            lk.lock();
            rc = r;
            done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<int()> j) {
        std::unique_lock<std::mutex> lk(m);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    ~LaneWorker() {
        {
            std::unique_lock<std::mutex> lk(m);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

struct lf_ctx {
    lfbb::BbCtx *bb = nullptr;   // BabyBearRingNTT backend (ring 1): every entry point forwards to it
    int device = 0;
    hipStream_t st_lane[LF_NLANES] = {nullptr, nullptr};
    int digit_mode = 0;   // balanced-digit rule of base-B decompositions (lf_set_digit_mode)
    ExtBasis xb;          // external coordinate basis of F_{p^tau} (lf_set_ext_basis); identity by default
    Tunables tn;          // environment switches, re-read at the start of every lf_linearize / lf_fold_step
    u32 lin_blocks = 0;   // grid bound of the linearization rounds while a fold step's commit chain runs on the other lane (0 = none)
    std::mutex mu, buf_mu, ev_mu;
    hipStream_t st_io = nullptr;   // lane 2: witness ingestion next to a running fold step (lf_witness_from_w_ccs_begin), lowest priority; own buffers ("lane2:" names)
    std::mutex io_mu;              // one ingestion at a time per context
    std::atomic<int> io_jobs{0};   // ingestion jobs whose worker has not finished (lf_ctx_destroy waits for them)
    hipStream_t stream() const { return t_lane == 2 ? st_io : st_lane[t_lane]; }
    // the same facts for either backend (the external-basis marshalling is ring-agnostic)
    bool have_ccs_any() const { return bb ? bb->have_ccs() : have_ccs; }
    const lf_params &params_any() const { return bb ? bb->params() : P; }
    size_t n_any() const { return bb ? bb->dim_n() : n; }
    size_t m_any() const { return bb ? bb->dim_m() : m; }
    size_t N_any() const { return bb ? bb->dim_N() : N; }
    HostRing ring;
    DevCrt dcrt;
    u64 *d_icrt = nullptr;
    u64 *d_icrt_sp_val = nullptr;   // the rows of the inverse CRT map in compressed form ([24][8] values / columns), null when a row has more than 8 entries
    u32 *d_icrt_sp_col = nullptr;
    // Ajtai (nA = columns held by this rank, starting at global column A_col0 of nA_total)
    LaneWorker lane1;
    bool A_loaded = false;
    unsigned char *dAb = nullptr;   // the matrix in coefficient form, bytes in int8-MFMA operand order (lf_ajtai_i8.hip); row chunks of <= 26
    u32 i8_nch = 0, i8_kc = 0;
    u32 kappa = 0;
    size_t nA = 0, nA_total = 0, A_col0 = 0;
    // intra-step sharding: rank/world and the all-gather callback supplied by the host language
    int sh_rank = 0, sh_world = 1;   // mirror comm.rank / comm.world
    int agreed_two_lanes = -1;       // lf_dist_init's handshake: the schedule ALL ranks agreed on (1 threaded / 0 one thread); -1 = no handshake ran (host transports, model)
    bool two_lanes_ok = false;       // the transport's two channels have been seen working concurrently (lf_dist_init's handshake; two host callbacks): a sharded
                                     // step then runs the threaded two-lane schedule unless LF_SHARD_TWO_LANES=0
    // exchange layer, one per lane: the two lanes of a fold step exchange concurrently (lane 0: linearization rounds and right evaluations,
    // lane 1: commits and left evaluations) and collectives of ONE communicator must be issued in the same order on every rank
    lfdist::Comm comm[2];
    lfdist::Comm &cm() { return comm[t_lane]; }
    // CCS
    bool have_ccs = false;
    bool ccs_general = false;   // some constraint matrix has more than ~1.5 entries per (non-empty) row: M z runs on k_spmv_rows (whole-element gathers from an element-major z)
    // sharded step: the columns of z this rank's row slice of the constraint matrices refers to (shard_col_range; (size_t)-1 = not computed)
    size_t shc_r0 = (size_t)-1, shc_rcnt = 0, shc_lo = 0, shc_hi = 0;
    lf_params P{};
    size_t N = 0, m = 0, n = 0;
    std::vector<u32 *> d_rowptr, d_col, d_colptr, d_rowidx;
    std::vector<u64 *> d_val, d_valT;
    LinCombDesc desc{};
    std::map<std::string, DevBuf> bufs;
    u64 *h_pin_lane[LF_NLANES] = {nullptr, nullptr};
    size_t h_pin_words_lane[LF_NLANES] = {0, 0};
    // lin sumcheck ABI state
    int sc_round = -1;
    size_t sc_n = 0;
    int sc_cur = 0;
    int sf_round = -1;   // folding-sumcheck ABI state (lf_sumcheck_fold_*)
    size_t sf_n = 0;
    int sf_cur = 0;
    // measurement
    float phase_ms[LF_N_PHASES] = {0};
    std::vector<std::pair<const char *, double>> tl_marks;   // wall-clock marks of the last fold step (lf_last_timeline)
    std::vector<EvPair> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<int, size_t>> ev_tags;  // (tag, event index)
    float k_fold_ms = 0, k_ajtai_ms = 0;
    int k_fold_n = 0, k_ajtai_n = 0;
    double host_tr_ms = 0;
    // v_s of the linearized instance computed inside the linearization (v = sum_k 2^k v_s[k]); reused by the right decomposition of the same step
    const lf_witness *vs_wit = nullptr;
    bool vs_keep = false;            // set by the fold step around its linearization: only there the decomposition that follows uses the same point
    const u64 *vs_eq = nullptr;
    u64 *vs_dev = nullptr;
    // bit-plane forms of the two witnesses of the running fold step (lf_sv_rounds.h), enqueued on the helper lane's stream before anything else
    const lf_witness *bits_wit[2] = {nullptr, nullptr};
    u32 *bits_ptr[2] = {nullptr, nullptr};
    hipEvent_t bits_ev[2] = {nullptr, nullptr};
    hipEvent_t ev_prep[2] = {nullptr, nullptr};   // fold prepare: fork / join of the right side's chain on the helper lane's stream
    hipEvent_t ev_yR = nullptr, ev_yL = nullptr;  // the right / left commit's results are in h_pin2 (second / first half)
    u64 *h_pin2 = nullptr;
    size_t h_pin2_words = 0;
    int pin2(size_t words) {
        if (words <= h_pin2_words) return LF_OK;
        if (h_pin2) (void)hipHostFree(h_pin2);
        h_pin2 = nullptr; h_pin2_words = 0;
        if (hipHostMalloc((void **)&h_pin2, words * 8) != hipSuccess) return LF_ERR_HIP;
        h_pin2_words = words;
        return LF_OK;
    }
    // linearization: the pass of the v_s evaluations over the witness starts on this stream while the last sumcheck rounds are still running (VsSplit)
    hipStream_t st_aux = nullptr;
    hipEvent_t ev_aux = nullptr;
    u64 *h_aux = nullptr;   // pinned, 1 KB: the known part of the point
    unsigned sv_round_mask = 0;      // rounds of the last folding sumcheck that ran as int8 GEMMs (bit i-1 = round i)
    unsigned fold_split_mask = 0;    // table rounds of the last folding sumcheck that ran in the split eq form (bit i-1 = round i)
    unsigned lin_split_rounds = 0;   // rounds of the last linearization sumcheck that ran in the split eq form (run_lin_sumcheck)


    int buf(const std::string &name, size_t bytes, void **out) {
        DevBuf *b;
        {
            std::lock_guard<std::mutex> g(buf_mu);
            b = &bufs[t_lane ? (t_lane == 1 ? "lane1:" : "lane2:") + name : name];  // std::map nodes are stable
        }
        int rc = b->ensure(bytes);
        *out = b->p;
        return rc;
    }
    // give a set-up scratch buffer back (caller has synchronised the stream that used it)
    void drop_buf(const std::string &name) {
        std::lock_guard<std::mutex> g(buf_mu);
        auto it = bufs.find(t_lane ? (t_lane == 1 ? "lane1:" : "lane2:") + name : name);
        if (it != bufs.end()) { it->second.release(); bufs.erase(it); }
    }
    template <class T>
    int tbuf(const std::string &name, size_t count, T **out) {
        void *p;
        int rc = buf(name, count * sizeof(T), &p);
        *out = (T *)p;
        return rc;
    }
    // Small host-to-device uploads inside a step (challenge powers, look-up tables, evaluation points) go through a pinned ring per lane:
    // the copy is truly asynchronous and the caller's stack / vector buffer is free at once -- no stream synchronisation per upload.
    unsigned char *stage[LF_NLANES] = {nullptr, nullptr};
    size_t stage_off[LF_NLANES] = {0, 0};
    static constexpr size_t STAGE_BYTES = (size_t)1 << 20;
    int h2d_small(void *dst, const void *src, size_t bytes) {
        unsigned char *&ring = stage[t_lane];
        if (!ring && hipHostMalloc((void **)&ring, STAGE_BYTES, hipHostMallocDefault) != hipSuccess) { ring = nullptr; return LF_ERR_HIP; }
        const size_t need = (bytes + 63) & ~(size_t)63;
        if (need > STAGE_BYTES) {   // not small: plain blocking copy
            HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream()));
            HIPCHK(hipStreamSynchronize(stream()));
            return LF_OK;
        }
        if (stage_off[t_lane] + need > STAGE_BYTES) {   // wrap: everything staged so far must have left the ring
            HIPCHK(hipStreamSynchronize(stream()));
            stage_off[t_lane] = 0;
        }
        unsigned char *slot = ring + stage_off[t_lane];
        stage_off[t_lane] += need;
        memcpy(slot, src, bytes);
        HIPCHK(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, stream()));
        return LF_OK;
    }
    u64 *h_round[LF_NLANES] = {nullptr, nullptr};   // pinned + device-mapped: sumcheck round kernels write their message straight to the host
    u64 *round_out() {
        u64 *&p = h_round[t_lane];
        if (!p && hipHostMalloc((void **)&p, 5 * 24 * 8 * 2, hipHostMallocMapped) != hipSuccess) p = nullptr;
        return p;
    }
    // persistent sumcheck tail (k_fold_tail): host-mapped mailbox + device scratch, created on first use
    TailMail *tail_mail = nullptr;
    u32 *tail_counters = nullptr;      // device, TAIL_MAX_ROUNDS u32 (zeroed once; self-resetting) followed by dev_chal
    u64 *tail_dev_chal = nullptr;
    u32 tail_epoch = 0;
    int num_cus = 0;
    int tail_setup() {
        if (tail_mail) return LF_OK;
        hipDeviceProp_t pr;
        HIPCHK(hipGetDeviceProperties(&pr, device));
        num_cus = pr.multiProcessorCount;
        void *d = nullptr;
        HIPCHK(lf_dev_malloc(&d, 4096));
        HIPCHK(hipMemset(d, 0, 4096));
        tail_counters = (u32 *)d;
        tail_dev_chal = (u64 *)((char *)d + 1024);
        HIPCHK(hipHostMalloc((void **)&tail_mail, sizeof(TailMail), hipHostMallocMapped | hipHostMallocCoherent));   // fine-grained: the kernel and this thread talk through it while the kernel runs
        memset(tail_mail, 0, sizeof(TailMail));
        return LF_OK;
    }
    u64 *d_poseidon = nullptr;   // device copy of the Poseidon constants: ark [720] then mds [576]
    int poseidon_setup() {
        if (d_poseidon) return LF_OK;
        const u64 *a, *m;
        Transcript::params(&a, &m);
        HIPCHK(lf_dev_malloc(&d_poseidon, (720 + 576) * 8));
        HIPCHK(hipMemcpy(d_poseidon, a, 720 * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_poseidon + 720, m, 576 * 8, hipMemcpyHostToDevice));
        return LF_OK;
    }
    hipEvent_t ev_theta = nullptr;
    hipEvent_t ev_block = nullptr;   // hipEventBlockingSync: lane 1 (long waits) yields its CPU instead of spinning
    int lane_sync() {
        if (t_lane == 1 && ev_block) {
            HIPCHK(hipEventRecord(ev_block, st_lane[1]));
            HIPCHK(hipEventSynchronize(ev_block));
            return LF_OK;
        }
        HIPCHK(hipStreamSynchronize(stream()));
        return LF_OK;
    }
    u64 *&h_pin_ref() { return h_pin_lane[t_lane]; }
    int pin(size_t words) {
        u64 *&hp = h_pin_lane[t_lane];
        size_t &hw = h_pin_words_lane[t_lane];
        if (words <= hw) return LF_OK;
        if (hp) (void)hipHostFree(hp);
        hp = nullptr;
        if (words < 8192) words = 8192;
        if (hipHostMalloc((void **)&hp, words * 8) != hipSuccess) return LF_ERR_HIP;
        hw = words;
        return LF_OK;
    }
    // timed-launch helpers: tag 0 = fold round kernels, 1 = ajtai, 10+i = phase i
    size_t ev_begin(int tag) {
        std::lock_guard<std::mutex> g(ev_mu);
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
    void ev_end(size_t i) {
        if (i == (size_t)-1) return;
        std::lock_guard<std::mutex> g(ev_mu);
        (void)hipEventRecord(ev_pool[i].b, stream());
    }
    void ev_reset() {
        ev_used = 0;
        ev_tags.clear();
    }
    void ev_collect() {
        (void)hipStreamSynchronize(st_lane[0]);
        (void)hipStreamSynchronize(st_lane[1]);
        k_fold_ms = k_ajtai_ms = 0;
        k_fold_n = k_ajtai_n = 0;
        for (int i = 0; i < LF_N_PHASES; i++) phase_ms[i] = 0;
        for (auto &tg : ev_tags) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev_pool[tg.second].a, ev_pool[tg.second].b);
            if (tg.first == 0) { k_fold_ms += ms; k_fold_n++; }
            else if (tg.first == 1) { k_ajtai_ms += ms; k_ajtai_n++; }
            else if (tg.first >= 10 && tg.first < 10 + LF_N_PHASES) phase_ms[tg.first - 10] += ms;
        }
        phase_ms[6] = (float)host_tr_ms;
    }
};


// ---- shared between lf_capi.cpp / lf_prove.cpp / lf_fold.cpp (hidden: not part of the ABI) -------------------------------------------------------
#pragma GCC visibility push(hidden)
Fq3Const f3c(Fq3 a);
bool shard_keep(const lf_ctx *c, int kind, size_t n);
int build_eq_dev(lf_ctx *c, const Fq3 *pt, u32 nv, u64 *eq_dev);
int exchange_modsum_dev(lf_ctx *c, u64 *inout_dev, size_t words);
int down_small(lf_ctx *c, const u64 *dsrc, size_t words, u64 *host);
void shard_slice(const lf_ctx *c, size_t n, size_t *i0, size_t *cnt);
struct GatherPart { const u64 *src; size_t src_ld; u64 *dst; size_t planes; };
struct HostTimer {
    lf_ctx *c;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(lf_ctx *cc) : c(cc), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { c->host_tr_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
int shard_col_range(lf_ctx *c, size_t r0, size_t rcnt, size_t *lo, size_t *hi);
Fq3 sc_round_transcript(Transcript &tr, const u64 *evals, u32 npts);
void sc_prologue(Transcript &tr, u32 nv, u32 deg);
int gather_slices(lf_ctx *c, u64 *buf, size_t planes, size_t n);
struct SideState {
    const int32_t *planes = nullptr;
    u64 *z = nullptr;       // [K][24][n]
    u64 *eq_r = nullptr;    // [3][m]
    std::vector<u64> lcccs;  // K flat LCCCS (host)
    // z_k (and x_s in the proof) may be built ahead of the evaluation point by the other lane (decompose_prepare_z): 1 = published
    // (z, x_s valid once z_ev has completed), -1 = that lane failed, 0 = nobody built it yet
    std::atomic<int> z_state{0};
    hipEvent_t z_ev = nullptr;
    u32 *sv_bits = nullptr;  // bit-plane form of the witness planes for the GEMM rounds of the folding sumcheck (lf_sv_rounds.h), if built ahead
    ~SideState() { if (z_ev) (void)hipEventDestroy(z_ev); }
};
int gather_parts(lf_ctx *c, const GatherPart *parts, int np, size_t lcl);
inline thread_local bool t_xb_active = false;   // external-basis conversion in progress on this thread
struct XB {
    lf_ctx *c;
    size_t RE, TAU;
    std::vector<std::unique_ptr<std::vector<u64>>> keep;
    lf_transcript *tr = nullptr;
    explicit XB(lf_ctx *cc) : c(cc), RE((size_t)lf_ring_words(lf_ctx_ring(cc))), TAU((size_t)lf_ring_tau(lf_ctx_ring(cc))) { t_xb_active = true; }
    ~XB() {
        t_xb_active = false;
        if (tr) { tr->t.set_basis(nullptr, nullptr); if (tr->bb) tr->bb->set_basis(nullptr, nullptr); }
    }
    const u64 *ring_in(const u64 *p, size_t elems) {   // NTT-form ring elements, external -> internal (copy)
        if (!p) return p;
        keep.emplace_back(new std::vector<u64>(p, p + elems * RE));
        c->xb.to_int(keep.back()->data(), elems * 8);
        return keep.back()->data();
    }
    const u64 *ext_in(const u64 *p, size_t n) {        // F_{p^tau} elements (tau words each)
        if (!p) return p;
        keep.emplace_back(new std::vector<u64>(p, p + n * TAU));
        c->xb.to_int(keep.back()->data(), n);
        return keep.back()->data();
    }
    void ring_out(u64 *p, size_t elems) { if (p) c->xb.to_ext(p, elems * 8); }
    void ext_out(u64 *p, size_t n) { if (p) c->xb.to_ext(p, n); }
    void transcript(lf_transcript *t) {
        tr = t;
        if (t->bb) t->bb->set_basis(c->xb.T, c->xb.Ti);
        else t->t.set_basis(c->xb.T, c->xb.Ti);
    }
};
#define LF_XB(c) ((c) && (c)->xb.on && !t_xb_active)
int fold_impl(lf_ctx *c, Transcript &tr, SideState *S /* [2] */, u64 *lcccs_out, lf_witness **w_out, u64 *proof);
size_t dec_proof_len(const lf_params *p);
size_t lin_proof_len(const lf_params *p);
int commit_planes_i8(lf_ctx *c, const int32_t *planes, size_t ld, u32 k0, u32 NP, u64 *out_dev, const lf_witness *wit = nullptr);
int commit_download(lf_ctx *c, const u64 *dev, size_t words, u64 *host);
int up_ring(lf_ctx *c, const u64 *host, size_t n, u64 *dst);
int dot_batch_dev(lf_ctx *c, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n, u64 *dpart, u64 *od, hipStream_t st = nullptr,
                         const char *tag = "", unsigned char *yb_pre = nullptr);
int coef_eval_dev(lf_ctx *c, const int32_t *planes, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits, u64 *partial, u64 *od, size_t ldp,
                         const lf_witness *wit = nullptr);
int lin_tail_rounds(lf_ctx *c, Transcript &tr, const u64 *cur, const u64 *cure, size_t n, u64 *tout, u64 *partial, u32 round, Fq3 *point,
                           u64 *msgs, u32 deg, const std::function<void(u32)> *after_round);
int down_ring(lf_ctx *c, const u64 *src, size_t n, u64 *host);
#pragma GCC visibility pop

// lf_capi.cpp -- context, device-resident witnesses, the host driver that replays
// `NIFSProver::prove` (crates/latticefold/src/nifs.rs:48-103) on the GPU kernels, and the C ABI (include/lfhip.h).
//
// Host <-> device traffic inside a fold step is O(proof size): per sumcheck round (D+1) ring elements come back
// and one F_{p^3} challenge goes down; everything of size N stays in HBM.
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

static thread_local int t_lane = 0;  // 0 = caller thread, 1 = helper thread running the left decomposition
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
static thread_local Timeline *t_tl = nullptr;
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
    hipStream_t stream() const { return st_lane[t_lane]; }
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
    hipEvent_t ev_evals[2] = {nullptr, nullptr};  // right evaluations in two stages (decompose_evals, EvalStages): first / second half of the u_s downloaded
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
        const bool prio = !getenv("LF_NO_PRIO");
        const int p0 = least;
        if (hipStreamCreateWithPriority(&c->st_lane[0], hipStreamDefault, prio ? p0 : 0) != hipSuccess ||
            hipStreamCreateWithPriority(&c->st_lane[1], hipStreamDefault, prio ? greatest : 0) != hipSuccess) { delete c; return LF_ERR_HIP; }
    }
    if (!getenv("LF_SPIN_ALL")) (void)hipEventCreateWithFlags(&c->ev_block, hipEventBlockingSync | hipEventDisableTiming);
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
    (void)hipSetDevice(c->device);
    planes_pool_drop(c->device);
    if (c->bb) { c->bb->destroy(); delete c; return; }
    (void)hipStreamSynchronize(c->st_lane[0]);
    (void)hipStreamSynchronize(c->st_lane[1]);
    free_ccs(c);
    if (getenv("LF_MEM_REPORT")) {   // what the context held, largest first
        std::vector<std::pair<size_t, std::string>> v;
        size_t tot = 0;
        for (auto &kv : c->bufs) { v.push_back({kv.second.bytes, kv.first}); tot += kv.second.bytes; }
        const AjtaiI8Ring R = ajtai_i8_goldilocks();
        const size_t ab = c->dAb ? (c->nA + 7) / 8 * (R.RD / 8) * ajtai_i8_row_tiles(R, c->i8_kc) * 1024 * c->i8_nch : 0;
        v.push_back({ab, "(Ajtai byte planes)"});
        std::sort(v.begin(), v.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
        fprintf(stderr, "[lf mem] named buffers %.2f GiB + Ajtai %.2f GiB\n", tot / 1073741824.0, ab / 1073741824.0);
        for (size_t i = 0; i < v.size() && v[i].first >= ((size_t)16 << 20); i++) fprintf(stderr, "[lf mem]   %8.1f MiB  %s\n", v[i].first / 1048576.0, v[i].second.c_str());
    }
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
static thread_local bool t_xb_active = false;
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
// all-gather `words` canonical words from every rank and add them mod p (RCCL has no modular reduction): host buffer ...
static int exchange_modsum(lf_ctx *c, u64 *inout, size_t words) {
    if (c->sh_world <= 1) return LF_OK;
    std::vector<u64> all((size_t)c->sh_world * words);
    RET(c->cm().allgather_host(inout, all.data(), words, c->stream()));
    return lf_modsum(all.data(), (size_t)c->sh_world, words, inout);
}
// ... and device buffer, ordered on the lane's stream (RCCL: no host synchronisation; the reduction is k_modsum)
static int exchange_modsum_dev(lf_ctx *c, u64 *inout_dev, size_t words) {
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
    long limit_ms = 20000;
    if (const char *e = getenv("LF_DIST_HANDSHAKE_MS")) limit_ms = atol(e);
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
static int up_ring(lf_ctx *c, const u64 *host, size_t n, u64 *dst) {
    if (!n) return LF_OK;
    u64 *tmp;
    RET(c->tbuf("stage_aos", n * 24, &tmp));
    HIPCHK(hipMemcpyAsync(tmp, host, n * 24 * 8, hipMemcpyHostToDevice, c->stream()));
    launch_aos_to_soa(tmp, dst, n, c->stream());
    return LF_OK;
}
static int down_ring(lf_ctx *c, const u64 *src, size_t n, u64 *host) {
    if (!n) return LF_OK;
    u64 *tmp;
    RET(c->tbuf("stage_aos", n * 24, &tmp));
    launch_soa_to_aos(src, tmp, n, c->stream());
    HIPCHK(hipMemcpyAsync(host, tmp, n * 24 * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    return LF_OK;
}
// small device array -> host (through pinned memory)
static int down_small(lf_ctx *c, const u64 *dsrc, size_t words, u64 *host) {
    RET(c->pin(words));
    HIPCHK(hipMemcpyAsync(c->h_pin_ref(), dsrc, words * 8, hipMemcpyDeviceToHost, c->stream()));
    RET(c->lane_sync());
    memcpy(host, c->h_pin_ref(), words * 8);
    return LF_OK;
}
static Fq3Const f3c(Fq3 a) { Fq3Const r; r.c[0] = a.c[0]; r.c[1] = a.c[1]; r.c[2] = a.c[2]; return r; }

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
static int commit_planes_i8(lf_ctx *c, const int32_t *planes, size_t ld, u32 k0, u32 NP, u64 *out_dev, const lf_witness *wit = nullptr) {
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
static int commit_download(lf_ctx *c, const u64 *dev, size_t words, u64 *host) {
    RET(exchange_modsum_dev(c, (u64 *)dev, words));   // sharded: ncclAllGather of the partial commitments + k_modsum, in stream
    return down_small(c, dev, words, host);
}
// index slice of this rank: [*i0, *i0 + *cnt) of n items (the last rank takes the remainder)
static void shard_slice(const lf_ctx *c, size_t n, size_t *i0, size_t *cnt) {
    size_t per = (n + (size_t)c->sh_world - 1) / (size_t)c->sh_world;
    size_t lo = per * (size_t)c->sh_rank;
    if (lo > n) lo = n;
    *i0 = lo;
    *cnt = lo + per > n ? n - lo : per;
}
// Sharded sumchecks: tables of `n` entries stay sharded while every rank keeps at least 64 pairs AND the tables are larger than the hand-over size of the
// sumcheck (kind 0 linearization, 1 folding; Tunables::shard_lin_min / shard_fold_min, never above m / 16 so that small instances still exercise the sharded
// rounds).  Every rank evaluates the same predicate on the same numbers: the ranks leave the sharded form in the same round.
static bool shard_keep(const lf_ctx *c, int kind, size_t n) {
    const size_t Gw = (size_t)c->sh_world;
    if (Gw <= 1 || n / 2 < Gw * 64) return false;
    size_t thr = kind ? c->tn.shard_fold_min : c->tn.shard_lin_min;
    if (thr > (c->m >> 4)) thr = c->m >> 4;
    return n > thr;
}
// all-gather the ranks' column slices of `planes` tables stored with GLOBAL layout [plane][n] (rank g holds entries
// [g*n/G, (g+1)*n/G) of every plane) and fill in the others' slices
static int gather_slices(lf_ctx *c, u64 *buf, size_t planes, size_t n) {
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
struct GatherPart { const u64 *src; size_t src_ld; u64 *dst; size_t planes; };
static int gather_parts(lf_ctx *c, const GatherPart *parts, int np, size_t lcl) {
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
static int build_eq_dev(lf_ctx *c, const Fq3 *pt, u32 nv, u64 *eq_dev) {
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
static size_t lin_proof_len(const lf_params *p) { return (size_t)p->s * (p->d + 2) + 3 + p->t; }
static size_t dec_proof_len(const lf_params *p) { return (size_t)p->K * (p->t + 3 + p->l + 1 + p->kappa); }
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
int lf_witness_from_w_ccs(lf_ctx *c, const uint64_t *w_ccs, lf_witness **out) {
    if (LF_XB(c) && w_ccs && c->have_ccs_any()) { XB x(c); return lf_witness_from_w_ccs(c, x.ring_in(w_ccs, c->params_any().wit_len), out); }
    if (!c || !w_ccs || !out) return LF_ERR_INVALID;
    if (c->bb) return c->bb->witness_from_w_ccs(w_ccs, out);
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    // Witness::from_w_ccs, arith.rs:230-248: ICRT -> gadget_decompose(B, L)
    u64 *a, *b, *d;
    RET(c->tbuf("io_a", (size_t)c->P.wit_len * 24, &a));
    RET(c->tbuf("io_b", (size_t)c->P.wit_len * 24, &b));
    RET(c->tbuf("io_c", c->N * 24, &d));
    RET(up_ring(c, w_ccs, c->P.wit_len, a));
    launch_icrt_dense(c->d_icrt, a, b, c->P.wit_len, c->stream());
    launch_decompose(b, c->P.wit_len, c->P.B, c->P.L, 0, d, c->stream(), c->digit_mode);
    return witness_from_coef_table(c, d, out);
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

// =================================================================================================================================
// the driver
struct HostTimer {
    lf_ctx *c;
    std::chrono::steady_clock::time_point t0;
    explicit HostTimer(lf_ctx *cc) : c(cc), t0(std::chrono::steady_clock::now()) {}
    ~HostTimer() { c->host_tr_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// sumcheck transcript prologue: absorb R::from(nvars), R::from(degree)  (utils/sumcheck.rs:60-62)
static void sc_prologue(Transcript &tr, u32 nv, u32 deg) {
    tr.absorb_u64_as_ring(nv);
    tr.absorb_u64_as_ring(deg);
}
static Fq3 sc_round_transcript(Transcript &tr, const u64 *evals, u32 npts) {
    tr.absorb_ring(evals, npts);
    Fq3 r = tr.get_challenge();
    tr.absorb_fq3_as_ring(r);
    return r;
}

static int lin_tail_rounds(lf_ctx *c, Transcript &tr, const u64 *cur, const u64 *cure, size_t n, u64 *tout, u64 *partial, u32 round, Fq3 *point,
                           u64 *msgs, u32 deg, const std::function<void(u32)> *after_round);
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
    bool split = beta && Gw == 1 && !c->tn.lin_no_split && !c->tn.lin_unfused && P.s >= 2 && m >= c->tn.lin_split_min && m >= 16 && dT >= 1 && deg <= 4;
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
            } else if (split || (Gw == 1 && !c->tn.lin_unfused && n >= 8)) {
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
static int shard_col_range(lf_ctx *c, size_t r0, size_t rcnt, size_t *lo, size_t *hi) {
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

static int coef_eval_dev(lf_ctx *c, const int32_t *planes, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits, u64 *partial, u64 *od, size_t ldp,
                         const lf_witness *wit = nullptr);
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
    if (P.b == 2 && c->sh_world == 1 && !c->tn.force_exchange && !c->tn.lin_v_direct && !c->tn.coef_valu && !c->tn.coef_planes && !c->tn.lin_vs_whole && P.s >= 12 &&
        P.K <= 16) {
        const u32 back = c->tn.lin_vs_back;                  // rounds before the last one after which the pass starts
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
        if (P.b == 2 && c->sh_world == 1 && !c->tn.force_exchange && !c->tn.lin_v_direct) {
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
static int coef_eval_dev(lf_ctx *c, const int32_t *planes, size_t n, const u64 *eq, size_t ldeq, u32 K, int mode_bits, u64 *partial, u64 *od, size_t ldp,
                         const lf_witness *wit) {
    // the whole witness of a running fold step, digit planes: from its bit-plane form with the round-1 GEMM of lf_sv_rounds.hip (the digit
    // cutting of k_coef_eval_i8 from the int32 planes is what that kernel spends its time on)
    if (wit && mode_bits && !c->tn.coef_valu && !c->tn.coef_planes && planes == wit->planes && n == c->N)
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
    if (!c->tn.coef_valu && n >= 64) {
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
static int dot_batch_dev(lf_ctx *c, const u64 *X, size_t ldx, u32 na, const u64 *Y, size_t ldy, u32 nb, size_t n, u64 *dpart, u64 *od, hipStream_t st = nullptr,
                         const char *tag = "", unsigned char *yb_pre = nullptr) {
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
// The evaluations of a decomposition in two stages (the right side of a fold step): stage 0 = all v_s and the u_s of the parts k < ksplit, stage 1 = the
// other u_s.  decompose_evals then enqueues both downloads and returns without waiting; decompose_evals_collect(stage) waits for that stage's event and
// moves its words into the proof -- so the host can absorb the first K/2 parts (x_k, y_k, u_k, v_k: half of a ~1 ms sponge chain) while the GPU is
// still computing the inner products of the second half.  The absorb order (decomposition.rs:65-83: part by part) is unchanged.
struct EvalStages {
    u32 ksplit = 0;
    bool active = false;
};
static int decompose_evals_collect(lf_ctx *c, const EvalStages &st, int stage, u64 *proof) {
    const lf_params &P = c->P;
    const u32 K = P.K;
    u64 *u_s = proof, *v_s = u_s + (size_t)K * P.t * 24;
    HIPCHK(hipEventSynchronize(c->ev_evals[stage]));
    const u64 *h = c->h_pin_ref();
    if (stage == 0) {
        memcpy(v_s, h, (size_t)K * 72 * 8);
        memcpy(u_s, h + 32 * 72, (size_t)st.ksplit * P.t * 24 * 8);
    } else {
        memcpy(u_s + (size_t)st.ksplit * P.t * 24, h + 32 * 72 + (size_t)st.ksplit * P.t * 24, (size_t)(K - st.ksplit) * P.t * 24 * 8);
    }
    return LF_OK;
}
static int decompose_evals(lf_ctx *c, const u64 *lcccs, const std::vector<Fq3> &rpt, const lf_witness *wit, const char *side,
                           u64 *eq_r /* built already or nullptr */, SideState &S, u64 *proof, EvalStages *stages = nullptr) {
    const lf_params &P = c->P;
    size_t m = c->m, n = c->n, N = c->N;
    u32 K = P.K;
    std::string sd(side);
    const u64 *xh = lcccs + ((size_t)P.s + 3 + P.kappa + P.t) * 24;
    u64 *u_s = proof, *v_s = u_s + (size_t)K * P.t * 24;
    u64 *partial, *od, *q;
    RET(c->tbuf("red_partial", 256 * 4096, &partial));
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
        if (stages && c->sh_world == 1 && K >= 4 && !c->tn.evals_one_stage) {
            // two stages: parts [0, K/2) and [K/2, K), each with its own download and event
            const u32 ks = K / 2;
            const size_t words = (size_t)32 * 72 + (size_t)K * P.t * 24;
            RET(c->pin(words));
            for (int e = 0; e < 2; e++)
                if (!c->ev_evals[e]) HIPCHK(hipEventCreateWithFlags(&c->ev_evals[e], hipEventDisableTiming));
            RET(dot_batch_dev(c, z + c0, n, ks, q + c0, n, P.t, cnt, dpart, od_u));
            HIPCHK(hipMemcpyAsync(c->h_pin_ref(), od, ((size_t)32 * 72 + (size_t)ks * P.t * 24) * 8, hipMemcpyDeviceToHost, c->stream()));
            HIPCHK(hipEventRecord(c->ev_evals[0], c->stream()));
            RET(dot_batch_dev(c, z + c0 + (size_t)ks * 24 * n, n, K - ks, q + c0, n, P.t, cnt, dpart, od_u + (size_t)ks * P.t * 24));
            HIPCHK(hipMemcpyAsync(c->h_pin_ref() + 32 * 72 + (size_t)ks * P.t * 24, od_u + (size_t)ks * P.t * 24, (size_t)(K - ks) * P.t * 24 * 8, hipMemcpyDeviceToHost, c->stream()));
            HIPCHK(hipEventRecord(c->ev_evals[1], c->stream()));
            stages->ksplit = ks;
            stages->active = true;
            LF_TRACE(c, "decompose evals (staged)");
            c->ev_end(ph);
            return LF_OK;
        }
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
static int lin_tail_rounds(lf_ctx *c, Transcript &tr, const u64 *cur, const u64 *cure, size_t n, u64 *tout, u64 *partial, u32 round, Fq3 *point,
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
static int fold_impl(lf_ctx *c, Transcript &tr, SideState *S /* [2] */, u64 *lcccs_out, lf_witness **w_out, u64 *proof) {
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
        hipStream_t s0 = c->stream(), s1 = (t_lane == 0 && !c->tn.prep_one_stream) ? c->st_lane[1] : s0;
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
    const bool sv_two_streams = t_lane == 0 && !c->tn.prep_one_stream && Gw == 1;
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
        else if (fmode == 3 && c->dcrt.nu2p40 && !c->tn.fold_no_mutab) {
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
    hipStream_t s1f = (t_lane == 0 && !c->tn.prep_one_stream && c->sh_world == 1 && c->st_lane[1]) ? c->st_lane[1] : c->stream();
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
        hipStream_t s1 = (t_lane == 0 && !c->tn.prep_one_stream && c->sh_world == 1 && c->st_lane[1]) ? c->st_lane[1] : c->stream();
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
    c->lin_blocks = c->tn.lin_blocks >= 0 ? (u32)c->tn.lin_blocks : 0u;
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
    EvalStages est;
    if (rc == LF_OK) {
        lcccs_point(P, lin.data(), rR);
        while (S[1].z_state.load(std::memory_order_acquire) == 0) std::this_thread::yield();   // published by lane 1 within its first millisecond (or -1)
        rc = decompose_evals(c, lin.data(), rR, w_i, "R", eq_r_R, S[1], decr, &est);
    }
    c->vs_wit = nullptr;
    TL_MARK(est.active ? "right evals enqueued" : "right evals done");
    int rc1 = c->lane1.wait();
    c->lin_blocks = 0;
    TL_MARK("lane 1 joined");
    if (rc == LF_OK) rc = rc1;
    if (rc == LF_OK && est.active) {
        // the right decomposition is absorbed part by part (x_k, y_k, u_k, v_k): the first half as soon as its inner products are down, the second half
        // of the inner products is still running on the GPU meanwhile
        rc = decompose_evals_collect(c, est, 0, decr);
        TL_MARK("right evals (first half) down");
        if (rc == LF_OK) c->host_tr_ms += absorb_decomposition(P, tr, lin.data(), decr, S[1], 0, est.ksplit);
        if (rc == LF_OK) rc = decompose_evals_collect(c, est, 1, decr);
        TL_MARK("right evals done");
        if (rc == LF_OK) c->host_tr_ms += absorb_decomposition(P, tr, lin.data(), decr, S[1], est.ksplit, P.K);
    } else if (rc == LF_OK) c->host_tr_ms += absorb_decomposition(P, tr, lin.data(), decr, S[1]);
    if (rc != LF_OK && est.active) (void)hipStreamSynchronize(c->st_lane[0]);   // (nothing may still write the pinned staging when the step returns)
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

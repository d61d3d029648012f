// bb_capi.cpp -- BabyBearRingNTT backend of the C ABI, part 1: context, ring tables, staging, the component entry points (CRT, decomposition, Ajtai
// commitments, eq tables, MLE evaluations, SpMV), constraint-system load, device-resident witnesses, timing read-outs and the host verifier.  The provers are
// in bb_prove.cpp.
#include "bb_ctx.h"

namespace lfbb {

size_t bb_lcccs_len(const lf_params *p) { return (size_t)p->s + TAU + p->kappa + p->t + p->l + 1; }
size_t bb_cccs_len(const lf_params *p) { return (size_t)p->kappa + p->l; }
size_t lin_proof_len(const lf_params *p) { return (size_t)p->s * (p->d + 2) + TAU + p->t; }
size_t dec_proof_len(const lf_params *p) { return (size_t)p->K * (p->t + TAU + p->l + 1 + p->kappa); }
static size_t fold_proof_len(const lf_params *p) { return (size_t)p->s * (2 * p->b + 1) + 2 * (size_t)p->K * (TAU + p->t); }
size_t bb_proof_len(const lf_params *p) { return lin_proof_len(p) + 2 * dec_proof_len(p) + fold_proof_len(p); }

// ---------------------------------------------------------------------------------------------------------------
static int install_tables(C *c, u64 nonres, const u64 *y) {
    static thread_local BbTables T;
    if (bb_build_tables(nonres, y, T) != 0) return LF_ERR_BAD_TABLES;
    c->ring.T = T;
    c->dev = make_dev_bb(T);
    std::vector<fe> mat((size_t)D * D);
    for (int i = 0; i < D; i++)
        for (int j = 0; j < D; j++) mat[(size_t)i * D + j] = from_canon(T.icrt[i][j]);
    if (!c->d_icrt) HIPCHK(lf_dev_malloc(&c->d_icrt, mat.size() * sizeof(fe)));
    HIPCHK(hipMemcpy(c->d_icrt, mat.data(), mat.size() * sizeof(fe), hipMemcpyHostToDevice));
    // compressed rows for the digit pass of the general commitment (k_i8g_cut_ntt): the shipped tables have one entry per slot
    std::vector<fe> sv((size_t)D * 8, 0);
    std::vector<u32> sc((size_t)D * 8, 0xFFFFFFFFu);
    bool sparse = true;
    for (int r = 0; r < D && sparse; r++) {
        int q = 0;
        for (int col = 0; col < D; col++)
            if (T.icrt[r][col]) {
                if (q == 8) { sparse = false; break; }
                sv[(size_t)r * 8 + q] = mat[(size_t)r * D + col]; sc[(size_t)r * 8 + q] = (u32)col; q++;
            }
    }
    if (sparse) {
        if (!c->d_icrt_sp_val) { HIPCHK(lf_dev_malloc(&c->d_icrt_sp_val, sv.size() * sizeof(fe))); HIPCHK(lf_dev_malloc(&c->d_icrt_sp_col, sc.size() * sizeof(u32))); }
        HIPCHK(hipMemcpy(c->d_icrt_sp_val, sv.data(), sv.size() * sizeof(fe), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d_icrt_sp_col, sc.data(), sc.size() * sizeof(u32), hipMemcpyHostToDevice));
    } else if (c->d_icrt_sp_val) {
        (void)hipFree(c->d_icrt_sp_val); (void)hipFree(c->d_icrt_sp_col);
        c->d_icrt_sp_val = nullptr; c->d_icrt_sp_col = nullptr;
    }
    return LF_OK;
}
int BbCtx::create(BbCtx **out, lf_ctx *owner, int device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device < 0 || device >= cnt) return LF_ERR_HIP;
    HIPCHK(hipSetDevice(device));
    C *c = new C();
    c->owner = owner;
    c->device = device;
    {   // lane 1 carries the critical chain of a fold step (two commits back to back); its kernels get dispatch priority over
        // lane 0's latency-bound linearization, which has slack (LF_NO_PRIO=1: equal priorities)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool prio = true;
        if (hipStreamCreateWithPriority(&c->st_lane[0], hipStreamDefault, prio ? least : 0) != hipSuccess ||
            hipStreamCreateWithPriority(&c->st_lane[1], hipStreamDefault, prio ? greatest : 0) != hipSuccess) { delete c; return LF_ERR_HIP; }
    }
    c->arena_words = (size_t)1 << 19;   // 4 MiB per lane
    for (int l = 0; l < 2; l++) {
        if (hipHostMalloc((void **)&c->arena[l], c->arena_words * 8) != hipSuccess) { delete c; return LF_ERR_HIP; }
        if (hipEventCreateWithFlags(&c->ev_side[l], hipEventDisableTiming) != hipSuccess) { delete c; return LF_ERR_HIP; }
    }
    for (int l = 0; l < 4; l++)
        if (hipEventCreateWithFlags(&c->ev_dec[l], hipEventDisableTiming) != hipSuccess) { delete c; return LF_ERR_HIP; }
    u64 nr, y[8 * TAU];
    bb_default_ring(&nr, y);
    int rc = install_tables(c, nr, y);
    if (rc != LF_OK) { delete c; return rc; }
    BbCtx *b = new BbCtx();
    b->p = c;
    *out = b;
    return LF_OK;
}
static void free_ccs(C *c) {
    for (auto q : c->d_rowptr) (void)hipFree(q);
    for (auto q : c->d_col) (void)hipFree(q);
    for (auto q : c->d_val) (void)hipFree(q);
    for (auto q : c->d_colptr) (void)hipFree(q);
    for (auto q : c->d_rowidx) (void)hipFree(q);
    for (auto q : c->d_valT) (void)hipFree(q);
    c->d_rowptr.clear(); c->d_col.clear(); c->d_val.clear(); c->d_colptr.clear(); c->d_rowidx.clear(); c->d_valT.clear();
    c->have_ccs = false;
}
void BbCtx::destroy() {
    C *c = p;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->st_lane[0]);
    (void)hipStreamSynchronize(c->st_lane[1]);
    free_ccs(c);
    c->comm.destroy();
    for (auto &kv : c->bufs) kv.second.release();
    if (c->dA) (void)hipFree(c->dA);
    if (c->dAb) (void)hipFree(c->dAb);
    if (c->d_icrt) (void)hipFree(c->d_icrt);
    if (c->d_icrt_sp_val) { (void)hipFree(c->d_icrt_sp_val); (void)hipFree(c->d_icrt_sp_col); }
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_round) (void)hipHostFree(c->h_round);

    for (auto &e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (int l = 0; l < 2; l++) {
        if (c->arena[l]) (void)hipHostFree(c->arena[l]);
        if (c->ev_side[l]) (void)hipEventDestroy(c->ev_side[l]);
        if (c->ev_prep[l]) (void)hipEventDestroy(c->ev_prep[l]);
        (void)hipStreamDestroy(c->st_lane[l]);
    }
    for (int l = 0; l < 4; l++)
        if (c->ev_dec[l]) (void)hipEventDestroy(c->ev_dec[l]);
    delete c;
    delete this;
}
int BbCtx::set_ring_tables(uint64_t nonres, const uint64_t *y) {
    std::lock_guard<std::mutex> g(p->mu);
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->st_lane[0]));
    HIPCHK(hipStreamSynchronize(p->st_lane[1]));
    return install_tables(p, nonres, y);
}
int BbCtx::set_sharding(int rank, int world, lf_exchange_fn cb, void *user) {
    if (world < 1 || rank < 0 || rank >= world || (world & (world - 1)) != 0 || (world > 1 && !cb)) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(p->mu);
    if (p->dAb) return LF_ERR_STATE;   // choose the sharding before loading/generating the Ajtai matrix
    p->comm.destroy();
    p->comm.rank = p->sh_rank = rank; p->comm.world = p->sh_world = world; p->comm.cb = cb; p->comm.user = user;
    return LF_OK;
}
int BbCtx::dist_init(int rank, int world, const uint8_t *id128) {
    std::lock_guard<std::mutex> g(p->mu);
    if (p->dAb) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(p->device));
    p->comm.destroy();
    RET(lfdist::rccl_init(p->comm, rank, world, id128));
    p->sh_rank = rank; p->sh_world = world;
    return LF_OK;
}
lfdist::Comm *BbCtx::comm() { return &p->comm; }
void BbCtx::set_digit_mode(int mode) { p->digit_mode = mode; }
bool BbCtx::have_ccs() const { return p->have_ccs; }
const lf_params &BbCtx::params() const { return p->P; }
size_t BbCtx::dim_n() const { return p->n; }
size_t BbCtx::dim_m() const { return p->m; }
size_t BbCtx::dim_N() const { return p->N; }
uint32_t BbCtx::kappa() const { return p->kappa; }
int BbCtx::get_ring_tables(uint64_t *nonres, uint64_t *y) {
    *nonres = p->ring.T.nu;
    for (int k = 0; k < 8; k++)
        for (int q = 0; q < TAU; q++) y[TAU * k + q] = p->ring.T.y[k].c[q];
    return LF_OK;
}
int BbCtx::synchronize() {
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->st_lane[0]));
    HIPCHK(hipStreamSynchronize(p->st_lane[1]));
    return LF_OK;
}
int BbCtx::mem_info(size_t *f, size_t *t) {
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipMemGetInfo(f, t));
    return LF_OK;
}

// ---- host<->device staging of AoS ring-element arrays (canonical u64 at the ABI, Montgomery planes on the device) -----
int up_ring(C *c, const u64 *host, size_t n, fe *dst) {
    if (!n) return LF_OK;
    u64 *tmp;
    RET(c->tbuf("stage_aos", n * RE, &tmp));
    HIPCHK(hipMemcpyAsync(tmp, host, n * RE * 8, hipMemcpyHostToDevice, c->stream()));
    launch_aos_to_soa(tmp, dst, n, c->stream());
    return LF_OK;
}
int down_ring(C *c, const fe *src, size_t n, u64 *host) {
    if (!n) return LF_OK;
    u64 *tmp;
    RET(c->tbuf("stage_aos", n * RE, &tmp));
    launch_soa_to_aos(src, tmp, n, c->stream());
    HIPCHK(hipMemcpyAsync(host, tmp, n * RE * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    return LF_OK;
}
int down_small(C *c, const u64 *dsrc, size_t words, u64 *host) {
    RET(c->pin(words));
    HIPCHK(hipMemcpyAsync(c->h_pin, dsrc, words * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    memcpy(host, c->h_pin, words * 8);
    return LF_OK;
}
// all-gather `words` canonical words from every rank and add them mod p (RCCL has no modular reduction)
int exchange_modsum(C *c, u64 *inout, size_t words) {
    if (c->sh_world <= 1) return LF_OK;
    std::vector<u64> all((size_t)c->sh_world * words);
    RET(c->comm.allgather_host(inout, all.data(), words, c->stream()));
    for (size_t w = 0; w < words; w++) {
        u64 acc = 0;
        for (int g = 0; g < c->sh_world; g++) {
            u64 v = all[(size_t)g * words + w];
            if (v >= BB_P) return LF_ERR_INVALID;
            acc = hadd(acc, v);
        }
        inout[w] = acc;
    }
    return LF_OK;
}
static int shard_columns(C *c, size_t n, size_t *col0, size_t *cnt) {
    if (n % (size_t)c->sh_world) return LF_ERR_UNSUPPORTED;
    *cnt = n / c->sh_world;
    *col0 = *cnt * c->sh_rank;
    return LF_OK;
}
// wall-clock marks of a fold step on stderr (LF_TIMELINE=1; measurement only)
bool h9_inv(const H9 &a, u64 nu, H9 *out) {
    u64 M[TAU][TAU + 1];
    for (int i = 0; i < TAU; i++) {
        for (int j = 0; j < TAU; j++) M[i][j] = i >= j ? a.c[i - j] % BB_P : hmul(nu % BB_P, a.c[TAU + i - j] % BB_P);
        M[i][TAU] = i == 0;
    }
    for (int col = 0; col < TAU; col++) {
        int piv = -1;
        for (int r = col; r < TAU; r++)
            if (M[r][col]) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != col)
            for (int j = 0; j <= TAU; j++) std::swap(M[piv][j], M[col][j]);
        const u64 iv = hinv(M[col][col]);
        for (int j = col; j <= TAU; j++) M[col][j] = hmul(M[col][j], iv);
        for (int r = 0; r < TAU; r++) {
            if (r == col || !M[r][col]) continue;
            const u64 f = M[r][col];
            for (int j = col; j <= TAU; j++) M[r][j] = hsub(M[r][j], hmul(f, M[col][j]));
        }
    }
    for (int i = 0; i < TAU; i++) out->c[i] = M[i][TAU];
    return true;
}
bool is_diag(const u64 *e, H9 *out) {
    for (int k = 1; k < 8; k++)
        if (memcmp(e + TAU * k, e, TAU * 8)) return false;
    if (out) *out = h9_load(e);
    return true;
}

int BbCtx::selftest_field(uint64_t seed, uint32_t n, uint64_t *mismatches) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    std::vector<u64> in((size_t)n * 18), out((size_t)n * 12);
    u64 s = seed * 0x9E3779B97F4A7C15ULL + 1;
    for (size_t i = 0; i < in.size(); i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        in[i] = s % BB_P;
    }
    // edge operands in the first elements
    for (int q = 0; q < 18 && n > 2; q++) { in[q] = (BB_P - 1) / 2; in[18 + q] = q < 9 ? (BB_P - 1) / 2 : (BB_P + 1) / 2; in[36 + q] = BB_P - 1; }
    u64 *di, *dout;
    RET(c->tbuf("io_a", in.size() * 2, (fe **)&di));
    RET(c->tbuf("io_b", out.size() * 2, (fe **)&dout));
    HIPCHK(hipMemcpyAsync(di, in.data(), in.size() * 8, hipMemcpyHostToDevice, c->stream()));
    launch_selftest(di, dout, n, c->dev.nu, c->stream());
    HIPCHK(hipMemcpyAsync(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    u64 bad = 0;
    for (u32 i = 0; i < n; i++) {
        H9 a = h9_load(&in[(size_t)i * 18]), b = h9_load(&in[(size_t)i * 18 + 9]);
        H9 pr = c->ring.mul9(a, b);
        for (int q = 0; q < TAU; q++) bad += out[(size_t)i * 12 + q] != pr.c[q];
        bad += out[(size_t)i * 12 + 9] != hadd(a.c[0], b.c[0]);
        bad += out[(size_t)i * 12 + 10] != hsub(a.c[0], b.c[0]);
        bad += out[(size_t)i * 12 + 11] != hmul(a.c[0], b.c[0]);
    }
    *mismatches = bad;
    return LF_OK;
}

// ---- a1/a2/a3 --------------------------------------------------------------------------------------------------------
int BbCtx::ntt_fwd(const uint64_t *in, uint64_t *out, size_t count) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *a, *b;
    RET(c->tbuf("io_a", count * RE, &a));
    RET(c->tbuf("io_b", count * RE, &b));
    RET(up_ring(c, in, count, a));
    launch_crt_fwd(c->dev, a, b, count, c->stream());
    return down_ring(c, b, count, out);
}
int BbCtx::ntt_inv(const uint64_t *in, uint64_t *out, size_t count) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *a, *b;
    RET(c->tbuf("io_a", count * RE, &a));
    RET(c->tbuf("io_b", count * RE, &b));
    RET(up_ring(c, in, count, a));
    launch_icrt_dense(c->d_icrt, a, b, count, c->stream());
    return down_ring(c, b, count, out);
}
static bool pow2(u64 b) { return b >= 2 && (b & (b - 1)) == 0; }
int BbCtx::decompose(const uint64_t *in, size_t count, uint64_t base, unsigned digits, int layout, uint64_t *out) {
    C *c = p;
    if (!pow2(base)) return LF_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *a, *b;
    RET(c->tbuf("io_a", count * RE, &a));
    RET(c->tbuf("io_b", count * digits * RE, &b));
    RET(up_ring(c, in, count, a));
    launch_decompose(a, count, base, digits, layout, b, c->stream(), c->digit_mode);
    if (layout == 0) return down_ring(c, b, count * digits, out);
    for (unsigned k = 0; k < digits; k++) RET(down_ring(c, b + (size_t)k * RE * count, count, out + (size_t)k * count * RE));
    return LF_OK;
}
int BbCtx::recompose(const uint64_t *in, size_t count_out, uint64_t base, unsigned digits, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *a, *b;
    RET(c->tbuf("io_a", count_out * digits * RE, &a));
    RET(c->tbuf("io_b", count_out * RE, &b));
    RET(up_ring(c, in, count_out * digits, a));
    launch_recompose(a, count_out, base, digits, b, c->stream());
    return down_ring(c, b, count_out, out);
}
int BbCtx::linf_check(const uint64_t *f_ntt, size_t count, uint64_t bound, int unsigned_variant, int *ok, uint64_t *max_out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *a, *b;
    u64 *mx;
    RET(c->tbuf("io_a", count * RE, &a));
    RET(c->tbuf("io_b", count * RE, &b));
    RET(c->tbuf("small_dev", 4096, &mx));
    RET(up_ring(c, f_ntt, count, a));
    launch_icrt_dense(c->d_icrt, a, b, count, c->stream());
    if (unsigned_variant) {   // literal Witness::within_bound (arith.rs:372-386): canonical coefficient < bound
        std::vector<u64> h(count * RE);
        RET(down_ring(c, b, count, h.data()));
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

// ---- a5 ----------------------------------------------------------------------------------------------------------------
// The int8 matrix-core commit kernel (lf_ajtai_i8.hip, shared with the Goldilocks backend) wants A in coefficient form, cut into its 4
// bytes, in MFMA operand order: built once per matrix.
static int prep_ajtai_i8(C *c) {
    if (c->dAb) { (void)hipFree(c->dAb); c->dAb = nullptr; }
    c->i8_nch = 0;
    const lf::AjtaiI8Ring R = lf::ajtai_i8_babybear();
    const u32 maxr = lf::ajtai_i8_max_rows(R), nch = (c->kappa + maxr - 1) / maxr, kc = (c->kappa + nch - 1) / nch;
    const size_t ntiles = (c->nA + 7) / 8;
    const u32 MT = lf::ajtai_i8_row_tiles(R, kc);
    const size_t chunk_bytes = ntiles * (R.RD / 8) * MT * 1024;
    HIPCHK(lf_dev_malloc(&c->dAb, chunk_bytes * nch + lf::ajtai_i8_slack_bytes()));
    HIPCHK(hipMemsetAsync(c->dAb, 0, chunk_bytes * nch + lf::ajtai_i8_slack_bytes(), c->stream()));
    fe *coef;
    u64 *canon;
    RET(c->tbuf("i8_prep_coef", (size_t)RE * c->nA, &coef));
    RET(c->tbuf("i8_prep_canon", (size_t)RE * c->nA, &canon));
    for (u32 i = 0; i < c->kappa; i++) {
        launch_icrt_dense(c->d_icrt, c->dA + (size_t)i * RE * c->nA, coef, c->nA, c->stream());
        launch_soa_to_aos(coef, canon, c->nA, c->stream());   // canonical u64, element-major
        lf::launch_ajtai_pack_i8(canon, 1, RE, c->nA, i % kc, MT, R.RD, R.NL, c->dAb + (size_t)(i / kc) * chunk_bytes, c->stream());
    }
    HIPCHK(hipStreamSynchronize(c->stream()));
    c->i8_nch = nch;
    c->i8_kc = kc;
    // the byte planes are the only resident form of A: digit-plane and general commitments (lf_ajtai_i8.hip / lf_ajtai_i8g.hip) both stream them
    (void)hipFree(c->dA);
    c->dA = nullptr;
    return LF_OK;
}
// digit planes k0 .. k0+NP-1 of `planes` (this rank's column slice) -> out_dev canonical u64 [NP][kappa][72], NTT form (PARTIAL when sharded)
int commit_planes_i8(C *c, const int32_t *planes, size_t ld, u32 k0, u32 NP, u64 *out_dev) {
    const lf::AjtaiI8Ring R = lf::ajtai_i8_babybear();
    const u32 nch = c->i8_nch, kc = c->i8_kc, MT = lf::ajtai_i8_row_tiles(R, kc), maxp = lf::ajtai_i8_max_planes_mt(R, MT);
    const size_t ntiles = (c->nA + 7) / 8, chunk_bytes = ntiles * (R.RD / 8) * MT * 1024;
    u32 nwg = c->tn.i8_wgs > 0 ? (u32)c->tn.i8_wgs : 224;   // 7/8 of the CUs: see the Goldilocks backend
    if (nwg > ntiles) nwg = (u32)ntiles;
    const u32 nslots = nwg < 16 ? 16 : nwg;    // (two plane groups run as 2 x 8 chunks at least: launch_ajtai_i8)
    int32_t *part, *dsum;
    long long *sum;
    u64 *coef;
    fe *cf, *ntt;
    const u32 NTmax = lf::ajtai_i8_col_tiles(R, maxp);
    RET(c->tbuf("i8_part", lf::ajtai_i8_part_words(nslots, MT, NTmax), &part));
    RET(c->tbuf("i8_dsum", (size_t)nslots * maxp * R.RD, &dsum));
    RET(c->tbuf("i8_sum", lf::ajtai_i8_sum_words(R, MT, NTmax, maxp), &sum));
    RET(c->tbuf("i8_coef", (size_t)RE * NP * c->kappa, &coef));
    RET(c->tbuf("i8_cf", (size_t)RE * NP * c->kappa, &cf));
    RET(c->tbuf("i8_ntt", (size_t)RE * NP * c->kappa, &ntt));
    for (u32 p0 = 0; p0 < NP; p0 += maxp) {
        const u32 np = NP - p0 < maxp ? NP - p0 : maxp;
        u64 *co = coef + (size_t)RE * p0 * c->kappa;   // element-major block of this plane group: [np*kappa][72] canonical
        for (u32 ch = 0; ch < nch; ch++) {
            const u32 row0 = ch * kc, kn = c->kappa - row0 < kc ? c->kappa - row0 : kc;
            size_t ev = c->ev_begin(1);
            int g = lf::launch_ajtai_i8(R, c->dAb + (size_t)ch * chunk_bytes, MT, planes, ld, c->nA, kn, row0, c->kappa, k0 + p0, np, nwg, part, dsum, sum, co,
                                        c->stream());
            c->ev_end(ev);
            if (g < 0) return LF_ERR_UNSUPPORTED;
        }
        const size_t ne = (size_t)np * c->kappa;
        launch_aos_to_soa(co, cf, ne, c->stream());            // canonical -> Montgomery planes
        launch_crt_fwd(c->dev, cf, ntt, ne, c->stream());
        launch_soa_to_aos(ntt, out_dev + (size_t)p0 * c->kappa * RE, ne, c->stream());
    }
    return LF_OK;
}

int BbCtx::ajtai_load(const uint64_t *A, size_t kappa, size_t n) {
    C *c = p;
    if (kappa > 32) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    size_t col0, cnt;
    RET(shard_columns(c, n, &col0, &cnt));   // a sharded rank keeps only its column slice of the caller's matrix
    if (c->dA) { (void)hipFree(c->dA); c->dA = nullptr; }
    HIPCHK(lf_dev_malloc(&c->dA, kappa * cnt * RE * sizeof(fe)));
    for (size_t i = 0; i < kappa; i++) RET(up_ring(c, A + (i * n + col0) * RE, cnt, c->dA + i * RE * cnt));
    HIPCHK(hipStreamSynchronize(c->stream()));
    c->kappa = (u32)kappa;
    c->nA = cnt; c->nA_total = n; c->A_col0 = col0;
    return prep_ajtai_i8(c);
}
int BbCtx::ajtai_generate(uint64_t seed, size_t kappa, size_t n) {
    C *c = p;
    if (kappa > 32) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    size_t col0, cnt;
    RET(shard_columns(c, n, &col0, &cnt));
    if (c->dA) { (void)hipFree(c->dA); c->dA = nullptr; }
    HIPCHK(lf_dev_malloc(&c->dA, kappa * cnt * RE * sizeof(fe)));
    launch_fill_ajtai(c->dA, (u32)kappa, cnt, n, col0, seed, c->stream());
    HIPCHK(hipStreamSynchronize(c->stream()));
    c->kappa = (u32)kappa;
    c->nA = cnt; c->nA_total = n; c->A_col0 = col0;
    return prep_ajtai_i8(c);
}
// General commitments from the resident byte planes of A (lf_ajtai_i8g.hip, shared with the Goldilocks backend): commit_ntt for `batch` vectors
// F [batch][72][ldF] in NTT form (pointing at this rank's first column), or Witness::commit for the centred int32 planes of a witness handle
// (F null, batch 1).  Five balanced base-128 digit planes cover the centred 31-bit residues.  out_dev: canonical u64 [batch][kappa][72], NTT form.
static int commit_dev_i8g(C *c, const fe *F, size_t ldF, u32 batch, const int32_t *planes, size_t ldp, u64 *out_dev, bool timed) {
    if (!c->i8_nch || !c->dAb) return LF_ERR_STATE;
    const lf::AjtaiI8Ring R = lf::ajtai_i8_babybear();
    const u32 nch = c->i8_nch, kc = c->i8_kc, MT = lf::ajtai_i8_row_tiles(R, kc);
    const size_t ntiles = (c->nA + 7) / 8, chunk_bytes = ntiles * (R.RD / 8) * MT * 1024;
    const u32 NP = planes ? lf::ajtai_i8g_planes_i32() : lf::ajtai_i8g_planes_general(R);
    const char *e_wgs = getenv("LF_I8G_WGS");           // (test hook: workgroups of the general commit kernel; default one per CU)
    const u32 nwg = e_wgs && atoi(e_wgs) > 0 ? (u32)atoi(e_wgs) : 256;
    size_t pw, dw, sw;
    if (lf::ajtai_i8g_scratch(R, MT, c->nA, NP, nwg, &pw, &dw, &sw) != 0) return LF_ERR_UNSUPPORTED;
    unsigned long long *pre;
    int32_t *part, *dsum;
    long long *sum;
    u64 *co;
    fe *cf, *ntt;
    RET(c->tbuf("i8g_pre", (size_t)NP * RE * ntiles, &pre));
    RET(c->tbuf("i8g_part", pw, &part));
    RET(c->tbuf("i8g_dsum", dw, &dsum));
    RET(c->tbuf("i8g_sum", sw, &sum));
    RET(c->tbuf("i8g_co", (size_t)RE * c->kappa, &co));
    RET(c->tbuf("i8g_cf", (size_t)RE * c->kappa, &cf));
    RET(c->tbuf("i8g_ntt", (size_t)RE * c->kappa, &ntt));
    for (u32 b = 0; b < batch; b++) {
        const size_t ev = timed ? c->ev_begin(1) : 0;   // the whole device side of one commitment: digit pass, contraction, recombination, CRT
        if (planes) lf::launch_i8g_cut_i32(planes, ldp, c->nA, RE, NP, pre, ntiles, c->stream());
        else launch_i8g_cut_ntt(c->d_icrt, c->d_icrt_sp_val, c->d_icrt_sp_col, F + (size_t)b * RE * ldF, ldF, c->nA, NP, pre, ntiles, c->stream());
        for (u32 ch = 0; ch < nch; ch++) {
            const u32 row0 = ch * kc, kn = c->kappa - row0 < kc ? c->kappa - row0 : kc;
            const int g = lf::launch_ajtai_i8g(R, c->dAb + (size_t)ch * chunk_bytes, MT, pre, ntiles, c->nA, kn, row0, c->kappa, NP, nwg, part, dsum, sum, co, c->stream());
            if (g < 0) return LF_ERR_UNSUPPORTED;
        }
        launch_aos_to_soa(co, cf, c->kappa, c->stream());       // canonical -> Montgomery planes
        launch_crt_fwd(c->dev, cf, ntt, c->kappa, c->stream());
        launch_soa_to_aos(ntt, out_dev + (size_t)b * c->kappa * RE, c->kappa, c->stream());
        if (timed) c->ev_end(ev);
    }
    return LF_OK;
}
// F: [batch][72][ldF]; out_dev: canonical u64 [batch][kappa][72]
static int commit_dev(C *c, const fe *F, size_t ldF, u32 batch, u64 *out_dev, bool timed) { return commit_dev_i8g(c, F, ldF, batch, nullptr, 0, out_dev, timed); }
int BbCtx::ajtai_commit(const uint64_t *f, size_t n, size_t batch, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->dAb) return LF_ERR_STATE;
    if (n != c->nA_total) return LF_ERR_INVALID;   // CommitmentError::WrongWitnessLength(n, width)
    HIPCHK(hipSetDevice(c->device));
    fe *F;
    u64 *o;
    RET(c->tbuf("io_a", batch * n * RE, &F));
    RET(c->tbuf("io_o", batch * c->kappa * RE, &o));
    for (size_t b = 0; b < batch; b++) RET(up_ring(c, f + b * n * RE, n, F + b * RE * n));
    c->ev_reset();
    RET(commit_dev(c, F + c->A_col0, n, (u32)batch, o, true));   // timed: lf_last_kernel_stats reports the stand-alone kernel
    c->ev_collect();
    RET(down_small(c, o, batch * c->kappa * RE, out));
    return exchange_modsum(c, out, batch * c->kappa * RE);
}

// ---- a8/a9/a11 ---------------------------------------------------------------------------------------------------------
// no host synchronisation: constants are staged in the lane's pinned arena (valid until the next fold step)
int build_eq_async(C *c, const H9 *pt, u32 nv, fe *eq_dev) {
    E9PreC *rd;
    RET(c->tbuf("eq_point_async", 2 * 64, &rd));
    size_t words = (2 * (size_t)nv * sizeof(E9PreC) + 7) / 8;
    E9PreC *h = (E9PreC *)c->arena_alloc(words);
    if (!h) return LF_ERR_HIP;
    for (u32 i = 0; i < nv; i++) {
        h[i] = e9pre_from_h9(pt[i], c->ring.T.nu);
        H9 om;
        for (int q = 0; q < TAU; q++) om.c[q] = hsub(q == 0 ? 1 : 0, pt[i].c[q]);
        h[nv + i] = e9pre_from_h9(om, c->ring.T.nu);
    }
    HIPCHK(hipMemcpyAsync(rd, h, 2 * (size_t)nv * sizeof(E9PreC), hipMemcpyHostToDevice, c->stream()));
    launch_build_eq(c->dev, rd, rd + nv, nv, eq_dev, c->stream());
    return LF_OK;
}
int build_eq_dev(C *c, const H9 *pt, u32 nv, fe *eq_dev) {
    E9PreC *rd;
    RET(c->tbuf("eq_point", 2 * 64, &rd));
    std::vector<E9PreC> h(2 * (size_t)nv);
    for (u32 i = 0; i < nv; i++) {
        h[i] = e9pre_from_h9(pt[i], c->ring.T.nu);
        h[nv + i] = e9pre_from_h9(h9_sub(h9_one(), pt[i]), c->ring.T.nu);
    }
    HIPCHK(hipMemcpyAsync(rd, h.data(), h.size() * sizeof(E9PreC), hipMemcpyHostToDevice, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    launch_build_eq(c->dev, rd, rd + nv, nv, eq_dev, c->stream());
    return LF_OK;
}
int BbCtx::build_eq(const uint64_t *point, unsigned nv, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    size_t n = (size_t)1 << nv;
    fe *eq;
    RET(c->tbuf("io_a", TAU * n, &eq));
    std::vector<H9> pt(nv);
    for (unsigned i = 0; i < nv; i++) pt[i] = h9_load(point + (size_t)TAU * i);
    RET(build_eq_dev(c, pt.data(), nv, eq));
    std::vector<fe> h(TAU * n);
    HIPCHK(hipMemcpyAsync(h.data(), eq, h.size() * sizeof(fe), hipMemcpyDeviceToHost, c->stream()));
    HIPCHK(hipStreamSynchronize(c->stream()));
    for (size_t i = 0; i < n; i++)
        for (int q = 0; q < TAU; q++) out[TAU * i + q] = to_canon(h[(size_t)q * n + i]);
    return LF_OK;
}
int BbCtx::mle_eval_batch(const uint64_t *tables, size_t ntables, size_t len, const uint64_t *point, unsigned nv, uint64_t *out) {
    C *c = p;
    size_t n = (size_t)1 << nv;
    if (len > n || len == 0) return LF_ERR_INVALID;   // MleEvaluationError::IncorrectLength
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *eq, *X;
    i64 *partial;
    u64 *o;
    RET(c->tbuf("io_eq", TAU * n, &eq));
    RET(c->tbuf("io_a", ntables * len * RE, &X));
    RET(c->tbuf("red_partial", red_partial_words((u32)(ntables * RE > 16 * RE * TAU ? ntables * RE : 16 * RE * TAU)), &partial));
    RET(c->tbuf("io_o", ntables * RE, &o));
    std::vector<H9> pt(nv);
    for (unsigned i = 0; i < nv; i++) pt[i] = h9_load(point + (size_t)TAU * i);
    RET(build_eq_dev(c, pt.data(), nv, eq));
    for (size_t a = 0; a < ntables; a++) RET(up_ring(c, tables + a * len * RE, len, X + a * RE * len));
    launch_dot_eq(c->dev, X, len, (u32)ntables, eq, n, len, partial, o, c->stream());
    return down_small(c, o, ntables * RE, out);
}

// ---- CCS ------------------------------------------------------------------------------------------------------------------
int BbCtx::ccs_load(const lf_params *P, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val,
                    const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *cc) {
    C *c = p;
    if (P->s < 3 || P->s > 28 || P->t == 0 || P->t > 4 || P->q == 0 || P->q > 8 || P->K == 0 || P->K > 16 || P->L == 0 || P->L > 8 ||
        P->d + 1 > 4 || P->wit_len == 0)
        return LF_ERR_UNSUPPORTED;
    if (P->b != 2) return LF_ERR_UNSUPPORTED;
    if (!pow2(P->B) || P->B > (1ULL << 30)) return LF_ERR_UNSUPPORTED;
    {
        u64 half = P->B / 2;
        u32 need = 0;
        while ((half >> need) != 0) need++;
        if (need > P->K) return LF_ERR_UNSUPPORTED;
    }
    size_t m = (size_t)1 << P->s, N = (size_t)P->wit_len * P->L, n = (size_t)P->l + 1 + P->wit_len;
    if (N > m) return LF_ERR_SIZE_BOUNDS;   // sanity_check, nifs.rs:165-173
    {
        u32 next = 0;
        for (u32 i = 0; i < P->q; i++)
            for (u32 k = S_off[i]; k < S_off[i + 1]; k++)
                if (S_idx[k] != next++) return LF_ERR_UNSUPPORTED;
        if (next != P->t || S_off[P->q] > 16) return LF_ERR_UNSUPPORTED;
    }
    RET(lf_validate_csr(P->t, m, n, rowptr, col, val, RE, BB_P));   // before any context state is touched
    for (size_t k = 0; k < (size_t)P->q * RE; k++)
        if (cc[k] >= BB_P) return LF_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    free_ccs(c);
    c->P = *P; c->N = N; c->m = m; c->n = n;
    memset(&c->desc, 0, sizeof(c->desc));
    c->desc.t = P->t; c->desc.q = P->q;
    for (u32 i = 0; i <= P->q; i++) c->desc.S_off[i] = S_off[i];
    for (u32 k = 0; k < S_off[P->q]; k++) c->desc.S_idx[k] = S_idx[k];
    for (u32 i = 0; i < P->q; i++) {
        u64 one[RE], mone[RE];
        BbHostRing::from_u64(1, one);
        BbHostRing::from_u64(BB_P - 1, mone);
        const u64 *ci = cc + (size_t)i * RE;
        for (int w = 0; w < RE; w++) c->desc.c[i][w] = from_canon(ci[w]);
        c->desc.c_unit[i] = !memcmp(ci, one, sizeof(one)) ? 1 : (!memcmp(ci, mone, sizeof(mone)) ? -1 : 0);
    }
    auto dalloc = [](auto &vec, size_t bytes) -> void * {   // registered in the context at once: a failure half-way leaks nothing
        void *ptr = nullptr;
        if (lf_dev_malloc(&ptr, bytes) != hipSuccess) return nullptr;
        vec.push_back((typename std::remove_reference<decltype(vec)>::type::value_type)ptr);
        return ptr;
    };
    for (u32 j = 0; j < P->t; j++) {
        size_t nnz = rowptr[j][m];
        std::vector<fe> v(nnz * RE + 1), vT(nnz * RE + 1);
        for (size_t k = 0; k < nnz * RE; k++) v[k] = from_canon(val[j][k]);
        std::vector<u32> cp(n + 1, 0), ri(nnz + 1);
        for (size_t k = 0; k < nnz; k++) cp[col[j][k] + 1]++;
        for (size_t i = 0; i < n; i++) cp[i + 1] += cp[i];
        std::vector<u32> fill(cp.begin(), cp.end() - 1);
        for (size_t r = 0; r < m; r++)
            for (u32 k = rowptr[j][r]; k < rowptr[j][r + 1]; k++) {
                u32 pos = fill[col[j][k]]++;
                ri[pos] = (u32)r;
                memcpy(&vT[(size_t)pos * RE], &v[(size_t)k * RE], RE * sizeof(fe));
            }
        void *drp = dalloc(c->d_rowptr, (m + 1) * 4), *dci = dalloc(c->d_col, (nnz + 1) * 4), *dv = dalloc(c->d_val, (nnz + 1) * RE * sizeof(fe));
        void *dcp = dalloc(c->d_colptr, (n + 1) * 4), *dri = dalloc(c->d_rowidx, (nnz + 1) * 4), *dvT = dalloc(c->d_valT, (nnz + 1) * RE * sizeof(fe));
        if (!drp || !dci || !dv || !dcp || !dri || !dvT) return LF_ERR_HIP;
        HIPCHK(hipMemcpy(drp, rowptr[j], (m + 1) * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dci, col[j], nnz * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dv, v.data(), nnz * RE * sizeof(fe), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dcp, cp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dri, ri.data(), nnz * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dvT, vT.data(), nnz * RE * sizeof(fe), hipMemcpyHostToDevice));
    }
    c->have_ccs = true;
    return LF_OK;
}
int BbCtx::spmv(unsigned j, const uint64_t *z, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    if (j >= c->P.t) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    fe *zd, *od;
    RET(c->tbuf("io_a", c->n * RE, &zd));
    RET(c->tbuf("io_b", c->m * RE, &od));
    RET(up_ring(c, z, c->n, zd));
    launch_spmv(c->dev, c->d_rowptr[j], c->d_col[j], c->d_val[j], zd, c->n, od, c->m, 0, c->stream());
    return down_ring(c, od, c->m, out);
}

// ---- witnesses -----------------------------------------------------------------------------------------------------------
static int witness_from_coef_table(C *c, const fe *coef_dev, lf_witness **out) {
    int32_t *pl;
    HIPCHK(lf_dev_malloc(&pl, c->N * RE * 4));
    int *viol;
    if (c->tbuf("small_dev", 4096, (u64 **)&viol) != LF_OK) { (void)hipFree(pl); return LF_ERR_HIP; }
    (void)hipMemsetAsync(viol, 0, 4, c->stream());
    launch_coef_to_i32(coef_dev, pl, c->N, (u32)(c->P.B / 2), viol, c->stream());
    int hv = 0;
    if (hipMemcpyAsync(&hv, viol, 4, hipMemcpyDeviceToHost, c->stream()) != hipSuccess || hipStreamSynchronize(c->stream()) != hipSuccess) {
        (void)hipFree(pl);
        return LF_ERR_HIP;
    }
    if (hv) { (void)hipFree(pl); return LF_ERR_NORM; }
    *out = new lf_witness{c->owner, pl, c->N, lf_ctx_device(c->owner), c->N * RE * 4};
    return LF_OK;
}
int BbCtx::witness_from_w_ccs(const uint64_t *w_ccs, lf_witness **out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    fe *a, *b, *d;   // Witness::from_w_ccs, arith.rs:230-248: ICRT -> gadget_decompose(B, L)
    RET(c->tbuf("io_a", (size_t)c->P.wit_len * RE, &a));
    RET(c->tbuf("io_b", (size_t)c->P.wit_len * RE, &b));
    RET(c->tbuf("io_c", c->N * RE, &d));
    RET(up_ring(c, w_ccs, c->P.wit_len, a));
    launch_icrt_dense(c->d_icrt, a, b, c->P.wit_len, c->stream());
    launch_decompose(b, c->P.wit_len, c->P.B, c->P.L, 0, d, c->stream(), c->digit_mode);
    return witness_from_coef_table(c, d, out);
}
int BbCtx::witness_from_f_coeff(const uint64_t *f_coeff, lf_witness **out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    fe *d;
    RET(c->tbuf("io_c", c->N * RE, &d));
    RET(up_ring(c, f_coeff, c->N, d));
    return witness_from_coef_table(c, d, out);
}
int BbCtx::witness_from_f(const uint64_t *f_ntt, lf_witness **out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    fe *a, *d;
    RET(c->tbuf("io_a", c->N * RE, &a));
    RET(c->tbuf("io_c", c->N * RE, &d));
    RET(up_ring(c, f_ntt, c->N, a));
    launch_icrt_dense(c->d_icrt, a, d, c->N, c->stream());
    return witness_from_coef_table(c, d, out);
}
int BbCtx::witness_get_f_coeff(const lf_witness *w, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    fe *d;
    RET(c->tbuf("io_c", w->N * RE, &d));
    launch_i32_to_coef(w->planes, d, w->N, c->stream());
    return down_ring(c, d, w->N, out);
}
int BbCtx::witness_get_f(const lf_witness *w, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(hipSetDevice(c->device));
    if (w->f_ntt) return down_ring(c, (const fe *)w->f_ntt, w->N, out);      // built inside the fold step that produced this witness
    fe *d, *e;
    RET(c->tbuf("io_c", w->N * RE, &d));
    RET(c->tbuf("io_b", w->N * RE, &e));
    launch_i32_to_coef(w->planes, d, w->N, c->stream());
    launch_crt_fwd(c->dev, d, e, w->N, c->stream());
    return down_ring(c, e, w->N, out);
}
int BbCtx::witness_get_w_ccs(const lf_witness *w, uint64_t *out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->have_ccs) return LF_ERR_STATE;
    HIPCHK(hipSetDevice(c->device));
    if (w->w_ccs && w->w_bytes == (size_t)c->P.wit_len * RE * sizeof(fe)) return down_ring(c, (const fe *)w->w_ccs, c->P.wit_len, out);
    fe *e;
    RET(c->tbuf("io_b", (size_t)c->P.wit_len * RE, &e));
    launch_recompose_crt(c->dev, w->planes, w->N, c->P.wit_len, c->P.L, c->P.B, 1, 0, e, c->P.wit_len, 0, c->stream());
    return down_ring(c, e, c->P.wit_len, out);
}
int BbCtx::witness_commit(const lf_witness *w, uint64_t *cm_out) {
    C *c = p;
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->dAb) return LF_ERR_STATE;
    if (w->N != c->nA_total) return LF_ERR_INVALID;
    HIPCHK(hipSetDevice(c->device));
    u64 *o;
    RET(c->tbuf("io_o", (size_t)c->kappa * RE, &o));
    // the int32 planes of the handle are the operand
    c->ev_reset();
    RET(commit_dev_i8g(c, nullptr, 0, 1, w->planes + c->A_col0, w->N, o, true));   // timed: lf_last_kernel_stats reports the stand-alone kernel
    c->ev_collect();
    RET(down_small(c, o, (size_t)c->kappa * RE, cm_out));
    return exchange_modsum(c, cm_out, (size_t)c->kappa * RE);
}

int BbCtx::last_phase_ms(float *out) {
    for (int i = 0; i < NPH; i++) out[i] = p->phase_ms[i];
    return LF_OK;
}
unsigned BbCtx::fold_paths() const { return p->sv_round_mask; }
unsigned BbCtx::fold_split_rounds() const { return p->fold_split_mask; }
int BbCtx::last_kernel_stats(float *fold_ms, int *fold_n, float *aj_ms, int *aj_n) {
    if (fold_ms) *fold_ms = p->k_fold_ms;
    if (fold_n) *fold_n = p->k_fold_n;
    if (aj_ms) *aj_ms = p->k_ajtai_ms;
    if (aj_n) *aj_n = p->k_ajtai_n;
    return LF_OK;
}

// ---- host-side verifier ------------------------------------------------------------------------------------------------------------
namespace {
struct BbV {
    static constexpr int RE = lfbb::RE, TAU = lfbb::TAU;
    static u64 modulus() { return BB_P; }
    typedef H9 Ext;
    typedef BbTranscript Tr;
    BbHostRing ring;
    void mul(const u64 *a, const u64 *b, u64 *o) const { ring.mul_ntt(a, b, o); }
    void mul_ext(const u64 *a, const Ext &s, u64 *o) const { ring.mul_h9(a, s, o); }
    static void add(const u64 *a, const u64 *b, u64 *o) { BbHostRing::add(a, b, o); }
    static void sub(const u64 *a, const u64 *b, u64 *o) { BbHostRing::sub(a, b, o); }
    static void from_u64(u64 v, u64 *o) { BbHostRing::from_u64(v, o); }
    static void from_ext(const Ext &e, u64 *o) { BbHostRing::from_h9(e, o); }
    static Ext ext_of(const u64 *e) { return h9_load(e); }
    static Ext ext_from_u64(u64 v) { Ext r; memset(&r, 0, sizeof(r)); r.c[0] = v % BB_P; return r; }
    Ext ext_mul(const Ext &a, const Ext &b) const { return ring.mul9(a, b); }
    static Ext ext_add(const Ext &a, const Ext &b) { Ext r; for (int i = 0; i < TAU; i++) r.c[i] = hadd(a.c[i], b.c[i]); return r; }
    static Ext ext_sub(const Ext &a, const Ext &b) { return h9_sub(a, b); }
    Ext ext_inv(const Ext &a) const {   // solve (multiplication-by-a) x = 1 over F_p
        u64 M[TAU][TAU + 1];
        Ext yb = ext_from_u64(0);
        yb.c[1] = 1;
        Ext cur = a;
        for (int j = 0; j < TAU; j++) {
            for (int i = 0; i < TAU; i++) M[i][j] = cur.c[i];
            cur = ring.mul9(cur, yb);
        }
        for (int i = 0; i < TAU; i++) M[i][TAU] = i == 0;
        for (int c = 0; c < TAU; c++) {
            int piv = -1;
            for (int r = c; r < TAU; r++) if (M[r][c]) { piv = r; break; }
            if (piv < 0) return ext_from_u64(0);
            if (piv != c) for (int k = 0; k <= TAU; k++) std::swap(M[piv][k], M[c][k]);
            u64 inv = hinv(M[c][c]);
            for (int k = 0; k <= TAU; k++) M[c][k] = hmul(M[c][k], inv);
            for (int r = 0; r < TAU; r++) {
                if (r == c || !M[r][c]) continue;
                u64 f = M[r][c];
                for (int k = 0; k <= TAU; k++) M[r][k] = hsub(M[r][k], hmul(f, M[c][k]));
            }
        }
        Ext r;
        for (int i = 0; i < TAU; i++) r.c[i] = M[i][TAU];
        return r;
    }
    static void absorb_ext(Tr &tr, const Ext &e) { tr.absorb_h9_as_ring(e); }
    void crt(const u64 *c, u64 *o) const { ring.crt(c, o); }
    static u64 fmul(u64 a, u64 b) { return hmul(a % BB_P, b % BB_P); }
    static u64 fadd(u64 a, u64 b) { return hadd(a, b); }
    static void rot_x(u64 *a) {   // multiply by X modulo X^72 - X^36 + 1
        u64 top = a[RE - 1];
        for (int j = RE - 1; j > 0; j--) a[j] = a[j - 1];
        a[0] = top ? BB_P - top : 0;
        a[RE / 2] = hadd(a[RE / 2], top);
    }
};
}  // namespace

int bb_verify_host(const lf_params *p, const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *c, BbTranscript &tr, const uint64_t *acc,
                   const uint64_t *cm_i, const uint64_t *proof, uint64_t *lcccs_out, int *failed_stage) {
    static const BbV *bv = [] {
        BbV *g = new BbV();
        u64 nr, y[8 * TAU];
        bb_default_ring(&nr, y);
        bb_build_tables(nr, y, g->ring.T);
        return g;
    }();
    lfv::Verifier<BbV> V(*bv, *p, S_off, S_idx, c);
    int rc = V.verify(tr, acc, cm_i, proof, lcccs_out);
    if (failed_stage) *failed_stage = V.stage;
    return rc;
}

}  // namespace lfbb


// bb_capi.h -- BabyBearRingNTT backend behind the C ABI (include/lfhip.h).  lf_capi.cpp forwards every entry point of a
// context created with lf_ctx_create_ring(.., LF_RING_BABYBEAR) to the matching BbCtx method.  Same flat layouts as the
// Goldilocks backend with ring elements of 72 words and tau = 9.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/lfhip.h"
#include "bb_host.h"

struct lf_witness;
namespace lfdist { struct Comm; }

namespace lfbb {

struct BbCtxImpl;

struct BbCtx {
    BbCtxImpl *p;
    static int create(BbCtx **out, lf_ctx *owner, int device);
    void destroy();

    int set_sharding(int rank, int world, lf_exchange_fn cb, void *user);
    int dist_init(int rank, int world, const uint8_t *id128);
    lfdist::Comm *comm();
    void set_digit_mode(int mode);
    bool have_ccs() const;
    const lf_params &params() const;
    size_t dim_n() const;
    size_t dim_m() const;
    size_t dim_N() const;
    uint32_t kappa() const;
    int set_ring_tables(uint64_t nonres, const uint64_t *y);
    int get_ring_tables(uint64_t *nonres, uint64_t *y);
    int synchronize();
    int mem_info(size_t *free_bytes, size_t *total_bytes);
    int selftest_field(uint64_t seed, uint32_t n, uint64_t *mismatches);
    int ntt_fwd(const uint64_t *in, uint64_t *out, size_t count);
    int ntt_inv(const uint64_t *in, uint64_t *out, size_t count);
    int decompose(const uint64_t *in, size_t count, uint64_t base, unsigned digits, int layout, uint64_t *out);
    int recompose(const uint64_t *in, size_t count_out, uint64_t base, unsigned digits, uint64_t *out);
    int linf_check(const uint64_t *f_ntt, size_t count, uint64_t bound, int unsigned_variant, int *ok, uint64_t *max_out);
    int ajtai_load(const uint64_t *A, size_t kappa, size_t n);
    int ajtai_generate(uint64_t seed, size_t kappa, size_t n);
    int ajtai_commit(const uint64_t *f, size_t n, size_t batch, uint64_t *out);
    int build_eq(const uint64_t *point, unsigned nv, uint64_t *out);
    int mle_eval_batch(const uint64_t *tables, size_t ntables, size_t len, const uint64_t *point, unsigned nv, uint64_t *out);
    int ccs_load(const lf_params *p, const uint32_t *const *rowptr, const uint32_t *const *col, const uint64_t *const *val,
                 const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *cc);
    int spmv(unsigned j, const uint64_t *z, uint64_t *out);
    int witness_from_w_ccs(const uint64_t *w_ccs, lf_witness **out);
    int witness_from_f_coeff(const uint64_t *f_coeff, lf_witness **out);
    int witness_from_f(const uint64_t *f_ntt, lf_witness **out);
    int witness_get_f_coeff(const lf_witness *w, uint64_t *out);
    int witness_get_f(const lf_witness *w, uint64_t *out);
    int witness_get_w_ccs(const lf_witness *w, uint64_t *out);
    int witness_commit(const lf_witness *w, uint64_t *cm_out);
    int sumcheck_lin_begin(const uint64_t *tables, const uint64_t *eq_point);
    int sumcheck_lin_round(const uint64_t *r_prev, uint64_t *evals_out);
    int sumcheck_lin_end();
    int sumcheck_fold_begin(const uint64_t *tables, const uint64_t *mu);
    int sumcheck_fold_round(const uint64_t *r_prev, uint64_t *evals_out);
    int sumcheck_fold_end();
    int lincomb(const uint64_t *coef, const uint64_t *tables, size_t n_terms, size_t len, uint64_t *out);
    int horner_combine(const uint64_t *tables, size_t groups, size_t per_group, size_t len, const uint64_t *challenges, uint64_t *out);
    int linearize(BbTranscript &tr, const uint64_t *cccs, const lf_witness *wit, uint64_t *lcccs_out, uint64_t *lin_proof_out);
    int fold_step(BbTranscript &tr, const uint64_t *acc, const lf_witness *w_acc, const uint64_t *cm_i, const lf_witness *w_i,
                  uint64_t *lcccs_out, lf_witness **w_out, uint64_t *proof);
    int decomposition_prove(BbTranscript &tr, const uint64_t *lcccs, const lf_witness *wit, uint64_t *lcccs_s_out, uint64_t *dec_proof_out);
    int folding_prove(BbTranscript &tr, const uint64_t *lcccs_s, const lf_witness *w_left, const lf_witness *w_right, uint64_t *lcccs_out,
                      lf_witness **w_out, uint64_t *fold_proof_out);
    int last_phase_ms(float *out);
    int last_kernel_stats(float *fold_ms, int *fold_n, float *aj_ms, int *aj_n);
    unsigned fold_split_rounds() const;   // table rounds of the last folding sumcheck that ran in the split eq form
    unsigned fold_paths() const;   // rounds of the last folding sumcheck that ran as int8 GEMMs (bit i-1 = round i)
};

int bb_verify_host(const lf_params *p, const uint32_t *S_off, const uint32_t *S_idx, const uint64_t *c, BbTranscript &tr, const uint64_t *acc,
                   const uint64_t *cm_i, const uint64_t *proof, uint64_t *lcccs_out, int *failed_stage);
size_t bb_lcccs_len(const lf_params *p);
size_t bb_cccs_len(const lf_params *p);
size_t bb_proof_len(const lf_params *p);

}  // namespace lfbb

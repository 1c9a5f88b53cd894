// libbean_hip.so: C ABI over the BEAN SVI kernels (see include/bean_hip.h).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the functions are resolved with dlsym (bean_hip_comm_*)

#include <string>
#include <vector>

#include "../../include/bean_hip.h"
#include "bean_kernels.hpp"

using namespace bean;

static thread_local std::string g_err;
static int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}
#define HIP_OK(expr)                                                                 \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess)                                                        \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct bean_hip_ctx {
    bean_hip_shape shape;
    DevArgs d;
    void* slot_ptr[BEAN_BUF_COUNT];
    uint64_t slot_bytes[BEAN_BUF_COUNT];
    void* workspace;
    uint64_t workspace_bytes;
    uint64_t loss_capacity;
    bool prepared;
    bool fused_guide;  // false: BEAN_HIP_GUIDE=split selects the sample / lik / pi-terms launches
    bool wave_guide;   // sorting variant families, default: one wave per (guide tile, replicate)
    bool wave2;        // ... in its second form, k_guide_wave2 (BEAN_HIP_GUIDE=wave1 selects the first)
    bool surv_wave;    // survival variant families: k_guide_survival_wave (BEAN_HIP_SURVIVAL=block: k_guide_survival)
    bool fused_step;   // bean_hip_svi_run steps with ONE launch, k_step_wave2 (bean_step_v2.hpp; BEAN_HIP_STEP=pair: two)
    std::vector<hipGraphExec_t> graphs_fused;  // [k]: 2^(k+1) launches of k_step_wave2
    long long* loss_acc;  // library-owned fixed-point loss accumulators, kLossWords per loss_hist slot
    int* tile_targets_dev;
    double* tsum_buf;   // DevArgs::tsum / tdesc (sized by bean_hip_prepare: they depend on tile_targets)
    int2* tdesc_buf;
    int* live_slots;   // tiling: compact list of the allele slots that hold an allele (own allocation, (A - 1) G ints)
    bool tiling_wave;  // tiling families, default: k_guide_tiling_wave (BEAN_HIP_TILING=block: k_guide_tiling)
    bool tiling_wide;  // more alleles per guide than this build's kAMax: bean_tiling_wide.hpp
    int tiling_rep_w;  // ... with this many waves per workgroup
    bool tiling_rep;   // tiling families, default: k_guide_tiling_rep, the replicates of a guide share a wave (bean_tiling_v2.hpp)
    // tiling: the allele tables (tabP, tabPmu, tabPy, mu_a, sig_a) belong to the draw that is on the device.  A PREP launch
    // of k_param<..., 3> with allele blocks leaves them so (round 5); any other PREP launch leaves a new draw and stale
    // tables, and launch_guide runs k_allele first.  BEAN_HIP_ALLELE=split: k_param never has allele blocks (A/B).
    bool alleles_fresh;
    bool allele_blocks;
    bool resume_head_fresh;  // the resume graphs were captured with fresh tables at their head (their first node is a guide launch)
    double* gsum_ws;  // library-owned normaliser buffer (replaced by BEAN_BUF_XCHG_GSUM when bound)
    double* sq_ws;    // library-owned projection sums (replaced by BEAN_BUF_XCHG_SQ when bound)
    double* cov_sum_ws;  // library-owned covariate gradient sums (replaced by BEAN_BUF_XCHG_COV when bound)
    // graph cache: graphs[k] replays 2^k {k_param, guide} pairs
    std::vector<hipGraphExec_t> graphs;
    unsigned long long graph_seed;
    // profiling of the dominant kernel
    bool profile;
    bool profile_param;  // profile mode 2: time k_param<FINISH, ADAM, PREP> instead of the guide kernel
    std::vector<hipEvent_t> ev;  // start/stop pairs
    // one launch per call for the variant sorting families (bean_tile_svi.hpp): target-aligned tiles
    bool tile_svi;            // eligible shape and not switched off (BEAN_HIP_STEP=pair)
    bool tile_ready;          // tile table built by bean_hip_prepare
    int* tile_tab;            // device: tile_g0[n_tiles + 1], tile_t0[n_tiles + 1]
    int n_tiles, tile_ntm, tile_gbm, tile_blocks;
    float* step_sizes;        // device: ClippedAdam step sizes of the steps of one call
    DevArgs* dargs_dev;       // device copy of DevArgs for k_svi_tile's out-of-line pieces
    uint64_t step_sizes_cap;
    std::vector<uint64_t> ev_steps;  // profile mode: SVI steps covered by each timed launch
    // RCCL communicator of a guide-sharded fit (bean_hip_comm_init) and graphs of 2^k exchanged steps
    ncclComm_t comm;
    int comm_world;
    std::vector<hipGraphExec_t> graphs_xchg;
    // bean_hip_svi_resume: graphs of 2^k {guide, k_param<FINISH, ADAM, PREP>} pairs, and what the last call left
    // prepared on the device: the draw and tables of step resume_next (same seed, same stream)
    std::vector<hipGraphExec_t> graphs_resume;
    bool resume_ok;
    uint64_t sharded_steps;  // bean_hip_sharded_update calls since bean_hip_sharded_begin: the slots its last call finalizes
    uint64_t resume_next, resume_seed;
    void* resume_stream;
    // all the steps of a call in ONE launch, tile-asynchronous (bean_async_v2.hpp): eligible shape and switched on
    bool async_step;
    int* async_ws;      // device: 8 queue counters (kAsyncQueueStride apart), the abort word, done[n_tiles]
    int async_blocks;   // grid: resident single-wave workgroups that pull items, a multiple of 8 (0: not yet measured)
    int async_fin_blocks;  // ... and that only finish tiles (0: no roles, the last arriver finishes)
    size_t async_ws_ints;
    unsigned long long* async_stamps;  // diagnostic builds (-DBEAN_ASYNC_STAMP): the last call's item timeline
    size_t async_stamp_words;
};

extern "C" const char* bean_hip_version(void) {
    return BEAN_AMAX <= 8 ? "bean_hip 0.1.0 (gfx950)"
                          : (BEAN_AMAX <= 16 ? "bean_hip 0.1.0 (gfx950, 16 alleles per guide / 16 conditions)"
                                             : "bean_hip 0.1.0 (gfx950, 32 alleles per guide / 16 conditions)");
}
extern "C" const char* bean_hip_last_error(void) { return g_err.c_str(); }

static bool is_survival(const bean_hip_shape& s) { return s.selection == BEAN_SELECTION_SURVIVAL; }
static bool is_tiling(const bean_hip_shape& s) { return s.family == BEAN_FAMILY_MULTI_MIXTURE; }
// families with a Dirichlet pi site (reporter models)
// survival NormalModel: the drawn initial guide abundance q_0 ~ Dirichlet over ALL guides enters
// the likelihood (survival_model.py:15-130, 629-648)
static bool is_surv_normal(const bean_hip_shape& s) {
    return s.selection == BEAN_SELECTION_SURVIVAL && s.family == BEAN_FAMILY_NORMAL;
}
static bool is_mixture(const bean_hip_shape& s) {
    return s.family == BEAN_FAMILY_MIXTURE_NORMAL || s.family == BEAN_FAMILY_MULTI_MIXTURE;
}

// Bytes the shape implies for a slot (0 = slot not used by this shape).
static uint64_t expected_bytes(const bean_hip_shape& s, int slot) {
    const uint64_t R = s.n_reps, B = s.n_condits, G = s.n_guides, T = s.n_targets;
    const uint64_t A = s.n_max_alleles, C = s.n_ctrl;
    auto param_elems = [&](int i) -> uint64_t {
        switch (i) {
            case 0: case 1: return T;
            case 2: case 3: return is_survival(s) ? 0 : T;  // survival models have no sd latent
            case 7: return ((is_survival(s) && s.family == BEAN_FAMILY_MIXTURE_NORMAL) || is_surv_normal(s)) ? G : 0;
            case 4: return is_mixture(s) ? G * A : 0;
            case 5: case 6:
                if (s.n_sample_covariates > 0) return (uint64_t)s.n_sample_covariates;  // mu_cov_loc / mu_cov_scale
                return (is_mixture(s) && (s.flags & BEAN_FLAG_SCALE_BY_ACC) && (s.flags & BEAN_FLAG_FIT_NOISE)) ? G : 0;
        }
        return 0;
    };
    if (slot >= BEAN_BUF_P_MU_LOC && slot < BEAN_BUF_P_MU_LOC + 8) return 4 * param_elems(slot - BEAN_BUF_P_MU_LOC);
    if (slot >= BEAN_BUF_G_MU_LOC && slot < BEAN_BUF_G_MU_LOC + 8) return 4 * param_elems(slot - BEAN_BUF_G_MU_LOC);
    if (slot >= BEAN_BUF_M_MU_LOC && slot < BEAN_BUF_M_MU_LOC + 8) return 4 * param_elems(slot - BEAN_BUF_M_MU_LOC);
    if (slot >= BEAN_BUF_V_MU_LOC && slot < BEAN_BUF_V_MU_LOC + 8) return 4 * param_elems(slot - BEAN_BUF_V_MU_LOC);
    switch (slot) {
        case BEAN_BUF_X: return 4 * R * B * G;
        case BEAN_BUF_X_BC: return (s.flags & BEAN_FLAG_USE_BCMATCH) ? 4 * R * B * G : 0;
        case BEAN_BUF_ALLELE_CTRL: return is_mixture(s) ? 4 * R * C * G * A : 0;
        case BEAN_BUF_REPGUIDE: return R * G;
        case BEAN_BUF_SIZE_FACTOR: return 8 * R * B;
        case BEAN_BUF_SIZE_FACTOR_BC: return (s.flags & BEAN_FLAG_USE_BCMATCH) ? 8 * R * B : 0;
        case BEAN_BUF_SAMPLE_MASK: return 8 * R * B;
        case BEAN_BUF_A0: return 8 * G;
        case BEAN_BUF_A0_BC: return (s.flags & BEAN_FLAG_USE_BCMATCH) ? 8 * G : 0;
        case BEAN_BUF_PI_A0: return is_mixture(s) ? 8 * G : 0;
        case BEAN_BUF_Z_HI: case BEAN_BUF_Z_LO: return is_survival(s) ? 0 : 8 * B;
        case BEAN_BUF_TIMEPOINTS: return is_survival(s) ? 8 * B : 0;
        case BEAN_BUF_CONTROL_TIME: return (is_survival(s) && is_mixture(s)) ? 8 * C : 0;
        case BEAN_BUF_LOG_OBS0: return (is_survival(s) && s.family == BEAN_FAMILY_MIXTURE_NORMAL) ? 8 * R * G : 0;
        case BEAN_BUF_X0_IN: case BEAN_BUF_X0_OUT:
            return ((is_survival(s) && s.family == BEAN_FAMILY_MIXTURE_NORMAL) || is_surv_normal(s)) ? 8 * R * G : 0;
        case BEAN_BUF_NEGCTRL_MASK: return is_surv_normal(s) ? G : 0;
        case BEAN_BUF_XCHG_GSUM:
            return ((is_survival(s) && s.family == BEAN_FAMILY_MIXTURE_NORMAL) || is_surv_normal(s)) ? 8 * (R + 1) : 0;
        case BEAN_BUF_XCHG_SQ: return is_surv_normal(s) ? 8 * R : 0;
        case BEAN_BUF_XCHG_TGRAD: return 8 * 2 * T;
        case BEAN_BUF_XCHG_COV: return s.n_sample_covariates > 0 ? 8 * R : 0;
        case BEAN_BUF_EPS_U_IN: case BEAN_BUF_EPS_U_OUT: return (is_survival(s) && is_mixture(s)) ? 8 * G : 0;
        case BEAN_BUF_GUIDE_IDS: return is_tiling(s) ? 4 * G : 0;
        case BEAN_BUF_PRIOR_IA: return (is_surv_normal(s) && s.prior_ia_total > 0.0) ? 8 * G : 0;
        case BEAN_BUF_TARGET_OFFSETS: return is_tiling(s) ? 0 : 4 * (T + 1);
        case BEAN_BUF_GUIDE_TO_TARGET: return is_tiling(s) ? 0 : 4 * G;
        case BEAN_BUF_A2E_PTR: return is_tiling(s) ? 4 * (G * (A - 1) + 1) : 0;
        case BEAN_BUF_A2E_IDX: case BEAN_BUF_E2A_IDX: return is_tiling(s) ? 4 * (uint64_t)s.n_a2e_nnz : 0;
        case BEAN_BUF_E2A_PTR: return is_tiling(s) ? 4 * ((uint64_t)s.n_edits + 1) : 0;
        case BEAN_BUF_ALLELE_MASK: return is_tiling(s) ? G * A : 0;
        case BEAN_BUF_ACCESSIBILITY: return (s.flags & BEAN_FLAG_SCALE_BY_ACC) ? 8 * G : 0;
        case BEAN_BUF_PRIOR_MU_LOC: case BEAN_BUF_PRIOR_MU_SCALE:
        case BEAN_BUF_PRIOR_SD_LOC: case BEAN_BUF_PRIOR_SD_SCALE: return 8 * T;
        case BEAN_BUF_EPS_MU_IN: case BEAN_BUF_EPS_MU_OUT: return 8 * T;
        case BEAN_BUF_EPS_SD_IN: case BEAN_BUF_EPS_SD_OUT: return is_survival(s) ? 0 : 8 * T;
        case BEAN_BUF_PI_IN: case BEAN_BUF_PI_OUT: return is_mixture(s) ? 8 * R * G * A : 0;
        case BEAN_BUF_EPS_NOISE_IN: case BEAN_BUF_EPS_NOISE_OUT:
            if (s.n_sample_covariates > 0) return 8 * (uint64_t)s.n_sample_covariates;  // eps of mu_cov
            return (s.flags & BEAN_FLAG_SCALE_BY_ACC) ? 8 * G : 0;
        case BEAN_BUF_REP_BY_COV: return 8 * R * (uint64_t)(s.n_sample_covariates > 0 ? s.n_sample_covariates : 0);
        case BEAN_BUF_LOSS_HIST: return 8;  // minimum; any multiple of 8 accepted
    }
    return 0;
}

static void sync_devargs(bean_hip_ctx* c) {
    DevArgs& d = c->d;
    auto P = [&](int s) { return c->slot_ptr[s]; };
    d.X = (const float*)P(BEAN_BUF_X);
    d.Xbc = (const float*)P(BEAN_BUF_X_BC);
    d.allele = (const float*)P(BEAN_BUF_ALLELE_CTRL);
    d.rg = (const uint8_t*)P(BEAN_BUF_REPGUIDE);
    d.sf = (const double*)P(BEAN_BUF_SIZE_FACTOR);
    d.sf_bc = (const double*)(P(BEAN_BUF_SIZE_FACTOR_BC) ? P(BEAN_BUF_SIZE_FACTOR_BC) : P(BEAN_BUF_SIZE_FACTOR));
    d.smask = (const double*)P(BEAN_BUF_SAMPLE_MASK);
    d.a0 = (const double*)P(BEAN_BUF_A0);
    d.a0_bc = (const double*)P(BEAN_BUF_A0_BC);
    d.pi_a0 = (const double*)P(BEAN_BUF_PI_A0);
    d.z_hi = (const double*)P(BEAN_BUF_Z_HI);
    d.z_lo = (const double*)P(BEAN_BUF_Z_LO);
    d.acc = (const double*)P(BEAN_BUF_ACCESSIBILITY);
    d.a2e_ptr = (const int*)P(BEAN_BUF_A2E_PTR);
    d.a2e_idx = (const int*)P(BEAN_BUF_A2E_IDX);
    d.e2a_ptr = (const int*)P(BEAN_BUF_E2A_PTR);
    d.e2a_idx = (const int*)P(BEAN_BUF_E2A_IDX);
    d.amask = (const uint8_t*)P(BEAN_BUF_ALLELE_MASK);
    d.toff = (const int*)P(BEAN_BUF_TARGET_OFFSETS);
    d.g2t = (const int*)P(BEAN_BUF_GUIDE_TO_TARGET);
    d.pr_mu_loc = (const double*)P(BEAN_BUF_PRIOR_MU_LOC);
    d.pr_mu_scale = (const double*)P(BEAN_BUF_PRIOR_MU_SCALE);
    d.pr_sd_loc = (const double*)P(BEAN_BUF_PRIOR_SD_LOC);
    d.pr_sd_scale = (const double*)P(BEAN_BUF_PRIOR_SD_SCALE);
    for (int i = 0; i < 8; ++i) {
        d.p[i] = (float*)P(BEAN_BUF_P_MU_LOC + i);
        d.g[i] = (float*)P(BEAN_BUF_G_MU_LOC + i);
        d.m[i] = (float*)P(BEAN_BUF_M_MU_LOC + i);
        d.v[i] = (float*)P(BEAN_BUF_V_MU_LOC + i);
    }
    d.eps_mu_in = (const double*)P(BEAN_BUF_EPS_MU_IN);
    d.eps_sd_in = (const double*)P(BEAN_BUF_EPS_SD_IN);
    d.pi_in = (const double*)P(BEAN_BUF_PI_IN);
    d.eps_noise_in = (const double*)P(BEAN_BUF_EPS_NOISE_IN);
    d.eps_mu_out = (double*)P(BEAN_BUF_EPS_MU_OUT);
    d.eps_sd_out = (double*)P(BEAN_BUF_EPS_SD_OUT);
    d.pi_out = (double*)P(BEAN_BUF_PI_OUT);
    d.eps_noise_out = (double*)P(BEAN_BUF_EPS_NOISE_OUT);
    d.loss_hist = (double*)P(BEAN_BUF_LOSS_HIST);
    d.loss_acc = c->loss_acc;
    d.time = (const double*)P(BEAN_BUF_TIMEPOINTS);
    d.negctrl = (const uint8_t*)P(BEAN_BUF_NEGCTRL_MASK);
    d.rbc = (const double*)P(BEAN_BUF_REP_BY_COV);
    d.prior_ia = (const double*)P(BEAN_BUF_PRIOR_IA);
    d.gid = (const int*)P(BEAN_BUF_GUIDE_IDS);
    d.prior_ia_total = c->shape.prior_ia_total;
    if (P(BEAN_BUF_XCHG_GSUM)) d.gsum = (double*)P(BEAN_BUF_XCHG_GSUM);
    else d.gsum = c->gsum_ws;
    if (P(BEAN_BUF_XCHG_SQ)) d.sq = (double*)P(BEAN_BUF_XCHG_SQ);
    else d.sq = c->sq_ws;
    if (c->cov_sum_ws) d.cov_sum = P(BEAN_BUF_XCHG_COV) ? (double*)P(BEAN_BUF_XCHG_COV) : c->cov_sum_ws;
    d.ctrl_time = (const double*)P(BEAN_BUF_CONTROL_TIME);
    d.log_obs0 = (const double*)P(BEAN_BUF_LOG_OBS0);
    d.x0_in = (const double*)P(BEAN_BUF_X0_IN);
    d.eps_u_in = (const double*)P(BEAN_BUF_EPS_U_IN);
    d.x0_out = (double*)P(BEAN_BUF_X0_OUT);
    d.eps_u_out = (double*)P(BEAN_BUF_EPS_U_OUT);
    d.flags = c->shape.flags & ~kDumpPi;
    if (d.pi_out && (c->shape.flags & BEAN_FLAG_DUMP_PI)) d.flags |= kDumpPi;
}

static void drop_graph(bean_hip_ctx* c) {
    for (hipGraphExec_t g : c->graphs)
        if (g) (void)hipGraphExecDestroy(g);
    c->graphs.clear();
    for (hipGraphExec_t g : c->graphs_fused)
        if (g) (void)hipGraphExecDestroy(g);
    c->graphs_fused.clear();
    for (hipGraphExec_t g : c->graphs_xchg)
        if (g) (void)hipGraphExecDestroy(g);
    c->graphs_xchg.clear();
    for (hipGraphExec_t g : c->graphs_resume)
        if (g) (void)hipGraphExecDestroy(g);
    c->graphs_resume.clear();
    c->resume_ok = false;  // (called whenever a buffer, the shape-dependent state or the seed changes)
}

// The captured launches hold DevArgs - and with it the seed - BY VALUE, and the four graph families share ONE
// graph_seed: every stepping entry point therefore drops ALL the families when the seed changes, whichever of them
// it replays itself (run(A), resume(B), run(B) used to replay run's graphs with A baked in).
static void drop_graph_on_seed_change(bean_hip_ctx* c, unsigned long long seed) {
    const bool any = !c->graphs.empty() || !c->graphs_fused.empty() || !c->graphs_xchg.empty() || !c->graphs_resume.empty();
    if (any && c->graph_seed != seed) drop_graph(c);
}

extern "C" int bean_hip_create(const bean_hip_shape* s, bean_hip_ctx** out) {
    if (!s || !out) return fail("bean_hip_create: null argument");
    if (s->selection != BEAN_SELECTION_SORTING && s->selection != BEAN_SELECTION_SURVIVAL)
        return fail("bean_hip_create: unknown selection");
    if (is_survival(*s) && !(s->negctrl_scale > 0.0)) return fail("bean_hip_create: negctrl_scale must be > 0");
    if (s->family < BEAN_FAMILY_NORMAL || s->family > BEAN_FAMILY_MULTI_MIXTURE)
        return fail("bean_hip_create: unknown family");
    if (s->n_reps < 1 || s->n_guides < 1 || s->n_targets < 1)
        return fail("bean_hip_create: R, G, T must be >= 1");
    if (s->n_condits < 1 || s->n_condits > kBCap)
        return fail("bean_hip_create: n_condits must be in [1, " + std::to_string(kBCap) +
                    "] (libbean_hip.so holds 8 conditions, libbean_hip_a16.so 64)");
    if (s->family == BEAN_FAMILY_MIXTURE_NORMAL && s->n_max_alleles != 2)
        return fail("bean_hip_create: MixtureNormal requires n_max_alleles == 2");
    if (is_tiling(*s)) {
        if (s->n_max_alleles < 2 || s->n_max_alleles > kWideMaxA)
            return fail("bean_hip_create: MultiMixtureNormal requires n_max_alleles in [2, " + std::to_string(kWideMaxA) +
                        "] (up to " + std::to_string(kAMax) + " alleles per guide run in the register-resident kernels "
                        "of this build, more in the allele-parallel ones)");
        if (s->n_edits < 1 || s->n_targets != s->n_edits)
            return fail("bean_hip_create: MultiMixtureNormal requires n_targets == n_edits >= 1");
        if (s->n_a2e_nnz < 0) return fail("bean_hip_create: n_a2e_nnz must be >= 0");
    }
    if (is_mixture(*s) && s->n_ctrl < 1) return fail("bean_hip_create: MixtureNormal requires n_ctrl >= 1");
    if (s->family == BEAN_FAMILY_CONTROL_NORMAL && s->n_targets != 1)
        return fail("bean_hip_create: ControlNormal requires n_targets == 1");
    if (!(s->lrd > 0.0) || !(s->initial_lr > 0.0)) return fail("bean_hip_create: lr and lrd must be > 0");
    if (s->n_sample_covariates < 0 || s->n_sample_covariates > 64)
        return fail("bean_hip_create: n_sample_covariates must be in [0, 64]");
    if (s->n_sample_covariates > 0 && !(s->family == BEAN_FAMILY_NORMAL && s->selection == BEAN_SELECTION_SORTING))
        return fail("bean_hip_create: sample covariates belong to the sorting NormalModel only (bean/model/model.py:73-91)");
    if (s->guide_offset < 0 || s->target_offset < 0 ||
        (s->n_guides_total > 0 && s->guide_offset + s->n_guides > s->n_guides_total))
        return fail("bean_hip_create: shard offsets outside the whole screen");

    bean_hip_ctx* c = new bean_hip_ctx();
    memset(&c->d, 0, sizeof(DevArgs));
    memset(c->slot_ptr, 0, sizeof(c->slot_ptr));
    memset(c->slot_bytes, 0, sizeof(c->slot_bytes));
    c->shape = *s;
    c->prepared = false;
    {
        // diagnostic A/B switch for the sorting variant families: the split (three launch) form
        const char* env = getenv("BEAN_HIP_SPLIT_GUIDE");
        const char* mode = getenv("BEAN_HIP_GUIDE");
        c->fused_guide = !((env && env[0] == '1') || (mode && !strcmp(mode, "split")));
        c->wave_guide = c->fused_guide;
        c->wave2 = !(mode && !strcmp(mode, "wave1"));
        const char* smode = getenv("BEAN_HIP_SURVIVAL");
        c->surv_wave = !(smode && !strcmp(smode, "block"));
        const char* tmode = getenv("BEAN_HIP_TILING");
        c->tiling_wave = !(tmode && !strcmp(tmode, "block"));
#if BEAN_AMAX > 8
        c->tiling_wave = true;  // this build has no block form
#endif
#ifndef BEAN_AB_KERNELS
        // the product build holds the default kernels only; the superseded forms these switches select
        // live in libbean_hip_ab.so (-DBEAN_AB_KERNELS; HipSVI(..., lib_variant="ab"))
        const char* stp = getenv("BEAN_HIP_STEP");
        if (!c->fused_guide || !c->wave2 || !c->surv_wave || !c->tiling_wave ||
            (stp && (!strcmp(stp, "fused") || !strcmp(stp, "tile")))) {
            delete c;
            return fail("bean_hip_create: BEAN_HIP_GUIDE / _SURVIVAL / _TILING=block / _STEP select A/B reference kernels, "
                        "which this build does not contain (build with -DBEAN_AB_KERNELS: libbean_hip_ab.so)");
        }
#endif
    }
    c->graph_seed = 0;
    c->resume_ok = false;
    c->resume_next = c->resume_seed = 0;
    c->sharded_steps = 0;
    c->resume_stream = nullptr;
    c->comm = nullptr;
    c->comm_world = 0;
    c->tile_svi = false;
    c->tile_ready = false;
    c->tile_tab = nullptr;
    c->live_slots = nullptr;
    c->alleles_fresh = false;
    c->resume_head_fresh = false;
    {
        const char* am = getenv("BEAN_HIP_ALLELE");
        c->allele_blocks = !(am && !strcmp(am, "split"));
    }
    c->n_tiles = c->tile_ntm = c->tile_gbm = c->tile_blocks = 0;
    c->step_sizes = nullptr;
    c->step_sizes_cap = 0;
    c->dargs_dev = nullptr;
    c->async_step = false;
    c->async_ws = nullptr;
    c->async_blocks = 0;
    c->async_fin_blocks = 0;
    c->async_ws_ints = 0;
    c->async_stamps = nullptr;
    c->async_stamp_words = 0;
    c->loss_acc = nullptr;
    c->profile = false;
    c->profile_param = false;
    c->loss_capacity = 0;
    DevArgs& d = c->d;
    d.R = s->n_reps; d.B = s->n_condits; d.G = s->n_guides; d.T = s->n_targets;
    d.A = s->n_max_alleles; d.C = s->n_ctrl; d.E = s->n_edits;
    d.family = s->family; d.flags = s->flags; d.mask_thres = s->mask_thres;
    d.wide_targets = (!is_tiling(*s) && (s->n_targets < 64 || s->max_target_len > 256)) ? 1 : 0;
    // lanes per target of k_param's target part: 4 where a target has few rows to add (survival: ~15
    // (guide, replicate) rows; tiling: the alleles carrying an edit - unless the table is unfiltered and an
    // edit sits in dozens of them), 16 otherwise (sorting variant families: the Phi table's 2 B edges)
    d.lpt = kLanesPerTarget;
    if (is_tiling(*s)) {
        if ((int64_t)s->n_a2e_nnz <= 16 * (int64_t)s->n_targets) d.lpt = kLanesPerTargetNarrow;
    } else if (is_survival(*s)) {
        d.lpt = kLanesPerTargetNarrow;
    }
    d.g_off = s->guide_offset; d.t_off = s->target_offset;
    d.G_tot = s->n_guides_total > 0 ? s->n_guides_total : s->n_guides;
    d.sd_prior_scale = s->sd_prior_scale; d.lr0 = s->initial_lr; d.log_lrd = log(s->lrd);
    d.clip = s->clip_norm;
    d.survival = is_survival(*s) ? 1 : 0;
    d.neg_loc = s->negctrl_loc; d.neg_scale = s->negctrl_scale;

    const uint64_t B = d.B, T = d.T, G = d.G;
    const uint64_t A1 = is_tiling(*s) ? (uint64_t)(d.A - 1) : 0;
    const uint64_t n_cov = (uint64_t)s->n_sample_covariates;
    // table columns: allele slots or targets (per replicate with sample covariates)
    const uint64_t n_tab = is_tiling(*s) ? A1 * G : T * (n_cov ? (uint64_t)s->n_reps : 1);
    const uint64_t n_part = is_tiling(*s) ? (uint64_t)(s->n_max_alleles > kAMax ? tq_num(s->n_max_alleles) : kTNumPart)
                                          : (uint64_t)kNumPart;
    const bool surv_mix = is_survival(*s) && s->family == BEAN_FAMILY_MIXTURE_NORMAL;
    const bool surv_tiling = is_survival(*s) && is_tiling(*s);
    const bool surv_norm = is_surv_normal(*s);
    d.surv_q0lik = surv_norm ? 1 : 0;
    d.not_loss_owner = (s->flags & BEAN_FLAG_NOT_LOSS_OWNER) ? 1 : 0;
    c->gsum_ws = nullptr;
    c->sq_ws = nullptr;
    c->cov_sum_ws = nullptr;
    const uint64_t Rr = d.R;
    // guide blocks of k_param: kParamBlock guides each.  The survival families with a
    // Dirichlet-over-all-guides site have q0 blocks as well (after the alpha_pi blocks of MixtureNormal),
    // kParamBlock guides each: parameter update, gamma draws and normalisers of that site
    // (q0_draws_and_totals)
    d.q0_blocks = (surv_mix || surv_norm) ? 1 : 0;
    d.q0_blk0 = surv_mix ? (int)((G + kParamBlock - 1) / kParamBlock) : 0;
    const uint64_t n_gblk = (G + kParamBlock - 1) / kParamBlock;
    d.n_gamma_blocks = (int)n_gblk;
    const uint64_t n_surv = (surv_mix ? 2 * G + Rr * G + n_gblk * (Rr + 1) + (Rr + 1) : 0) +
                            (surv_norm ? 2 * Rr * G + n_gblk * (Rr + 1) + (Rr + 1) + Rr : 0) +
                            (surv_tiling ? 2 * G : 0);
    const bool split_ok = !is_survival(*s) && !is_tiling(*s);
    const bool use_split = split_ok && !c->fused_guide;
    c->wave_guide = c->wave_guide && split_ok;
    c->wave2 = c->wave2 && c->wave_guide;
    if (s->n_sample_covariates > 0 && !c->wave2) {
        delete c;
        return fail("bean_hip_create: sample covariates need the default guide kernel (k_guide_wave2)");
    }
    d.n_cov = s->n_sample_covariates;
    // k_guide_wave2's tiles follow the GLOBAL guide index (bean_guide_v2.hpp): a shard that does not start at a
    // multiple of 64 has g_sh empty lanes at the head of its first tile
    d.g_sh = c->wave2 ? s->guide_offset % 64 : 0;
    d.n_tiles = (int)((d.g_sh + G + 63) / 64);
    {
        const int seg = s->max_target_len < 1 ? 64 : (s->max_target_len > 64 ? 64 : s->max_target_len);
        d.seg_steps = 0;
        while ((1 << d.seg_steps) < seg) ++d.seg_steps;
    }
    c->tsum_buf = nullptr;
    c->tdesc_buf = nullptr;
    c->surv_wave = c->surv_wave && is_survival(*s) && !is_tiling(*s);
    d.rows_v2 = (c->wave2 || c->surv_wave) ? 1 : 0;

    c->tiling_wide = is_tiling(*s) && s->n_max_alleles > kAMax;
    c->tiling_wave = c->tiling_wave && is_tiling(*s) && !c->tiling_wide;
    {
        // default: the replicates of a guide share a wave and every row is reduced there
        // (BEAN_HIP_TILING=wave: the per-replicate wave form + k_sum_trow, the A/B reference)
        const char* tmode = getenv("BEAN_HIP_TILING");
        c->tiling_rep = c->tiling_wave && s->n_reps <= kTilingRepMaxR && !(tmode && !strcmp(tmode, "wave"));
        if (c->tiling_rep) c->tiling_wave = false;
        // BEAN_HIP_TILING_W (1, 2 or 4; experiments only) overrides the number of waves per workgroup
        const int w_env = getenv("BEAN_HIP_TILING_W") ? atoi(getenv("BEAN_HIP_TILING_W")) : 0;
        long n_simd = 0;
        {
            int dev = 0, n_cu = 0;
            if (hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
                n_simd = 4l * n_cu;
            else
                (void)hipGetLastError();
        }
        c->tiling_rep_w = (w_env == 1 || w_env == 2 || w_env == 4)
                              ? w_env
                              : tiling_rep_waves(s->n_reps, s->n_condits, (s->flags & BEAN_FLAG_SCALE_BY_ACC) != 0,
                                                 (long)s->n_guides, n_simd);
        if (s->n_condits > 16) c->tiling_rep_w = 1;  // LDS: (3 B + ...) x 64 W doubles per workgroup
    }
    d.wide_alleles = c->tiling_wide ? 1 : 0;
    d.trow_summed = (c->tiling_wave || c->tiling_rep) ? 1 : 0;
    {
        const char* tc = getenv("BEAN_HIP_TOT_CONST");  // =0: the total terms are evaluated every step (A/B)
        d.tot_const = ((c->wave2 || c->surv_wave || c->tiling_wave || c->tiling_rep || c->tiling_wide) && !(tc && !strcmp(tc, "0"))) ? 1 : 0;
    }
#ifdef BEAN_AB_KERNELS
    {
        // one launch per step (bean_step_v2.hpp), opt-in (BEAN_HIP_STEP=fused; measured slower than the
        // two-launch path, see the header): the variant sorting families of k_guide_wave2 whose
        // parameters are all per target or per guide, no target longer than a tile
        const char* sm = getenv("BEAN_HIP_STEP");
        c->fused_step = c->wave2 && !d.wide_targets && s->max_target_len <= 64 && s->n_sample_covariates == 0 &&
                        (s->family == BEAN_FAMILY_MIXTURE_NORMAL || s->family == BEAN_FAMILY_NORMAL) &&
                        (sm && !strcmp(sm, "fused"));
    }
    {
        // OPT-IN (BEAN_HIP_STEP=tile): ONE launch per call, a workgroup per tile of targets
        // (bean_tile_svi.hpp).  Bit-identical to the two launches per step and measured slower (see the header).
        const char* sm = getenv("BEAN_HIP_STEP");
        const int gb = s->n_reps <= 64 ? kTileThreads / s->n_reps : 0;
        c->tile_svi = c->wave2 && !d.wide_targets && s->n_sample_covariates == 0 && !c->fused_step &&
                      (s->family == BEAN_FAMILY_MIXTURE_NORMAL || s->family == BEAN_FAMILY_NORMAL) &&
                      gb >= 1 && s->max_target_len <= gb && (sm && !strcmp(sm, "tile"));
    }
#else
    c->fused_step = false;
    c->tile_svi = false;
#endif
    {
        // ALL the steps of a call in one launch, no grid-wide boundary between steps (bean_async_v2.hpp): the variant
        // sorting families of k_guide_wave2 whose parameters are all per target or per guide, no target longer than a
        // tile.  BEAN_HIP_STEP=async switches it on, =pair off.
        const char* sm = getenv("BEAN_HIP_STEP");
        const long items = (long)d.n_tiles * s->n_reps;
        int dev_ = 0, n_cu_ = 256;
        if (hipGetDevice(&dev_) != hipSuccess || hipDeviceGetAttribute(&n_cu_, hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess)
            n_cu_ = 256;
        (void)hipGetLastError();
        const bool on = sm ? !strcmp(sm, "async") : async_size_in_range(items, 4l * n_cu_);
        c->async_step = on && c->wave2 && !d.wide_targets && s->max_target_len <= 64 && s->n_sample_covariates == 0 &&
                        !c->fused_step && !c->tile_svi &&
                        (s->family == BEAN_FAMILY_MIXTURE_NORMAL || s->family == BEAN_FAMILY_NORMAL) && !is_survival(*s);
    }
    const uint64_t n_trow = c->tiling_wave ? (uint64_t)kTNumPart * Rr * G
                                           : (c->tiling_wide ? (uint64_t)tq_num(d.A) * Rr * G : 0);
    const uint64_t n_split = use_split ? (3 + (is_mixture(*s) ? 4 : 0)) * Rr * G
                                       : (c->wave_guide ? (uint64_t)(c->wave2 ? kW2Rows : kNumPart + 2) * Rr * G
                                                        : (c->surv_wave ? (uint64_t)kW2Rows * Rr * G : 0));
#ifdef BEAN_STAMP
    const uint64_t n_dbg = 8 * 2 * Rr * (((uint64_t)((G + 63) / 64 + 1) + 7) / 8 * 8);
#else
    const uint64_t n_dbg = 0;
#endif
    // per-wave loss parts of the wave-form guide kernels (1-D grid of padded tiles x replicates)
    const uint64_t n_lpart = (c->wave2 || c->surv_wave) ? ((uint64_t)d.n_tiles + 7) / 8 * 8 * Rr : 0;
    const uint64_t n_dgq = ((c->wave2 || c->surv_wave) && s->family == BEAN_FAMILY_MIXTURE_NORMAL) ? 6 * G : 0;
    const uint64_t n_dgq_t = (c->tiling_wave || c->tiling_rep) ? (uint64_t)(kAMax + 1) * G : 0;
    // arrival counters of the fused step kernel: per tile and per tile boundary (ints, zero between launches)
    const uint64_t n_ctr = c->wave2 ? ((uint64_t)d.n_tiles + 7) / 8 * 8 + 2 + 3 * B + 2 : 0;
    const uint64_t n_kacc = (s->flags & BEAN_FLAG_SCALE_BY_ACC) ? G : 0;
    // tiling, k_param's allele blocks: arrival counters and go flags, one 128-byte line each (kAlleleCtrLines)
    const uint64_t n_actr = (!c->wave2 && is_tiling(*s)) ? (uint64_t)kAlleleCtrLines * 16 : 0;
    const uint64_t n_dbl = n_actr + n_kacc + n_ctr + 3 * B * n_tab + B + 4 * T + n_part * G + 2 * G + 2 * A1 * G + 2 + 8 + kLossWords + n_surv +
                           n_split + n_dbg + n_trow + 3 * n_lpart + n_dgq + n_dgq_t + 2 * n_cov + 2 * Rr + 1;
    c->workspace_bytes = n_dbl * 8;
    hipError_t e = hipMalloc(&c->workspace, c->workspace_bytes);
    if (e != hipSuccess) {
        delete c;
        return fail(std::string("hipMalloc workspace: ") + hipGetErrorString(e));
    }
    e = hipMemset(c->workspace, 0, c->workspace_bytes);
    if (e != hipSuccess) {
        (void)hipFree(c->workspace);
        delete c;
        return fail(std::string("hipMemset workspace: ") + hipGetErrorString(e));
    }
    double* w = (double*)c->workspace;
    d.tabP = w; w += B * n_tab;
    d.tabPmu = w; w += B * n_tab;
    d.tabPy = w; w += B * n_tab;
    d.P0 = w; w += B;
    d.mu_t = w; w += T;
    d.y_t = w; w += T;
    d.eps_mu = w; w += T;
    d.eps_sd = w; w += T;
    d.part = w; w += n_part * G;
    d.mu_a = w; w += A1 * G;
    d.sig_a = w; w += A1 * G;
    d.lpn = w; w += G;
    d.eps_noise = w; w += G;
    if (n_kacc) {
        d.kacc = w; w += G;  // filled by bean_hip_prepare (k_acc_scale)
    }
    d.loss_const = w; w += 1;
    d.const_acc = (long long*)w; w += kLossWords;
    c->tile_targets_dev = (int*)w; w += 1;
    d.tile_targets = 64;
    if (n_ctr) {
        d.tile_ctr = (int*)w;
        d.bnd_ctr = d.tile_ctr + (n_ctr - 3 * B - 2);
        d.n_arrival_ctr = (int)(2 * (n_ctr - 3 * B - 2));  // ints: two per double of this region
        w += n_ctr - 3 * B - 2;
        d.ue_z = w; w += 2 * B;
        d.ue_idx = (int*)w; w += B + 2;  // 2 B + 1 ints
    }
    if (n_actr) {  // (k_set_step zeroes tile_ctr[0 .. n_arrival_ctr))
        d.tile_ctr = (int*)w;
        d.n_arrival_ctr = (int)(2 * n_actr);
        w += n_actr;
    }
    if (n_dbg) {
        d.dbg = (unsigned long long*)w; w += n_dbg;
    }
    if (c->tiling_wave || c->tiling_wide) {
        d.trow = w; w += n_trow;
    }
    if (c->wave_guide && c->wave2) {
        d.wrow = w; w += (uint64_t)kW2Rows * Rr * G;  // count totals are re-summed in the kernel: no nobs
    } else if (c->wave_guide) {
        d.wrow = w; w += (uint64_t)kNumPart * Rr * G;
        d.nobs = w; w += 2 * Rr * G;
    } else if (c->surv_wave) {
        d.wrow = w; w += (uint64_t)kW2Rows * Rr * G;
    }
    if (use_split) {
        d.rrow = w; w += 3 * Rr * G;
        if (is_mixture(*s)) {
            d.pi_ws = w; w += 2 * Rr * G;
            d.gpi_ws = w; w += 2 * Rr * G;
        }
    }
    if (surv_mix) {
        d.u_g = w; w += G;
        d.eps_u = w; w += G;
        d.gam = w; w += Rr * G;
        d.gpart = w; w += n_gblk * (Rr + 1);
        d.gsum = w; w += Rr + 1;
        c->gsum_ws = d.gsum;
    }
    if (surv_tiling) {
        d.u_g = w; w += G;
        d.eps_u = w; w += G;
    }
    if (surv_norm) {
        d.gam = w; w += Rr * G;
        d.gpart = w; w += n_gblk * (Rr + 1);
        d.gsum = w; w += Rr + 1;
        c->gsum_ws = d.gsum;
        d.gq = w; w += Rr * G;
        d.sq = w; w += Rr;
        c->sq_ws = d.sq;
    }
    if (n_cov) {
        d.cov_mu = w; w += n_cov;
        d.cov_eps = w; w += n_cov;
        d.cov_shift = w; w += Rr;
        d.cov_sum = w; w += Rr;
        c->cov_sum_ws = d.cov_sum;
    }
    if (n_dgq) {
        d.dgq = w; w += n_dgq;
    }
    if (n_dgq_t) {
        d.dgq_t = w; w += n_dgq_t;
    }
    if (n_lpart) {
        d.lpart = (long long*)w; w += 3 * n_lpart;
        d.n_lpart = (int)n_lpart;
    }
    d.q0_ctr = (int*)w; w += 1;
    static_assert(sizeof(StepCtr) == 24, "two StepCtr take 6 of the 8 spare workspace doubles");
    d.ctrA = (StepCtr*)w; w += 3;
    d.ctrB = (StepCtr*)w; w += 3;
    *out = c;
    return 0;
}

extern "C" int bean_hip_comm_destroy(bean_hip_ctx* c);
extern "C" int bean_hip_destroy(bean_hip_ctx* c) {
    if (!c) return 0;
    drop_graph(c);
    (void)bean_hip_comm_destroy(c);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->workspace) (void)hipFree(c->workspace);
    if (c->tile_tab) (void)hipFree(c->tile_tab);
    if (c->live_slots) (void)hipFree(c->live_slots);
    if (c->tsum_buf) (void)hipFree(c->tsum_buf);
    if (c->tdesc_buf) (void)hipFree(c->tdesc_buf);
    if (c->step_sizes) (void)hipFree(c->step_sizes);
    if (c->dargs_dev) (void)hipFree(c->dargs_dev);
    if (c->async_ws) (void)hipFree(c->async_ws);
    if (c->loss_acc) (void)hipFree(c->loss_acc);
    delete c;
    return 0;
}

extern "C" int bean_hip_bind(bean_hip_ctx* c, int slot, void* ptr, uint64_t nbytes) {
    if (!c) return fail("bean_hip_bind: null handle");
    if (slot < 0 || slot >= BEAN_BUF_COUNT) return fail("bean_hip_bind: slot out of range");
    if (ptr) {
        const uint64_t want = expected_bytes(c->shape, slot);
        if (want == 0)
            return fail("bean_hip_bind: slot " + std::to_string(slot) + " is not used by this shape");
        if (slot == BEAN_BUF_LOSS_HIST) {
            if (nbytes < 8 || nbytes % 8) return fail("bean_hip_bind: loss_hist must hold >= 1 double");
            if (nbytes / 8 != c->loss_capacity || !c->loss_acc) {
                if (c->loss_acc) (void)hipFree(c->loss_acc);
                c->loss_acc = nullptr;
                HIP_OK(hipMalloc((void**)&c->loss_acc, (nbytes / 8) * kLossSub * kLossWords * sizeof(long long)));
                HIP_OK(hipMemset(c->loss_acc, 0, (nbytes / 8) * kLossSub * kLossWords * sizeof(long long)));
            }
            c->loss_capacity = nbytes / 8;
        } else if (nbytes != want) {
            return fail("bean_hip_bind: slot " + std::to_string(slot) + " expects " + std::to_string(want) +
                        " bytes, got " + std::to_string(nbytes));
        }
    }
    c->slot_ptr[slot] = ptr;
    c->slot_bytes[slot] = ptr ? nbytes : 0;
    sync_devargs(c);
    drop_graph(c);
    if (slot < BEAN_BUF_P_MU_LOC && slot != BEAN_BUF_XCHG_GSUM && slot != BEAN_BUF_XCHG_TGRAD &&
        slot != BEAN_BUF_XCHG_SQ)  // (BEAN_BUF_XCHG_COV lies above the parameter slots)
        c->prepared = false;  // data changed: the data-only precomputation is stale
    return 0;
}

static int require(bean_hip_ctx* c, int slot, const char* what) {
    if (!c->slot_ptr[slot]) return fail(std::string("required buffer not bound: ") + what);
    return 0;
}

static int check_bound(bean_hip_ctx* c, bool need_grads, bool need_moments) {
    const bean_hip_shape& s = c->shape;
#define REQ(slot) if (require(c, slot, #slot)) return -1
    REQ(BEAN_BUF_X); REQ(BEAN_BUF_REPGUIDE); REQ(BEAN_BUF_SIZE_FACTOR); REQ(BEAN_BUF_SAMPLE_MASK);
    REQ(BEAN_BUF_A0); REQ(BEAN_BUF_LOSS_HIST);
    if (is_survival(s)) {
        REQ(BEAN_BUF_TIMEPOINTS);
        if (is_mixture(s)) REQ(BEAN_BUF_CONTROL_TIME);
        if (s.family == BEAN_FAMILY_MIXTURE_NORMAL) REQ(BEAN_BUF_LOG_OBS0);
    } else {
        REQ(BEAN_BUF_Z_HI); REQ(BEAN_BUF_Z_LO);
    }
    if (is_tiling(s)) {
        REQ(BEAN_BUF_A2E_PTR); REQ(BEAN_BUF_E2A_PTR); REQ(BEAN_BUF_ALLELE_MASK);
        if (s.n_a2e_nnz > 0) { REQ(BEAN_BUF_A2E_IDX); REQ(BEAN_BUF_E2A_IDX); }
    } else {
        REQ(BEAN_BUF_TARGET_OFFSETS); REQ(BEAN_BUF_GUIDE_TO_TARGET);
    }
    if (s.flags & BEAN_FLAG_USE_BCMATCH) { REQ(BEAN_BUF_X_BC); REQ(BEAN_BUF_SIZE_FACTOR_BC); REQ(BEAN_BUF_A0_BC); }
    if (is_mixture(s)) { REQ(BEAN_BUF_ALLELE_CTRL); REQ(BEAN_BUF_PI_A0); }
    if (s.flags & BEAN_FLAG_SCALE_BY_ACC) REQ(BEAN_BUF_ACCESSIBILITY);
    if (s.n_sample_covariates > 0) REQ(BEAN_BUF_REP_BY_COV);
    if (is_surv_normal(s) && s.prior_ia_total > 0.0) REQ(BEAN_BUF_PRIOR_IA);
    for (int i = 0; i < 8; ++i) {
        if (expected_bytes(s, BEAN_BUF_P_MU_LOC + i) == 0) continue;
        REQ(BEAN_BUF_P_MU_LOC + i);
        if (need_grads) REQ(BEAN_BUF_G_MU_LOC + i);
        if (need_moments) { REQ(BEAN_BUF_M_MU_LOC + i); REQ(BEAN_BUF_V_MU_LOC + i); }
    }
#undef REQ
    return 0;
}

extern "C" int bean_hip_prepare(bean_hip_ctx* c, void* stream_) {
    if (!c) return fail("bean_hip_prepare: null handle");
    if (check_bound(c, false, false)) return -1;
    c->resume_ok = false;  // (bean_hip.h: a prepare between two windows forbids a resume, in every family)
    c->alleles_fresh = false;  // (tiling: the allele tables belong to no draw yet)
    hipStream_t stream = (hipStream_t)stream_;
    HIP_OK(hipMemsetAsync(c->d.const_acc, 0, kLossWords * sizeof(long long), stream));
    const long n = (long)c->d.R * c->d.G;
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c->d);
    HIP_OK(hipGetLastError());
    if (c->d.kacc) {
        if (!c->d.acc) return fail("bean_hip_prepare: BEAN_FLAG_SCALE_BY_ACC needs BEAN_BUF_ACCESSIBILITY");
        hipLaunchKernelGGL(k_acc_scale, dim3((unsigned)((c->d.G + 255) / 256)), dim3(256), 0, stream, c->d.acc, c->d.G,
                           c->d.kacc);
        HIP_OK(hipGetLastError());
    }
    if (c->wave_guide) {
        // LDS sizing of k_guide_wave: a host read-back (4 bytes, setup only; the other one is the tiling work list below)
        HIP_OK(hipMemsetAsync(c->tile_targets_dev, 0, 2 * sizeof(int), stream));
        const int tiles = c->wave2 ? c->d.n_tiles : (c->d.G + 63) / 64;
        hipLaunchKernelGGL(k_tile_targets, dim3((tiles + 255) / 256), dim3(256), 0, stream, c->d.g2t, c->d.G,
                           c->d.g_sh, c->tile_targets_dev);
        HIP_OK(hipGetLastError());
        // the shape's max_target_len decides the kernels' layouts (scan steps, two slots per target): checked
        hipLaunchKernelGGL(k_max_target_len, dim3((c->d.T + 255) / 256), dim3(256), 0, stream, c->d.toff, c->d.T,
                           c->tile_targets_dev + 1);
        HIP_OK(hipGetLastError());
        int ntl[2] = {0, 0};
        HIP_OK(hipMemcpyAsync(ntl, c->tile_targets_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        const int nt = ntl[0];
        if (nt < 1 || nt > 64) return fail("bean_hip_prepare: guides are not sorted by target (BEAN_BUF_GUIDE_TO_TARGET)");
        if (ntl[1] > c->shape.max_target_len)
            return fail("bean_hip_prepare: bean_hip_shape.max_target_len = " + std::to_string(c->shape.max_target_len) +
                        " but BEAN_BUF_TARGET_OFFSETS holds a target of " + std::to_string(ntl[1]) + " guides");
        if (nt != c->d.tile_targets) drop_graph(c);
        c->d.tile_targets = nt;
        if (c->wave2) {
            // per-target-part sums of d/dmu_t, d/dy_t (k_guide_wave2 -> k_param) and where each target's lie
            // no target longer than a tile (thin mode): two fixed slots per target, no descriptor to read
            c->d.tsum_direct = (!c->d.wide_targets && c->shape.max_target_len >= 1 && c->shape.max_target_len <= 64) ? 1 : 0;
            const size_t n_sum = c->d.tsum_direct ? (size_t)2 * c->d.R * 2 * c->d.T : (size_t)2 * c->d.R * c->d.n_tiles * nt;
            if (c->tsum_buf) (void)hipFree(c->tsum_buf);
            c->tsum_buf = nullptr;
            HIP_OK(hipMalloc((void**)&c->tsum_buf, n_sum * sizeof(double)));
            HIP_OK(hipMemsetAsync(c->tsum_buf, 0, n_sum * sizeof(double), stream));
            if (!c->tdesc_buf) HIP_OK(hipMalloc((void**)&c->tdesc_buf, (size_t)c->d.T * sizeof(int2)));
            c->d.tsum = c->tsum_buf;
            c->d.tdesc = c->tdesc_buf;
            hipLaunchKernelGGL(k_tdesc, dim3((c->d.T + 255) / 256), dim3(256), 0, stream, c->d, c->tdesc_buf);
            HIP_OK(hipGetLastError());
            drop_graph(c);
        }
    }
    if (c->d.family == kMultiMixture) {
        // k_allele's work list: the allele slots that hold an allele, in (a1, g) order.  Setup only: the mask
        // and the CSR row pointers come to the host once (G A bytes + (G (A - 1) + 1) ints).
        const DevArgs& d = c->d;
        const long G = d.G, A = d.A, A1 = d.A - 1;
        std::vector<uint8_t> mask((size_t)(G * A));
        std::vector<int> ptr((size_t)(G * A1 + 1));
        HIP_OK(hipMemcpyAsync(mask.data(), d.amask, mask.size(), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(ptr.data(), d.a2e_ptr, ptr.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<int> live;
        live.reserve((size_t)(G * A1));
        // in the order the tables are laid out (tab_off): guide-contiguous for the register-resident kernels,
        // allele-contiguous for the allele-parallel ones - consecutive k_allele threads store next to each other
        auto consider = [&](long a1, long g) {
            const long slot = g * A1 + a1;
            if (mask[(size_t)(g * A + a1 + 1)] != 0 || ptr[(size_t)slot + 1] > ptr[(size_t)slot]) live.push_back((int)(a1 * G + g));
        };
        if (d.wide_alleles) {
            for (long g = 0; g < G; ++g)
                for (long a1 = 0; a1 < A1; ++a1) consider(a1, g);
        } else {
            for (long a1 = 0; a1 < A1; ++a1)
                for (long g = 0; g < G; ++g) consider(a1, g);
        }
        if (!c->live_slots) HIP_OK(hipMalloc(&c->live_slots, (size_t)(G * A1) * sizeof(int)));
        if (!live.empty())
            HIP_OK(hipMemcpyAsync(c->live_slots, live.data(), live.size() * sizeof(int), hipMemcpyHostToDevice, stream));
        HIP_OK(hipStreamSynchronize(stream));  // `live` goes out of scope
        // the slots left out hold what k_allele would write there: zeros
        {
            const long n_tab = (long)d.B * A1 * G;
            HIP_OK(hipMemsetAsync((void*)d.tabP, 0, (size_t)n_tab * sizeof(double), stream));
            HIP_OK(hipMemsetAsync((void*)d.tabPmu, 0, (size_t)n_tab * sizeof(double), stream));
            HIP_OK(hipMemsetAsync((void*)d.tabPy, 0, (size_t)n_tab * sizeof(double), stream));
            HIP_OK(hipMemsetAsync((void*)d.mu_a, 0, (size_t)(A1 * G) * sizeof(double), stream));
            if (d.sig_a) HIP_OK(hipMemsetAsync((void*)d.sig_a, 0, (size_t)(A1 * G) * sizeof(double), stream));
            drop_graph(c);
        }
        c->d.live_slots = c->live_slots;
        c->d.n_live_slots = (int)live.size();
    }
    {
        // more than 64 KB of dynamic LDS per workgroup (many conditions): the kernel has to be told
        const DevArgs& d = c->d;
        const bool acc = (d.flags & kAcc) != 0;
        size_t lds = 0;
        const void* fn = nullptr;
        if (c->wave_guide && c->wave2) {
            lds = guide_wave2_lds(d.B, d.tile_targets);
            fn = d.family == kMixture ? (acc ? (const void*)k_guide_wave2<kMixture, true> : (const void*)k_guide_wave2<kMixture, false>)
                                      : (const void*)k_guide_wave2<kNormal, false>;
        } else if (c->tiling_rep || c->tiling_wave) {
            const size_t nt = c->tiling_rep ? 64u * c->tiling_rep_w : 64u;
            lds = guide_tiling_lds(d.B, acc, nt, !c->tiling_rep);
            if (c->tiling_rep)
                fn = d.survival ? (acc ? (const void*)k_guide_tiling_rep<true, true> : (const void*)k_guide_tiling_rep<false, true>)
                                : (acc ? (const void*)k_guide_tiling_rep<true, false> : (const void*)k_guide_tiling_rep<false, false>);
            else
                fn = d.survival ? (acc ? (const void*)k_guide_tiling_wave<true, true> : (const void*)k_guide_tiling_wave<false, true>)
                                : (acc ? (const void*)k_guide_tiling_wave<true, false> : (const void*)k_guide_tiling_wave<false, false>);
        }
        if (lds > kLdsPerWorkgroupMax)
            return fail("bean_hip_prepare: " + std::to_string(d.B) + " conditions x " + std::to_string(d.tile_targets) +
                        " targets per 64-guide tile need " + std::to_string(lds) + " bytes of LDS per workgroup; a compute unit has " +
                        std::to_string(kLdsPerWorkgroupMax));
        if (fn && lds > 65536) HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
#ifdef BEAN_AB_KERNELS
    c->tile_ready = false;
    if (c->tile_svi) {
        // tiles of whole targets with at most kTileThreads / R guides: built on the host from the target
        // offsets (4 (T + 1) bytes read back once, next to the 4 bytes above)
        const int T = c->d.T, G = c->d.G, gb = kTileThreads / c->d.R;
        std::vector<int> toff((size_t)T + 1);
        HIP_OK(hipMemcpyAsync(toff.data(), c->d.toff, sizeof(int) * ((size_t)T + 1), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<int> tg0, tt0;
        int ntm = 1, gbm = 1;
        bool ok = toff[0] == 0 && toff[T] == G;
        for (int t = 0; ok && t < T;) {
            int t1 = t, g0 = toff[t];
            while (t1 < T && toff[t1 + 1] - g0 <= gb && toff[t1 + 1] > toff[t1]) ++t1;
            if (t1 == t) {  // an empty target or one longer than a tile: the two-launch path handles it
                ok = false;
                break;
            }
            tg0.push_back(g0);
            tt0.push_back(t);
            if (t1 - t > ntm) ntm = t1 - t;
            if (toff[t1] - g0 > gbm) gbm = toff[t1] - g0;
            t = t1;
        }
        if (ok && !tg0.empty()) {
            tg0.push_back(G);
            tt0.push_back(T);
            const size_t n1 = tg0.size();
            if (c->tile_tab) (void)hipFree(c->tile_tab);
            c->tile_tab = nullptr;
            HIP_OK(hipMalloc((void**)&c->tile_tab, 2 * n1 * sizeof(int)));
            HIP_OK(hipMemcpyAsync(c->tile_tab, tg0.data(), n1 * sizeof(int), hipMemcpyHostToDevice, stream));
            HIP_OK(hipMemcpyAsync(c->tile_tab + n1, tt0.data(), n1 * sizeof(int), hipMemcpyHostToDevice, stream));
            HIP_OK(hipStreamSynchronize(stream));
            c->n_tiles = (int)n1 - 1;
            c->tile_ntm = ntm;
            c->tile_gbm = gbm;
            c->tile_blocks = 0;
            c->tile_ready = true;
        }
    }
#endif
    c->prepared = true;
    return 0;
}

// ------------------------------------------------------------------ launches
static void grid_param(const bean_hip_ctx* c, int& n_target_blocks, int& n_blocks) {
    const DevArgs& d = c->d;
    n_target_blocks = d.wide_targets ? d.T : (int)(((long)d.T * d.lpt + kParamBlock - 1) / kParamBlock);
    int guide_blocks = 0;
    if (d.family == kMultiMixture)  // kAMax lanes per guide, or one wave per guide on the wide path
        guide_blocks = (int)(((long)d.G * (d.wide_alleles ? 64 : kAMax) + kParamBlock - 1) / kParamBlock);
    else if (d.surv_q0lik) guide_blocks = d.n_gamma_blocks;
    else if (d.family == kMixture)  // the alpha_pi blocks + (survival) the q0 blocks
        guide_blocks = (d.G + kParamBlock - 1) / kParamBlock + (d.q0_blocks ? d.n_gamma_blocks : 0);
    n_blocks = n_target_blocks + guide_blocks;
}

// Allele blocks of a PREP launch of k_param on DevArgs d (tgrad as launched): its specialised tiling build (KIND 3), a
// sorting screen, not switched off (BEAN_HIP_ALLELE=split; BEAN_HIP_PARAM_KIND=0 forces the generic kernel, which has none).
static bool param_generic_only() {
    static const bool g = getenv("BEAN_HIP_PARAM_KIND") && !strcmp(getenv("BEAN_HIP_PARAM_KIND"), "0");
    return g;
}
static bool param_kind3(const DevArgs& d) {
    return d.family == kMultiMixture && !d.wide_targets && !d.wide_alleles && !d.n_cov && !d.lpart &&
           d.trow_summed && !d.surv_q0lik && d.lpt == kLanesPerTargetNarrow;
}
static int param_allele_blocks(const bean_hip_ctx* c, const DevArgs& d) {
    if (param_generic_only() || !param_kind3(d) || d.survival || !c->allele_blocks || d.n_live_slots <= 0 || !d.tile_ctr) return 0;
    if (d.T <= 0) return 0;  // (no edit block would ever raise the go flags)
    return (int)(((long)d.n_live_slots + kParamBlock - 1) / kParamBlock);
}
// what a PREP launch leaves (with or without exchanged gradients: the same build of the kernel, the same answer)
static bool param_prep_leaves_alleles_fresh(const bean_hip_ctx* c) {
    return param_allele_blocks(c, c->d) > 0;
}

template <bool FINISH, bool ADAM, bool PREP>
static void launch_param(bean_hip_ctx* c, hipStream_t stream, const double* tgrad = nullptr, bool cov_exchanged = false) {
    int ntb, nb;
    grid_param(c, ntb, nb);
    DevArgs d = c->d;
    d.tgrad = tgrad;
    if (d.n_cov) {  // sample covariates: their own small step runs first (it owns the replicates' shifts)
        // (guide-sharded: the sums were formed by bean_hip_sharded_guide and all-reduced by the caller)
        if (FINISH && !cov_exchanged) hipLaunchKernelGGL(k_cov_sum, dim3(d.R), dim3(1024), 0, stream, d);
        hipLaunchKernelGGL((k_cov_step<FINISH, ADAM, PREP>), dim3(1), dim3(64), 0, stream, d);
    }
    // the specialised build of the kernel (k_param<..., 1>) where its launch conditions hold
    const bool kind1 = !d.survival && d.family != kMultiMixture && !d.wide_targets && !d.tgrad && !d.n_cov && d.wrow &&
                       d.rows_v2 && !d.rrow && !d.surv_q0lik && !d.not_loss_owner && d.lpart &&
                       (d.dgq || d.family != kMixture) && d.tsum;
    const bool kind2 = d.survival && d.family == kMixture && !d.surv_q0lik && d.dgq && !d.tsum && !d.wide_targets && !d.tgrad &&
                       !d.n_cov && d.wrow && d.rows_v2 && !d.rrow && d.lpart && d.lpt == kLanesPerTargetNarrow;
    const bool kind3 = param_kind3(d);
    // BEAN_HIP_PARAM_KIND=0 forces the generic build (the test that the specialised builds change nothing)
    const bool generic_only = param_generic_only();
    const int kind = generic_only ? 0 : (kind1 ? 1 : (kind2 ? 2 : (kind3 ? 3 : 0)));
    const bool prof = c->profile && c->profile_param && FINISH && PREP && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    }
    // allele blocks (k_param<..., 3>, sorting): k_allele's work as the tail of this grid
    const int nab = (PREP && kind == 3) ? param_allele_blocks(c, d) : 0;
    if (PREP) c->alleles_fresh = nab > 0;
    if (nab > 0) {
        // Dispatch order: edit blocks, `ahead` guide blocks, the allele blocks, the other guide blocks (DevArgs::q0_blk0,
        // which the sorting tiling family does not use otherwise).  Measured at BASELINE config 3 (1 792 places for 469 +
        // 1 563 + 753 blocks; us per step): the allele blocks last 146.5; behind the guide blocks that fill the first
        // resident round (1 323) 148.0 (they poll on places the other guide blocks wait for); behind 531 of them 152.2;
        // k_allele as a launch of its own 148.9.  BEAN_HIP_ALLELE_AHEAD: experiments.
        static const int ahead_env = getenv("BEAN_HIP_ALLELE_AHEAD") ? atoi(getenv("BEAN_HIP_ALLELE_AHEAD")) : -1;
        const int ngb = nb - ntb;
        d.q0_blk0 = ahead_env >= 0 && ahead_env < ngb ? ahead_env : ngb;
    }
    const dim3 grid(nb + nab), block(kParamBlock);
#define BEAN_LAUNCH_PARAM(K)                                                                                    \
    do {                                                                                                         \
        if (prof) hipExtLaunchKernelGGL((k_param<FINISH, ADAM, PREP, K>), grid, block, 0, stream, e0, e1, 0, d, ntb); \
        else hipLaunchKernelGGL((k_param<FINISH, ADAM, PREP, K>), grid, block, 0, stream, d, ntb);              \
    } while (0)
    switch (kind) {
        case 1: BEAN_LAUNCH_PARAM(1); break;
        case 2: BEAN_LAUNCH_PARAM(2); break;
        case 3: BEAN_LAUNCH_PARAM(3); break;
        default: BEAN_LAUNCH_PARAM(0); break;
    }
#undef BEAN_LAUNCH_PARAM
}

#ifdef BEAN_AB_KERNELS  // block forms and the split form: A/B references
template <int B>
static void launch_guide_b(bean_hip_ctx* c, hipStream_t stream, dim3 grid, dim3 block, size_t lds) {
    const DevArgs& d = c->d;
    if (d.survival && d.family != kMultiMixture) {
        if (d.family == kMixture) {
            if (d.flags & kAcc)
                hipLaunchKernelGGL((k_guide_survival<B, kMixture, true>), grid, block, lds, stream, d);
            else
                hipLaunchKernelGGL((k_guide_survival<B, kMixture, false>), grid, block, lds, stream, d);
        } else {
            hipLaunchKernelGGL((k_guide_survival<B, kNormal, false>), grid, block, lds, stream, d);
        }
    } else if (d.family == kMultiMixture) {
#if BEAN_AMAX <= 8
        const size_t tl = ((size_t)kTNumPart * 64 + 16) * sizeof(double);
        if (d.survival) {
            if (d.flags & kAcc)
                hipLaunchKernelGGL((k_guide_tiling<B, true, true>), grid, block, tl, stream, d);
            else
                hipLaunchKernelGGL((k_guide_tiling<B, false, true>), grid, block, tl, stream, d);
        } else if (d.flags & kAcc) {
            hipLaunchKernelGGL((k_guide_tiling<B, true, false>), grid, block, tl, stream, d);
        } else {
            hipLaunchKernelGGL((k_guide_tiling<B, false, false>), grid, block, tl, stream, d);
        }
#endif
    }
}

static int waves_per_block(const bean_hip_ctx* c) { return c->d.R < 8 ? c->d.R : 8; }

static void launch_lik(bean_hip_ctx* c, hipStream_t stream, dim3 grid, dim3 block, size_t lds) {
    const DevArgs& d = c->d;
    if (d.family == kMixture) {
        if (d.flags & kAcc)
            hipLaunchKernelGGL((k_lik<true, true>), grid, block, lds, stream, d);
        else
            hipLaunchKernelGGL((k_lik<true, false>), grid, block, lds, stream, d);
    } else {
        hipLaunchKernelGGL((k_lik<false, false>), grid, block, lds, stream, d);
    }
}

// sorting variant families as three launches (see bean_kernels.hpp, "split form")
static void launch_guide_split(bean_hip_ctx* c, hipStream_t stream) {
    const DevArgs& d = c->d;
    const bool mix = d.family == kMixture;
    const int nlik = (d.flags & kUseBc) ? 2 : 1;
    const int rep_waves = d.R < 8 ? d.R : 8;
    const dim3 grid((d.G + 63) / 64);
    if (mix) {
        const long n = (long)d.R * d.G;
        hipLaunchKernelGGL(k_sample_pi, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d);
    }
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, stream);
    }
    launch_lik(c, stream, dim3((d.G + 63) / 64, d.R), dim3(64 * nlik), 0);
    if (prof) {
        (void)hipEventRecord(e1, stream);
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    }
    if (mix) {
        const size_t l3 = ((size_t)rep_waves * 7 * 64 + 16) * sizeof(double);
        hipLaunchKernelGGL(k_pi_terms, grid, dim3(64 * rep_waves), l3, stream, d);
    }
}

#endif  // BEAN_AB_KERNELS

// sorting variant families, one wave per (guide tile, replicate), second form (bean_guide_v2.hpp)
static void launch_guide_wave2(bean_hip_ctx* c, hipStream_t stream) {
    const DevArgs& d = c->d;
    const int tiles = d.n_tiles;
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8) * (unsigned)d.R), block(64);
    // BEAN_HIP_LDS_PAD (bytes; experiments only): a larger LDS request lowers the number of resident waves
    static const size_t lds_pad = getenv("BEAN_HIP_LDS_PAD") ? (size_t)atol(getenv("BEAN_HIP_LDS_PAD")) : 0;
    const size_t lds = guide_wave2_lds(d.B, d.tile_targets) + lds_pad;
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        if (d.family == kMixture) {
            if (d.flags & kAcc)
                hipExtLaunchKernelGGL((k_guide_wave2<kMixture, true>), grid, block, lds, stream, e0, e1, 0, d);
            else
                hipExtLaunchKernelGGL((k_guide_wave2<kMixture, false>), grid, block, lds, stream, e0, e1, 0, d);
        } else {
            hipExtLaunchKernelGGL((k_guide_wave2<kNormal, false>), grid, block, lds, stream, e0, e1, 0, d);
        }
        c->ev.push_back(e0);
        c->ev.push_back(e1);
        return;
    }
    if (d.family == kMixture) {
        if (d.flags & kAcc)
            hipLaunchKernelGGL((k_guide_wave2<kMixture, true>), grid, block, lds, stream, d);
        else
            hipLaunchKernelGGL((k_guide_wave2<kMixture, false>), grid, block, lds, stream, d);
    } else {
        hipLaunchKernelGGL((k_guide_wave2<kNormal, false>), grid, block, lds, stream, d);
    }
}

#ifdef BEAN_AB_KERNELS
// sorting variant families, one wave per (guide tile, replicate)
static void launch_guide_wave(bean_hip_ctx* c, hipStream_t stream) {
    const DevArgs& d = c->d;
    const dim3 grid((d.G + 63) / 64, d.R), block(64);
    const size_t lds = ((size_t)3 * d.B * d.tile_targets + (size_t)kWaveMisc * 64) * sizeof(double) +
                       (size_t)2 * d.B * 64 * sizeof(float);
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        // profile mode: the events carry the kernel's own begin / end timestamps (hipExtLaunchKernelGGL),
        // i.e. the duration rocprofv3 --kernel-trace reports, without the dispatch gap
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        if (d.family == kMixture) {
            if (d.flags & kAcc)
                hipExtLaunchKernelGGL((k_guide_wave<kMixture, true>), grid, block, lds, stream, e0, e1, 0, d);
            else
                hipExtLaunchKernelGGL((k_guide_wave<kMixture, false>), grid, block, lds, stream, e0, e1, 0, d);
        } else {
            hipExtLaunchKernelGGL((k_guide_wave<kNormal, false>), grid, block, lds, stream, e0, e1, 0, d);
        }
        c->ev.push_back(e0);
        c->ev.push_back(e1);
        return;
    }
    if (d.family == kMixture) {
        if (d.flags & kAcc)
            hipLaunchKernelGGL((k_guide_wave<kMixture, true>), grid, block, lds, stream, d);
        else
            hipLaunchKernelGGL((k_guide_wave<kMixture, false>), grid, block, lds, stream, d);
    } else {
        hipLaunchKernelGGL((k_guide_wave<kNormal, false>), grid, block, lds, stream, d);
    }
}

#endif  // BEAN_AB_KERNELS

// tiling families, one wave per (guide tile, replicate), then the sum over replicates
static void launch_guide_tiling_wave(bean_hip_ctx* c, hipStream_t stream) {
    const DevArgs& d = c->d;
    const bool acc = (d.flags & kAcc) != 0;
    const dim3 grid((unsigned)(((d.G + 63) / 64 + 7) / 8 * 8) * (unsigned)d.R), block(64);
    const size_t lds = guide_tiling_lds(d.B, acc, 64, true);
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    if (prof) {  // events with the kernel's own timestamps, as in launch_guide_wave
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        if (d.survival) {
            if (acc) hipExtLaunchKernelGGL((k_guide_tiling_wave<true, true>), grid, block, lds, stream, e0, e1, 0, d);
            else hipExtLaunchKernelGGL((k_guide_tiling_wave<false, true>), grid, block, lds, stream, e0, e1, 0, d);
        } else {
            if (acc) hipExtLaunchKernelGGL((k_guide_tiling_wave<true, false>), grid, block, lds, stream, e0, e1, 0, d);
            else hipExtLaunchKernelGGL((k_guide_tiling_wave<false, false>), grid, block, lds, stream, e0, e1, 0, d);
        }
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    } else if (d.survival) {
        if (acc) hipLaunchKernelGGL((k_guide_tiling_wave<true, true>), grid, block, lds, stream, d);
        else hipLaunchKernelGGL((k_guide_tiling_wave<false, true>), grid, block, lds, stream, d);
    } else {
        if (acc) hipLaunchKernelGGL((k_guide_tiling_wave<true, false>), grid, block, lds, stream, d);
        else hipLaunchKernelGGL((k_guide_tiling_wave<false, false>), grid, block, lds, stream, d);
    }
    hipLaunchKernelGGL(k_sum_trow, dim3((d.G + 255) / 256, kTNumPart), dim3(256), 0, stream, d);
}

// tiling families, the replicates of a guide in one wave (bean_tiling_v2.hpp): rows go straight to `part`
static void launch_guide_tiling_rep(bean_hip_ctx* c, hipStream_t stream) {
    const DevArgs& d = c->d;
    const bool acc = (d.flags & kAcc) != 0;
    const int waves = c->tiling_rep_w;
    const int nt = 64 * waves;
    int gw = nt / d.R;  // guides per workgroup (R <= kTilingRepMaxR = 64, so >= 1)
    unsigned n_wg = (unsigned)((d.G + gw - 1) / gw);
    // which guides a workgroup takes: tiling_rep_slice, mode 1 (an XCD's workgroups take contiguous runs of the guide order,
    // one run per quarter of it; BEAN_HIP_TILING_MAP=0: workgroup b takes slice b).  Config 3, same bits: kernel 115.2 ->
    // 113.1 us, step 145.5 -> 144.8.  The grid is padded to a multiple of 32, the mode rides above the argument's low byte.
    static const int map_mode = getenv("BEAN_HIP_TILING_MAP") ? atoi(getenv("BEAN_HIP_TILING_MAP")) : 1;
    if (map_mode) {
        n_wg = (n_wg + 31) / 32 * 32;
        gw |= map_mode << 8;
    }
    // issue priority by progress (k_guide_tiling_rep's phase_prio): bit 16
    static const int prio_mode = getenv("BEAN_HIP_TILING_PRIO") ? atoi(getenv("BEAN_HIP_TILING_PRIO")) : 1;
    if (prio_mode) gw |= 1 << 16;
    const dim3 grid(n_wg), block(nt);
    const size_t lds = guide_tiling_lds(d.B, acc, (size_t)nt, false);
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {  // events with the kernel's own timestamps
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    }
#define BEAN_LAUNCH_TREP(ACC_, SURV_)                                                                              \
    do {                                                                                                          \
        if (prof) hipExtLaunchKernelGGL((k_guide_tiling_rep<ACC_, SURV_>), grid, block, lds, stream, e0, e1, 0, d, gw); \
        else hipLaunchKernelGGL((k_guide_tiling_rep<ACC_, SURV_>), grid, block, lds, stream, d, gw);              \
    } while (0)
    if (d.survival) {
        if (acc) BEAN_LAUNCH_TREP(true, true);
        else BEAN_LAUNCH_TREP(false, true);
    } else {
        if (acc) BEAN_LAUNCH_TREP(true, false);
        else BEAN_LAUNCH_TREP(false, false);
    }
#undef BEAN_LAUNCH_TREP
}

#ifdef BEAN_AB_KERNELS
// survival variant families, one wave per (guide tile, replicate) (bean_survival_v2.hpp)
// One SVI step in one launch (bean_step_v2.hpp); `flip` alternates the step-counter buffers.
static void launch_step_wave2(bean_hip_ctx* c, hipStream_t stream, int flip) {
    const DevArgs& d = c->d;
    const int tiles = d.n_tiles;
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8) * (unsigned)d.R), block(64);
    const size_t lds = guide_wave2_lds(d.B, d.tile_targets);
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        if (d.family == kMixture) {
            if (d.flags & kAcc)
                hipExtLaunchKernelGGL((k_step_wave2<kMixture, true>), grid, block, lds, stream, e0, e1, 0, d, flip);
            else
                hipExtLaunchKernelGGL((k_step_wave2<kMixture, false>), grid, block, lds, stream, e0, e1, 0, d, flip);
        } else {
            hipExtLaunchKernelGGL((k_step_wave2<kNormal, false>), grid, block, lds, stream, e0, e1, 0, d, flip);
        }
        c->ev.push_back(e0);
        c->ev.push_back(e1);
        return;
    }
    if (d.family == kMixture) {
        if (d.flags & kAcc)
            hipLaunchKernelGGL((k_step_wave2<kMixture, true>), grid, block, lds, stream, d, flip);
        else
            hipLaunchKernelGGL((k_step_wave2<kMixture, false>), grid, block, lds, stream, d, flip);
    } else {
        hipLaunchKernelGGL((k_step_wave2<kNormal, false>), grid, block, lds, stream, d, flip);
    }
}

#endif  // BEAN_AB_KERNELS

// survival variant families, one wave per (guide tile, replicate) (bean_survival_v2.hpp)
static void launch_guide_survival_wave(bean_hip_ctx* c, hipStream_t stream) {
    const DevArgs& d = c->d;
    const int tiles = (d.G + 63) / 64;
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8) * (unsigned)d.R), block(64);
    const size_t lds = guide_survival_wave_lds(d.B);
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        if (d.family == kMixture) {
            if (d.flags & kAcc)
                hipExtLaunchKernelGGL((k_guide_survival_wave<kMixture, true>), grid, block, lds, stream, e0, e1, 0, d);
            else
                hipExtLaunchKernelGGL((k_guide_survival_wave<kMixture, false>), grid, block, lds, stream, e0, e1, 0, d);
        } else {
            hipExtLaunchKernelGGL((k_guide_survival_wave<kNormal, false>), grid, block, lds, stream, e0, e1, 0, d);
        }
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    } else if (d.family == kMixture) {
        if (d.flags & kAcc)
            hipLaunchKernelGGL((k_guide_survival_wave<kMixture, true>), grid, block, lds, stream, d);
        else
            hipLaunchKernelGGL((k_guide_survival_wave<kMixture, false>), grid, block, lds, stream, d);
    } else {
        hipLaunchKernelGGL((k_guide_survival_wave<kNormal, false>), grid, block, lds, stream, d);
    }
    if (d.surv_q0lik)  // projection term of the G-dimensional Dirichlet's pathwise gradient
        hipLaunchKernelGGL(k_sum_q, dim3(d.R), dim3(1024), 0, stream, d);
}

static void launch_guide(bean_hip_ctx* c, hipStream_t stream) {
    const DevArgs& d = c->d;
#ifdef BEAN_AB_KERNELS
    if (!c->fused_guide && !d.survival && d.family != kMultiMixture) {
        launch_guide_split(c, stream);
        return;
    }
#endif
    if (c->wave_guide) {
#ifdef BEAN_AB_KERNELS
        if (!c->wave2) {
            launch_guide_wave(c, stream);
            return;
        }
#endif
        launch_guide_wave2(c, stream);
        return;
    }
    if (c->surv_wave) {
        launch_guide_survival_wave(c, stream);
        return;
    }
    if (d.family == kMultiMixture) {  // allele-level tables of this step's draw, unless k_param's allele blocks left them
        const long n = d.n_live_slots;
        if (n > 0 && !c->alleles_fresh) hipLaunchKernelGGL(k_allele, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d);
        c->alleles_fresh = true;
    }
    if (c->tiling_rep) {
        launch_guide_tiling_rep(c, stream);
        return;
    }
    if (c->tiling_wave) {
        launch_guide_tiling_wave(c, stream);
        return;
    }
    if (c->tiling_wide) {
        const bool acc = (d.flags & kAcc) != 0;
        const dim3 gridw((unsigned)(((long)d.G + 7) / 8 * 8 * d.R)), blockw(64);
        const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (prof) {
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0, stream);
        }
        if (d.survival) {
            if (acc) hipLaunchKernelGGL((k_guide_tiling_wide<true, true>), gridw, blockw, 0, stream, d);
            else hipLaunchKernelGGL((k_guide_tiling_wide<false, true>), gridw, blockw, 0, stream, d);
        } else {
            if (acc) hipLaunchKernelGGL((k_guide_tiling_wide<true, false>), gridw, blockw, 0, stream, d);
            else hipLaunchKernelGGL((k_guide_tiling_wide<false, false>), gridw, blockw, 0, stream, d);
        }
        if (prof) {
            (void)hipEventRecord(e1, stream);
            c->ev.push_back(e0);
            c->ev.push_back(e1);
        }
        return;
    }
#ifdef BEAN_AB_KERNELS  // block forms
    const int nw = waves_per_block(c);
    const dim3 grid((d.G + 63) / 64), block(64 * nw);
    const size_t lds = ((size_t)nw * kNumPart * 64 + 16) * sizeof(double);
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (prof) {
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, stream);
    }
    switch (d.B) {
        case 1: launch_guide_b<1>(c, stream, grid, block, lds); break;
        case 2: launch_guide_b<2>(c, stream, grid, block, lds); break;
        case 3: launch_guide_b<3>(c, stream, grid, block, lds); break;
        case 4: launch_guide_b<4>(c, stream, grid, block, lds); break;
        case 5: launch_guide_b<5>(c, stream, grid, block, lds); break;
        case 6: launch_guide_b<6>(c, stream, grid, block, lds); break;
        case 7: launch_guide_b<7>(c, stream, grid, block, lds); break;
#if BEAN_BMAX <= 8
        default: launch_guide_b<8>(c, stream, grid, block, lds); break;
#else
        case 8: launch_guide_b<8>(c, stream, grid, block, lds); break;
        case 9: launch_guide_b<9>(c, stream, grid, block, lds); break;
        case 10: launch_guide_b<10>(c, stream, grid, block, lds); break;
        case 11: launch_guide_b<11>(c, stream, grid, block, lds); break;
        case 12: launch_guide_b<12>(c, stream, grid, block, lds); break;
        case 13: launch_guide_b<13>(c, stream, grid, block, lds); break;
        case 14: launch_guide_b<14>(c, stream, grid, block, lds); break;
        case 15: launch_guide_b<15>(c, stream, grid, block, lds); break;
        default: launch_guide_b<16>(c, stream, grid, block, lds); break;
#endif
    }
    if (d.surv_q0lik)  // projection term of the G-dimensional Dirichlet's pathwise gradient
        hipLaunchKernelGGL(k_sum_q, dim3(d.R), dim3(1024), 0, stream, d);
    if (prof) {
        (void)hipEventRecord(e1, stream);
        c->ev.push_back(e0);
        c->ev.push_back(e1);
    }
#endif
}

static void launch_finalize(bean_hip_ctx* c, hipStream_t stream, uint64_t first, uint64_t n, bool cur) {
    const unsigned blocks = (unsigned)((n + 3) / 4);  // one wave per slot
    hipLaunchKernelGGL(k_loss_finalize, dim3(blocks), dim3(n == 1 ? 64 : 256), 0, stream, c->d, (unsigned long long)first,
                       (unsigned long long)n, cur ? 1 : 0);
}

// step counters to (step, slot) and the loss accumulators of n slots from `slot` to zero
static void launch_set_step(bean_hip_ctx* c, hipStream_t stream, uint64_t step, uint64_t slot, uint64_t n) {
    const uint64_t words = n * kLossSub * kLossWords;
    unsigned blocks = (unsigned)((words + 255) / 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_set_step, dim3(blocks), dim3(256), 0, stream, c->d, (unsigned long long)step,
                       (unsigned long long)slot, (unsigned long long)n);
}

extern "C" int bean_hip_elbo_grad(bean_hip_ctx* c, uint64_t seed, uint64_t step, uint64_t loss_index,
                                  void* stream_) {
    if (!c) return fail("bean_hip_elbo_grad: null handle");
    if (!c->prepared) return fail("bean_hip_elbo_grad: call bean_hip_prepare first");
    if (check_bound(c, true, false)) return -1;
    if (loss_index >= c->loss_capacity) return fail("bean_hip_elbo_grad: loss_index beyond loss_hist");
    hipStream_t stream = (hipStream_t)stream_;
    c->d.seed = seed;
    c->resume_ok = false;
    launch_set_step(c, stream, step, loss_index, 1);
    launch_param<false, false, true>(c, stream);
    launch_guide(c, stream);
    launch_param<true, false, false>(c, stream);
    launch_finalize(c, stream, loss_index, 1, false);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bean_hip_adam(bean_hip_ctx* c, uint64_t t, void* stream_) {
    if (!c) return fail("bean_hip_adam: null handle");
    if (check_bound(c, true, true)) return -1;
    if (t < 1) return fail("bean_hip_adam: t is 1-based");
    c->resume_ok = false;
    hipStream_t stream = (hipStream_t)stream_;
    for (int i = 0; i < 8; ++i) {
        const uint64_t bytes = expected_bytes(c->shape, BEAN_BUF_P_MU_LOC + i);
        if (!bytes) continue;
        const long n = (long)(bytes / 4);
        hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, c->d.p[i],
                           (const float*)c->d.g[i], c->d.m[i], c->d.v[i], n, c->d, (unsigned long long)t);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

static void enqueue_pairs(bean_hip_ctx* c, hipStream_t stream, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        launch_param<true, true, true>(c, stream);
        launch_guide(c, stream);
    }
}

// Capture 2^k {k_param, guide} pairs into an executable graph.  On any failure the stream is taken
// out of capture mode before returning.
static void enqueue_fused(bean_hip_ctx* c, hipStream_t stream, uint64_t n, int flip0 = 0) {
#ifdef BEAN_AB_KERNELS
    for (uint64_t i = 0; i < n; ++i) launch_step_wave2(c, stream, (int)((i + flip0) & 1));
#else
    (void)c; (void)stream; (void)n; (void)flip0;
#endif
}

static int capture_pairs(bean_hip_ctx* c, hipStream_t stream, uint64_t n, hipGraphExec_t* out, bool fused = false) {
    hipGraph_t graph = nullptr;
    HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    const bool fresh_was = c->alleles_fresh;  // (a capture runs nothing; these graphs begin with a PREP launch)
    if (fused) enqueue_fused(c, stream, n);
    else enqueue_pairs(c, stream, n);
    c->alleles_fresh = fresh_was;
    hipError_t e = hipStreamEndCapture(stream, &graph);
    if (e != hipSuccess) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            hipGraph_t junk = nullptr;
            (void)hipStreamEndCapture(stream, &junk);
            if (junk) (void)hipGraphDestroy(junk);
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    }
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
    return 0;
}

#ifdef BEAN_AB_KERNELS

// ---- one launch per call: bean_tile_svi.hpp
template <int FAM, bool ACC>
static int launch_svi_tile_t(bean_hip_ctx* c, hipStream_t stream, uint64_t step0, uint64_t slot0, uint64_t n_steps,
                             bool prep_last) {
    const DevArgs& d = c->d;
    const size_t lds = svi_tile_lds(d.B, d.R, c->tile_ntm, c->tile_gbm);
    if (c->tile_blocks == 0) {
        int per_cu = 0, dev = 0, n_cu = 0;
        HIP_OK(hipFuncSetAttribute((const void*)k_svi_tile<FAM, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_svi_tile<FAM, ACC>, kTileThreads, lds));
        HIP_OK(hipGetDevice(&dev));
        HIP_OK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        if (per_cu < 1) return fail("k_svi_tile does not fit on a compute unit (LDS " + std::to_string(lds) + " bytes)");
        c->tile_blocks = per_cu * n_cu;
        if (getenv("BEAN_HIP_VERBOSE"))
            fprintf(stderr, "k_svi_tile: %d tiles (<= %d targets, %d guides), LDS %zu B, %d workgroups per CU x %d CUs\n",
                    c->n_tiles, c->tile_ntm, c->tile_gbm, lds, per_cu, n_cu);
    }
    const int blocks = c->n_tiles < c->tile_blocks ? c->n_tiles : c->tile_blocks;
    const int* tg0 = c->tile_tab;
    const int* tt0 = c->tile_tab + (c->n_tiles + 1);
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    if (prof) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipExtLaunchKernelGGL((k_svi_tile<FAM, ACC>), dim3(blocks), dim3(kTileThreads), lds, stream, e0, e1, 0, d, (const DevArgs*)c->dargs_dev,
                              tg0, tt0,
                              c->n_tiles, c->tile_ntm, c->tile_gbm, (unsigned long long)step0, (unsigned long long)slot0,
                              (int)n_steps, prep_last ? 1 : 0, (const float*)c->step_sizes);
        c->ev.push_back(e0);
        c->ev.push_back(e1);
        c->ev_steps.push_back(n_steps);
    } else {
        hipLaunchKernelGGL((k_svi_tile<FAM, ACC>), dim3(blocks), dim3(kTileThreads), lds, stream, d, (const DevArgs*)c->dargs_dev, tg0, tt0,
                           c->n_tiles,
                           c->tile_ntm, c->tile_gbm, (unsigned long long)step0, (unsigned long long)slot0, (int)n_steps,
                           prep_last ? 1 : 0, (const float*)c->step_sizes);
    }
    return 0;
}

static int launch_svi_tile(bean_hip_ctx* c, hipStream_t stream, uint64_t step0, uint64_t n_steps) {
    const DevArgs& d = c->d;
    if (n_steps > c->step_sizes_cap) {
        if (c->step_sizes) (void)hipFree(c->step_sizes);
        c->step_sizes = nullptr;
        c->step_sizes_cap = 0;
        const uint64_t cap = n_steps < 256 ? 256 : n_steps;
        HIP_OK(hipMalloc((void**)&c->step_sizes, cap * sizeof(float)));
        c->step_sizes_cap = cap;
    }
    hipLaunchKernelGGL(k_step_sizes, dim3((unsigned)((n_steps + 255) / 256)), dim3(256), 0, stream, d,
                       (unsigned long long)step0, (int)n_steps, c->step_sizes);
    if (!c->dargs_dev) HIP_OK(hipMalloc((void**)&c->dargs_dev, sizeof(DevArgs)));
    // (the kernel-argument copy and this one are the same bytes: c->d as it is now)
    hipLaunchKernelGGL(k_put_args, dim3(1), dim3(64), 0, stream, d, c->dargs_dev);
    if (d.family == kMixture) {
        if (d.flags & kAcc) return launch_svi_tile_t<kMixture, true>(c, stream, step0, step0, n_steps, false);
        return launch_svi_tile_t<kMixture, false>(c, stream, step0, step0, n_steps, false);
    }
    return launch_svi_tile_t<kNormal, false>(c, stream, step0, step0, n_steps, false);
}

#endif  // BEAN_AB_KERNELS

// ---- all the steps of a call in one launch: bean_async_v2.hpp
// (n_steps: the call's; a group's queue counter is an int)
static bool async_candidate(const bean_hip_ctx* c, uint64_t n_steps) {
    const DevArgs& d = c->d;
    const uint64_t per_group = (uint64_t)((d.n_tiles + 7) / 8) * (uint64_t)d.R;
    if (n_steps == 0 || per_group * n_steps > 2000000000ull) return false;
    return c->async_step && !c->profile_param && !d.eps_mu_in && !d.eps_sd_in && !d.pi_in && !d.eps_noise_in &&
           !d.eps_mu_out && !d.eps_sd_out && !d.eps_noise_out && !(d.flags & kDumpPi) && d.lpart && d.tile_ctr &&
           (d.family != kMixture || d.dgq);
}

template <int FAM, bool ACC>
static int launch_svi_async_t(bean_hip_ctx* c, hipStream_t stream, const AsyncArgs& a_in, int* roles_out) {
    const DevArgs& d = c->d;
    const size_t lds = guide_wave2_lds(d.B, d.tile_targets);
    if (c->async_blocks == 0) {
        // resident single-wave workgroups: the grid (nothing depends on residency but speed: a block that is
        // dispatched late finds what is left of its group's queue)
        int per_cu = 0, dev = 0, n_cu = 0;
        // (more than 64 KB of dynamic LDS per workgroup - many conditions - has to be asked for)
        if (lds > 65536)
            HIP_OK(hipFuncSetAttribute((const void*)k_svi_async<FAM, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_svi_async<FAM, ACC>, 64, lds));
        HIP_OK(hipGetDevice(&dev));
        HIP_OK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        if (per_cu < 1) return fail("k_svi_async does not fit on a compute unit (LDS " + std::to_string(lds) + " bytes)");
        // the same number of waves on every SIMD (four per CU): async_waves_per_simd, within what fits
        int k = async_waves_per_simd((long)d.n_tiles * d.R, 4l * n_cu);
        if (k > per_cu / 4) k = per_cu / 4;
        c->async_blocks = k >= 1 ? k * 4 * n_cu : per_cu * n_cu / 8 * 8;
        // finisher roles: one more wave per SIMD that only finishes tiles, where there is room for it and the item waves
        // have more than ~2 items per step each (async_finisher_roles)
        c->async_fin_blocks = (k >= 1 && k + 1 <= per_cu / 4 && async_finisher_roles(k, (long)d.n_tiles * d.R, 4l * n_cu)) ? 4 * n_cu : 0;
        if (const char* e = getenv("BEAN_HIP_ASYNC_BLOCKS")) {  // experiments
            const int v = atoi(e) / 8 * 8;
            if (v >= 8) c->async_blocks = v;
        }
        if (const char* e = getenv("BEAN_HIP_ASYNC_FIN")) {  // experiments: finisher blocks (0: no roles; -1: roles without finishers)
            const int v = atoi(e);
            c->async_fin_blocks = v < 0 ? -1 : v / 8 * 8;
        }
        if (getenv("BEAN_HIP_VERBOSE"))
            fprintf(stderr, "k_svi_async: %d tiles x %d replicates per step, LDS %zu B, %d workgroups per CU x %d CUs -> grid %d\n",
                    d.n_tiles, d.R, lds, per_cu, n_cu, c->async_blocks);
    }
    // no more waves than one step has items (a small screen would only add pollers)
    const long items = (long)((d.n_tiles + 7) / 8 * 8) * d.R;
    int blocks = c->async_blocks;
    if ((long)blocks > items) blocks = (int)((items + 7) / 8 * 8);
    AsyncArgs a = a_in;
    int fin_blocks = 0;
    // the finish as two ring entries (targets / guides on two finishers at once) while the item waves have few items per
    // step - then a tile's chain is what a step waits for - and as one entry beyond (same box, split / one entry: 50k guides
    // 48.5 / 51.3 us per step, 56k 53.2 / 53.3, 62.5k 59.4 / 58.8, 68.75k 65.5 / 64.4, 125k 97.0 / 95.7)
    a.fin_split = (double)d.n_tiles * d.R / (double)(c->async_blocks > 0 ? c->async_blocks : 1) < 2.15 ? 1 : 0;
    if (const char* e = getenv("BEAN_HIP_ASYNC_SPLIT")) a.fin_split = atoi(e) != 0;  // experiments
    if (c->async_fin_blocks != 0 && a.fring_stride > 0 && d.n_tiles <= 8 * 65535) {
        fin_blocks = c->async_fin_blocks < 0 ? 0 : c->async_fin_blocks;
        const int tiles8 = (d.n_tiles + 7) / 8 * 8;
        if (fin_blocks > tiles8) fin_blocks = tiles8;  // (no more finishers than tiles)
        a.n_guide_blocks = blocks;
        blocks += fin_blocks;
    } else {
        a.n_guide_blocks = 0;
    }
    const bool prof = c->profile && !c->profile_param && c->ev.size() < 8192;
    if (prof) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipExtLaunchKernelGGL((k_svi_async<FAM, ACC>), dim3(blocks), dim3(64), lds, stream, e0, e1, 0,
                              (const DevArgs*)c->dargs_dev, d.R, d.n_tiles, a);
        c->ev.push_back(e0);
        c->ev.push_back(e1);
        c->ev_steps.push_back((uint64_t)a.n_steps);
    } else {
        hipLaunchKernelGGL((k_svi_async<FAM, ACC>), dim3(blocks), dim3(64), lds, stream, (const DevArgs*)c->dargs_dev, d.R,
                           d.n_tiles, a);
    }
    if (roles_out) *roles_out = a.n_guide_blocks;
    return 0;
}

// n steps from step0 (loss slots from slot0) on the draw and tables already on the device; leaves the draw and tables
// of step0 + n, the step counters as n {guide, k_param} pairs would, and the loss accumulators filled (not finalized)
static int launch_svi_async(bean_hip_ctx* c, hipStream_t stream, uint64_t step0, uint64_t slot0, uint64_t n_steps) {
    const DevArgs& d = c->d;
    if (n_steps > c->step_sizes_cap) {
        if (c->step_sizes) (void)hipFree(c->step_sizes);
        c->step_sizes = nullptr;
        c->step_sizes_cap = 0;
        const uint64_t cap = n_steps < 256 ? 256 : n_steps;
        HIP_OK(hipMalloc((void**)&c->step_sizes, cap * sizeof(float)));
        c->step_sizes_cap = cap;
    }
    // [8 x stride] queue | abort word (a line of its own) | [8 x stride] finish-ring heads | [8 x stride] tails | done[n_tiles]:
    // zeroed by every call; behind them the finish rings (n_steps x the group's tiles, zeroed when roles are on)
    const size_t ws_ints = (size_t)25 * kAsyncQueueStride + (size_t)2 * d.n_tiles;
    // (finish rings only for calls short enough to have finisher roles: a ring entry holds the step in 14 bits)
    const size_t fring_stride = n_steps < 8000 ? (size_t)2 * ((d.n_tiles + 7) / 8) * n_steps : 0;
    const size_t need_ints = ws_ints + 8 * fring_stride;
    if (need_ints > c->async_ws_ints) {
        if (c->async_ws) (void)hipFree(c->async_ws);
        c->async_ws = nullptr;
        c->async_ws_ints = 0;
        HIP_OK(hipMalloc((void**)&c->async_ws, need_ints * sizeof(int)));
        c->async_ws_ints = need_ints;
    }
    // the out-of-line pieces of the kernel read DevArgs from a copy in global memory (the same bytes: c->d as it is now)
    if (!c->dargs_dev) HIP_OK(hipMalloc((void**)&c->dargs_dev, sizeof(DevArgs)));
    AsyncArgs a;
    a.step0 = step0;
    a.slot0 = slot0;
    a.n_steps = (int)n_steps;
    a.queue = c->async_ws;
    a.abort_flag = c->async_ws + 8 * kAsyncQueueStride;
    a.fhead = c->async_ws + 9 * kAsyncQueueStride;
    a.ftail = c->async_ws + 17 * kAsyncQueueStride;
    a.done = c->async_ws + 25 * kAsyncQueueStride;
    a.fring = c->async_ws + ws_ints;
    a.fring_stride = (long)fring_stride;
    a.n_guide_blocks = 0;  // (set by launch_svi_async_t, where the grid is decided)
    a.fin_split = 0;
    a.step_sizes = c->step_sizes;
    a.stamps = nullptr;
    // queue, abort and completed-step words to zero, step sizes, the DevArgs copy, the step counters the call leaves
    {
        unsigned hb = (unsigned)((8 * fring_stride + 256 * 64 - 1) / (256 * 64));  // ~64 ring words per thread
        if (hb < 1) hb = 1;
        if (hb > 128) hb = 128;
        hipLaunchKernelGGL(k_async_head, dim3(hb), dim3(256), 0, stream, d, a, c->dargs_dev, c->async_ws, (int)ws_ints, c->step_sizes);
    }
#ifdef BEAN_ASYNC_STAMP
    {
        static unsigned long long* g_stamps = nullptr;
        const size_t words = (size_t)2 * kAsyncStampFinOff;  // item rows, then finish-phase rows
        if ((size_t)kAsyncStampSteps * ((d.n_tiles + 7) / 8 * 8) * d.R * 8 > (size_t)kAsyncStampFinOff)
            return fail("BEAN_ASYNC_STAMP: screen too large for the stamp buffer");
        if (!g_stamps) HIP_OK(hipMalloc((void**)&g_stamps, words * 8));
        HIP_OK(hipMemsetAsync(g_stamps, 0, words * 8, stream));
        a.stamps = g_stamps;
        c->async_stamps = g_stamps;
        c->async_stamp_words = words;
    }
#endif
    int rc;
    if (d.family == kMixture) {
        if (d.flags & kAcc) rc = launch_svi_async_t<kMixture, true>(c, stream, a, nullptr);
        else rc = launch_svi_async_t<kMixture, false>(c, stream, a, nullptr);
    } else {
        rc = launch_svi_async_t<kNormal, false>(c, stream, a, nullptr);
    }
    return rc;
}

extern "C" int bean_hip_svi_run(bean_hip_ctx* c, uint64_t seed, uint64_t first_step, uint64_t n_steps,
                                int32_t graph_chunk, void* stream_) {
    if (!c) return fail("bean_hip_svi_run: null handle");
    if (!c->prepared) return fail("bean_hip_svi_run: call bean_hip_prepare first");
    if (check_bound(c, false, true)) return -1;
    if (n_steps == 0) return 0;
    if (first_step + n_steps > c->loss_capacity)
        return fail("bean_hip_svi_run: loss_hist too small for first_step + n_steps");
    hipStream_t stream = (hipStream_t)stream_;
    drop_graph_on_seed_change(c, seed);
    c->d.seed = seed;
    c->resume_ok = false;
    const bool tile_candidate = c->tile_svi && c->tile_ready && !c->profile_param && !c->d.eps_mu_in && !c->d.eps_sd_in &&
                                !c->d.pi_in && !c->d.eps_noise_in && !c->d.eps_mu_out && !c->d.eps_sd_out &&
                                !c->d.eps_noise_out && !(c->d.flags & kDumpPi);
    const bool use_graph = graph_chunk > 0 && stream != nullptr && !c->profile && !tile_candidate;
    // one launch per step unless per-step noise is injected or dumped, or k_param itself is being timed
    const DevArgs& dd = c->d;
    const bool fused = c->fused_step && !c->profile_param && !dd.eps_mu_in && !dd.eps_sd_in && !dd.pi_in &&
                       !dd.eps_noise_in && !dd.eps_mu_out && !dd.eps_sd_out && !dd.eps_noise_out;
    if (use_graph) {
        // Graphs of 1, 2, 4, ... <= graph_chunk pairs, all instantiated at the first call (nothing is
        // instantiated inside a later, possibly timed, call); any number of pairs is then replayed
        // as a sum of powers of two.  The step counters live on the device, so the graphs do not
        // depend on the step.
        int kmax = 0;
        while ((2ull << kmax) <= (uint64_t)graph_chunk && kmax < 10) ++kmax;
        if (fused) {
            // graphs of 2, 4, ... <= graph_chunk launches
            const int n_graphs = kmax > 0 ? kmax : 1;
            if ((int)c->graphs_fused.size() != n_graphs) {
                drop_graph(c);
                for (int k = 0; k < n_graphs; ++k) {
                    hipGraphExec_t ge = nullptr;
                    if (capture_pairs(c, stream, 2ull << k, &ge, true)) {
                        drop_graph(c);
                        return -1;
                    }
                    c->graphs_fused.push_back(ge);
                }
                c->graph_seed = seed;
            }
        } else if ((int)c->graphs.size() != kmax + 1) {
            drop_graph(c);
            for (int k = 0; k <= kmax; ++k) {
                hipGraphExec_t ge = nullptr;
                if (capture_pairs(c, stream, 1ull << k, &ge)) {
                    drop_graph(c);
                    return -1;
                }
                c->graphs.push_back(ge);
            }
            c->graph_seed = seed;
        }
    }
    // one launch for all the steps of the call (bean_tile_svi.hpp) unless per-step noise is injected or
    // dumped, k_param is being timed, or the shape is not eligible
    const bool tile = c->tile_svi && c->tile_ready && !c->profile_param && !dd.eps_mu_in && !dd.eps_sd_in && !dd.pi_in &&
                      !dd.eps_noise_in && !dd.eps_mu_out && !dd.eps_sd_out && !dd.eps_noise_out && !(dd.flags & kDumpPi);
#ifdef BEAN_AB_KERNELS
    if (tile) {
        launch_set_step(c, stream, first_step, first_step, n_steps);
        launch_param<false, false, true>(c, stream);  // draws and tables of the first step
        if (launch_svi_tile(c, stream, first_step, n_steps)) return -1;
        launch_finalize(c, stream, first_step, n_steps, false);
        HIP_OK(hipGetLastError());
        return 0;
    }
#else
    (void)tile;
#endif
    if (async_candidate(c, n_steps) && stream != nullptr) {
        // one launch for the call: it also draws step first_step + n_steps, which nobody reads
        launch_set_step(c, stream, first_step, first_step, n_steps);
        launch_param<false, false, true>(c, stream);  // draws and tables of the first step
        if (launch_svi_async(c, stream, first_step, first_step, n_steps)) return -1;
        launch_finalize(c, stream, first_step, n_steps, false);
        HIP_OK(hipGetLastError());
        return 0;
    }
    launch_set_step(c, stream, first_step, first_step, n_steps);
    launch_param<false, false, true>(c, stream);
    if (fused) {
        // n launches of k_step_wave2 = {guide work, FINISH, PREP of the next step}; graphs hold even
        // numbers of launches (the step counters ping-pong), an odd remainder is launched last
        uint64_t left = n_steps;
        if (use_graph) {
            for (int k = (int)c->graphs_fused.size() - 1; k >= 0; --k)
                while (left >= (2ull << k)) {
                    HIP_OK(hipGraphLaunch(c->graphs_fused[k], stream));
                    left -= 2ull << k;
                }
        }
        enqueue_fused(c, stream, left);
    } else {
        launch_guide(c, stream);
        uint64_t pairs = n_steps - 1;
        if (use_graph) {
            // smallest graphs first: the device works on them while the host submits the larger ones
            // (a launch of the 64-pair graph costs the host more than the step or two already enqueued)
            const int kmax = (int)c->graphs.size() - 1;
            uint64_t big = pairs >> kmax;           // launches of the largest graph
            uint64_t rest = pairs - (big << kmax);  // < 2^kmax: one launch per set bit, ascending
            // (a graph launch costs ~8.5 us of idle device time at its boundary - measured on the kernel
            // timeline of a 20-step call - so one or two pairs are launched directly: six eager launches
            // run back to back)
            for (int k = 0; k < kmax; ++k)
                if (rest & (1ull << k)) {
                    if (k < 2) enqueue_pairs(c, stream, 1ull << k);
                    else {
                        HIP_OK(hipGraphLaunch(c->graphs[k], stream));
                        c->alleles_fresh = true;  // (a graph of pairs ends with a guide launch)
                    }
                }
            for (uint64_t i = 0; i < big; ++i) {
                HIP_OK(hipGraphLaunch(c->graphs[kmax], stream));
                c->alleles_fresh = true;
            }
            pairs = 0;
        }
        enqueue_pairs(c, stream, pairs);
        launch_param<true, true, false>(c, stream);
    }
    launch_finalize(c, stream, first_step, n_steps, false);
    HIP_OK(hipGetLastError());
    return 0;
}

// ---- bean_hip_svi_resume: the loop of a fit that is stepped in windows (run_inference: 100 steps at a time)
// n steps and the loss of those n slots (finalized from the device step counter: the group is self-contained,
// so a captured one closes itself and no launch has to follow the last graph of a call)
static void enqueue_resume_pairs(bean_hip_ctx* c, hipStream_t stream, uint64_t n) {
    if (n == 0) return;
    for (uint64_t i = 0; i < n; ++i) {
        launch_guide(c, stream);
        launch_param<true, true, true>(c, stream);
    }
    launch_finalize(c, stream, 0, n, true);
}

// Replay of a resume graph: if it was captured with fresh allele tables at its head and they are not (cannot happen in
// a resume chain - its last launch was a PREP k_param with allele blocks - but nothing else promises it), k_allele runs
// first.  The graph ends with a PREP launch, i.e. in the state it was entered with.
static int launch_resume_graph(bean_hip_ctx* c, hipGraphExec_t ge, hipStream_t stream) {
    const DevArgs& d = c->d;
    if (d.family == kMultiMixture && c->resume_head_fresh && !c->alleles_fresh && d.n_live_slots > 0)
        hipLaunchKernelGGL(k_allele, dim3((unsigned)(((long)d.n_live_slots + 255) / 256)), dim3(256), 0, stream, d);
    HIP_OK(hipGraphLaunch(ge, stream));
    c->alleles_fresh = c->resume_head_fresh;
    return 0;
}

static int capture_resume_pairs(bean_hip_ctx* c, hipStream_t stream, uint64_t n, hipGraphExec_t* out) {
    hipGraph_t graph = nullptr;
    HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    // the graph's first node is a guide launch: it is captured in the state a resume chain enters it with - behind a
    // PREP launch of k_param (resume_head_state) - and a replay checks that state (launch_resume_graph)
    const bool fresh_was = c->alleles_fresh;
    c->alleles_fresh = c->resume_head_fresh;
    enqueue_resume_pairs(c, stream, n);
    c->alleles_fresh = fresh_was;
    hipError_t e = hipStreamEndCapture(stream, &graph);
    if (e != hipSuccess) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
            hipGraph_t junk = nullptr;
            (void)hipStreamEndCapture(stream, &junk);
            if (junk) (void)hipGraphDestroy(junk);
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    }
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        *out = nullptr;
        return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
    return 0;
}

extern "C" int bean_hip_svi_resume(bean_hip_ctx* c, uint64_t seed, uint64_t first_step, uint64_t n_steps,
                                   int32_t graph_chunk, void* stream_) {
    if (!c) return fail("bean_hip_svi_resume: null handle");
    if (!c->prepared) return fail("bean_hip_svi_resume: call bean_hip_prepare first");
    const DevArgs& dd = c->d;
    // noise injected or dumped per step, a timed k_param, the opt-in steppers: the plain loop
    if (dd.eps_mu_in || dd.eps_sd_in || dd.pi_in || dd.eps_noise_in || dd.eps_mu_out || dd.eps_sd_out || dd.eps_noise_out ||
        dd.x0_in || dd.eps_u_in || dd.x0_out || dd.eps_u_out || (dd.flags & kDumpPi) || c->profile || c->profile_param ||
        c->fused_step || c->tile_svi || stream_ == nullptr)
        return bean_hip_svi_run(c, seed, first_step, n_steps, graph_chunk, stream_);
    if (check_bound(c, false, true)) return -1;
    if (n_steps == 0) return 0;
    if (first_step + n_steps > c->loss_capacity)
        return fail("bean_hip_svi_resume: loss_hist too small for first_step + n_steps");
    hipStream_t stream = (hipStream_t)stream_;
    drop_graph_on_seed_change(c, seed);
    const bool resumed = c->resume_ok && c->resume_next == first_step && c->resume_seed == seed &&
                         c->resume_stream == stream_;
    c->resume_ok = false;
    c->d.seed = seed;
    if (async_candidate(c, n_steps)) {
        // the whole window in one launch (bean_async_v2.hpp); windows chain exactly as the pairs do
        if (!resumed) {
            launch_set_step(c, stream, first_step, first_step, n_steps);
            launch_param<false, false, true>(c, stream);  // draw and tables of the first step
        }
        if (launch_svi_async(c, stream, first_step, first_step, n_steps)) return -1;
        launch_finalize(c, stream, first_step, n_steps, false);
        HIP_OK(hipGetLastError());
        c->resume_ok = true;
        c->resume_next = first_step + n_steps;
        c->resume_seed = seed;
        c->resume_stream = stream_;
        return 0;
    }
    int kmax = 0;
    if (graph_chunk > 1) {
        // graphs of 4, 8, ... <= graph_chunk pairs, all instantiated at the first call
        while ((8ull << kmax) <= (uint64_t)graph_chunk && kmax < 8) ++kmax;
        if ((int)c->graphs_resume.size() != kmax + 1) {
            drop_graph(c);
            c->resume_head_fresh = param_prep_leaves_alleles_fresh(c);
            for (int k = 0; k <= kmax; ++k) {
                hipGraphExec_t ge = nullptr;
                if (capture_resume_pairs(c, stream, 4ull << k, &ge)) {
                    drop_graph(c);
                    return -1;
                }
                c->graphs_resume.push_back(ge);
            }
            c->graph_seed = seed;
        }
    }
    if (!resumed) {
        launch_set_step(c, stream, first_step, first_step, n_steps);
        launch_param<false, false, true>(c, stream);  // draw and tables of the first step
    }
    uint64_t left = n_steps;
    if (graph_chunk > 1 && !c->graphs_resume.empty()) {
        // one to four pairs directly - the device starts on an eager launch some 15 us sooner than on a graph
        // launch, and works on them while the host submits the graphs - then graphs, smallest first
        const uint64_t head = left % 4 ? left % 4 : (left >= 4 ? 4 : 0);
        enqueue_resume_pairs(c, stream, head);
        left -= head;
        const uint64_t big = left >> (kmax + 2);
        const uint64_t rest = left - (big << (kmax + 2));
        for (int k = 0; k < kmax; ++k)
            if (rest & (4ull << k))
                if (launch_resume_graph(c, c->graphs_resume[k], stream)) return -1;
        for (uint64_t i = 0; i < big; ++i)
            if (launch_resume_graph(c, c->graphs_resume[kmax], stream)) return -1;
        left = 0;
    }
    enqueue_resume_pairs(c, stream, left);
    HIP_OK(hipGetLastError());
    // the last k_param has drawn step first_step + n_steps and filled its tables
    c->resume_ok = true;
    c->resume_next = first_step + n_steps;
    c->resume_seed = seed;
    c->resume_stream = stream_;
    return 0;
}

// ---- guide-sharded stepping with exchange points (see bean_hip.h)
extern "C" int bean_hip_sharded_begin(bean_hip_ctx* c, uint64_t seed, uint64_t first_step, uint64_t n_steps,
                                      void* stream_) {
    if (!c) return fail("bean_hip_sharded_begin: null handle");
    if (!c->prepared) return fail("bean_hip_sharded_begin: call bean_hip_prepare first");
    if (check_bound(c, false, true)) return -1;
    if (first_step + n_steps > c->loss_capacity)
        return fail("bean_hip_sharded_begin: loss_hist too small for first_step + n_steps");
    if (is_survival(c->shape) && c->shape.family == BEAN_FAMILY_MIXTURE_NORMAL && !c->slot_ptr[BEAN_BUF_XCHG_GSUM])
        return fail("bean_hip_sharded_begin: bind BEAN_BUF_XCHG_GSUM for a sharded survival MixtureNormal fit");
    if (is_surv_normal(c->shape) && (!c->slot_ptr[BEAN_BUF_XCHG_GSUM] || !c->slot_ptr[BEAN_BUF_XCHG_SQ]))
        return fail("bean_hip_sharded_begin: bind BEAN_BUF_XCHG_GSUM and BEAN_BUF_XCHG_SQ for a sharded survival "
                    "NormalModel fit");
    if ((c->shape.family == BEAN_FAMILY_CONTROL_NORMAL || is_tiling(c->shape)) && !c->slot_ptr[BEAN_BUF_XCHG_TGRAD])
        return fail("bean_hip_sharded_begin: bind BEAN_BUF_XCHG_TGRAD for a sharded ControlNormal / tiling fit");
    hipStream_t stream = (hipStream_t)stream_;
    c->d.seed = seed;
    c->resume_ok = false;
    launch_set_step(c, stream, first_step, first_step, n_steps);
    launch_param<false, false, true>(c, stream);
    c->sharded_steps = 0;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bean_hip_sharded_sums(bean_hip_ctx* c, void* stream_) {
    if (!c) return fail("bean_hip_sharded_sums: null handle");
    // the normalisers of the survival families' Dirichlet-over-all-guides draw are formed at the tail of
    // the k_param launch that prepares the step (bean_hip_sharded_begin / _update): this rank's part is
    // already in BEAN_BUF_XCHG_GSUM when this returns.  Kept as the exchange point of the protocol.
    (void)stream_;
    return 0;
}

extern "C" int bean_hip_sharded_guide(bean_hip_ctx* c, void* stream_) {
    if (!c) return fail("bean_hip_sharded_guide: null handle");
    hipStream_t stream = (hipStream_t)stream_;
    launch_guide(c, stream);
    if (c->slot_ptr[BEAN_BUF_XCHG_TGRAD]) {
        int ntb, nb;
        grid_param(c, ntb, nb);
        hipLaunchKernelGGL(k_target_reduce, dim3(ntb), dim3(kParamBlock), 0, stream, c->d,
                           (double*)c->slot_ptr[BEAN_BUF_XCHG_TGRAD]);
    }
    if (c->d.n_cov && c->slot_ptr[BEAN_BUF_XCHG_COV])
        hipLaunchKernelGGL(k_cov_sum, dim3(c->d.R), dim3(1024), 0, stream, c->d);
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" int bean_hip_sharded_update(bean_hip_ctx* c, int32_t last, void* stream_) {
    if (!c) return fail("bean_hip_sharded_update: null handle");
    hipStream_t stream = (hipStream_t)stream_;
    const double* tg = (const double*)c->slot_ptr[BEAN_BUF_XCHG_TGRAD];
    const bool covx = c->d.n_cov && c->slot_ptr[BEAN_BUF_XCHG_COV];
    if (last)
        launch_param<true, true, false>(c, stream, tg, covx);
    else
        launch_param<true, true, true>(c, stream, tg, covx);
    // the loss slots of the run's steps are finalized ONCE, by its last update (a launch per step cost every
    // exchanged step 4.7 us of device time - a twentieth of a 1/8-size tiling step, a tenth of a survival one)
    ++c->sharded_steps;
    if (last) {
        launch_finalize(c, stream, 0, c->sharded_steps, true);  // the slots that end with the step just finished
        c->sharded_steps = 0;
    }
    HIP_OK(hipGetLastError());
    return 0;
}


// ---- RCCL owned by the library (see bean_hip.h): resolved at run time from the shared object the
// caller names (the copy PyTorch has loaded), so libbean_hip.so has no link-time dependency on it
struct RcclApi {
    void* handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char* (*GetErrorString)(ncclResult_t);
};
static RcclApi g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

static int rccl_load(const char* path) {
    if (g_rccl.handle) return 0;
    void* h = nullptr;
    if (path && path[0]) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(std::string("RCCL not found (") + (path ? path : "") + "): " + (dlerror() ? dlerror() : ""));
    RcclApi a;
    a.handle = h;
#define BEAN_RCCL_SYM(field, name)                                               \
    a.field = (decltype(a.field))dlsym(h, name);                                 \
    if (!a.field) return fail(std::string("RCCL symbol missing: ") + name)
    BEAN_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    BEAN_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    BEAN_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    BEAN_RCCL_SYM(AllReduce, "ncclAllReduce");
    BEAN_RCCL_SYM(GroupStart, "ncclGroupStart");
    BEAN_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    BEAN_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef BEAN_RCCL_SYM
    g_rccl = a;
    return 0;
}
#define RCCL_OK(expr)                                                                        \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess)                                                               \
            return fail(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));              \
    } while (0)

extern "C" int bean_hip_comm_unique_id(const char* rccl_path, uint8_t* id) {
    if (!id) return fail("bean_hip_comm_unique_id: null id");
    if (rccl_load(rccl_path)) return -1;
    static_assert(sizeof(ncclUniqueId) == BEAN_HIP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    RCCL_OK(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return 0;
}

extern "C" int bean_hip_comm_init(bean_hip_ctx* c, const char* rccl_path, const uint8_t* id, int32_t rank,
                                  int32_t world) {
    if (!c || !id) return fail("bean_hip_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail("bean_hip_comm_init: rank / world out of range");
    if (c->comm) return fail("bean_hip_comm_init: communicator already initialised");
    if (rccl_load(rccl_path)) return -1;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    RCCL_OK(g_rccl.CommInitRank(&comm, world, u, rank));
    c->comm = comm;
    c->comm_world = world;
    drop_graph(c);
    return 0;
}

extern "C" int bean_hip_comm_all_reduce(bean_hip_ctx* c, double* buf, uint64_t n, void* stream_) {
    if (!c || !buf) return fail("bean_hip_comm_all_reduce: null argument");
    if (!c->comm) return fail("bean_hip_comm_all_reduce: no communicator (bean_hip_comm_init)");
    RCCL_OK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream_));
    return 0;
}

extern "C" int bean_hip_comm_destroy(bean_hip_ctx* c) {
    if (!c || !c->comm) return 0;
    drop_graph(c);
    ncclComm_t comm = c->comm;
    c->comm = nullptr;
    c->comm_world = 0;
    RCCL_OK(g_rccl.CommDestroy(comm));
    return 0;
}

// one exchanged step on `stream`: [all-reduce gsum] guide [all-reduce tgrad (+ sq)] update
// (fin_n > 0: this step closes the run - the loss slots of its fin_n steps are finalized behind it)
static int enqueue_exchanged_step(bean_hip_ctx* c, hipStream_t stream, bool last, uint64_t fin_n = 0) {
    const bean_hip_shape& s = c->shape;
    double* gsum = (double*)c->slot_ptr[BEAN_BUF_XCHG_GSUM];
    double* tg = (double*)c->slot_ptr[BEAN_BUF_XCHG_TGRAD];
    double* sq = (double*)c->slot_ptr[BEAN_BUF_XCHG_SQ];
    if (gsum) RCCL_OK(g_rccl.AllReduce(gsum, gsum, (size_t)s.n_reps + 1, ncclFloat64, ncclSum, c->comm, stream));
    launch_guide(c, stream);
    if (tg) {
        int ntb, nb;
        grid_param(c, ntb, nb);
        hipLaunchKernelGGL(k_target_reduce, dim3(ntb), dim3(kParamBlock), 0, stream, c->d, tg);
    }
    double* cov = c->d.n_cov ? (double*)c->slot_ptr[BEAN_BUF_XCHG_COV] : nullptr;
    if (cov) hipLaunchKernelGGL(k_cov_sum, dim3(c->d.R), dim3(1024), 0, stream, c->d);
    const int n_coll = (tg ? 1 : 0) + (sq ? 1 : 0) + (cov ? 1 : 0);
    {
        // a group that has been opened is ALWAYS closed, whatever fails inside it: a rank that returned between
        // GroupStart and GroupEnd would leave its peers inside the collective (and a capturing stream half-captured)
        const bool grouped = n_coll > 1;
        ncclResult_t r = ncclSuccess, rg = ncclSuccess;
        const char* what = "";
        if (grouped) {
            r = g_rccl.GroupStart();
            what = "ncclGroupStart";
        }
        const bool opened = grouped && r == ncclSuccess;
        if (r == ncclSuccess && tg) {
            r = g_rccl.AllReduce(tg, tg, (size_t)2 * s.n_targets, ncclFloat64, ncclSum, c->comm, stream);
            what = "ncclAllReduce(tgrad)";
        }
        if (r == ncclSuccess && sq) {
            r = g_rccl.AllReduce(sq, sq, (size_t)s.n_reps, ncclFloat64, ncclSum, c->comm, stream);
            what = "ncclAllReduce(sq)";
        }
        if (r == ncclSuccess && cov) {
            r = g_rccl.AllReduce(cov, cov, (size_t)s.n_reps, ncclFloat64, ncclSum, c->comm, stream);
            what = "ncclAllReduce(cov)";
        }
        if (opened) rg = g_rccl.GroupEnd();
        if (r != ncclSuccess) return fail(std::string(what) + ": " + g_rccl.GetErrorString(r));
        if (rg != ncclSuccess) return fail(std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(rg));
    }
    if (last) launch_param<true, true, false>(c, stream, tg, cov != nullptr);
    else launch_param<true, true, true>(c, stream, tg, cov != nullptr);
    if (fin_n) launch_finalize(c, stream, 0, fin_n, true);  // the slots that end with the step just finished
    return 0;
}

extern "C" int bean_hip_svi_run_exchanged(bean_hip_ctx* c, uint64_t seed, uint64_t first_step, uint64_t n_steps,
                                          int32_t graph_chunk, void* stream_) {
    if (!c) return fail("bean_hip_svi_run_exchanged: null handle");
    if (!c->comm) return fail("bean_hip_svi_run_exchanged: call bean_hip_comm_init first");
    if (n_steps == 0) return 0;
    hipStream_t stream = (hipStream_t)stream_;
    drop_graph_on_seed_change(c, seed);
    if (bean_hip_sharded_begin(c, seed, first_step, n_steps, stream_)) return -1;  // checks, loss window, draw of step 0
    uint64_t left = n_steps;
    // hipGraphs of 2, 4, ... <= graph_chunk exchanged steps (never holding a run's last step, whose update
    // prepares no further draw); the remainder is enqueued directly
    if (graph_chunk > 1 && stream != nullptr && !c->profile) {
        int kmax = 0;
        while ((4ull << kmax) <= (uint64_t)graph_chunk && kmax < 9) ++kmax;
        if (c->graphs_xchg.empty()) {
            for (int k = 0; k <= kmax; ++k) {
                hipGraph_t graph = nullptr;
                hipGraphExec_t ge = nullptr;
                bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                if (ok) {
                    // (tiling: a graph's first node is a guide launch; it is captured in the state every replay enters it
                    // with - behind a PREP launch of k_param, bean_hip_sharded_begin's or an exchanged update's - and a
                    // replay checks that state, as launch_resume_graph does)
                    const bool fresh_was = c->alleles_fresh;
                    c->resume_head_fresh = param_prep_leaves_alleles_fresh(c);
                    c->alleles_fresh = c->resume_head_fresh;
                    for (uint64_t i = 0; ok && i < (2ull << k); ++i) ok = enqueue_exchanged_step(c, stream, false) == 0;
                    c->alleles_fresh = fresh_was;
                    // the capture is ended whether or not a step failed inside it (enqueue_exchanged_step closes
                    // its RCCL group on every path); a partial graph is destroyed below
                    hipError_t e = hipStreamEndCapture(stream, &graph);
                    ok = ok && e == hipSuccess && graph != nullptr;
                }
                if (ok) ok = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0) == hipSuccess;
                if (graph) (void)hipGraphDestroy(graph);
                if (!ok) {
                    // RCCL (or HIP) refused the capture: forget graphs, keep going eagerly
                    (void)hipGetLastError();
                    drop_graph(c);
                    g_err = "bean_hip_svi_run_exchanged: the exchanged step could not be captured into a hipGraph; "
                            "running it with eager launches";
                    break;
                }
                c->graphs_xchg.push_back(ge);
            }
            c->graph_seed = seed;
        }
        for (int k = (int)c->graphs_xchg.size() - 1; k >= 0; --k)
            while (left > (2ull << k)) {  // strictly more: the last step stays outside the graphs
                if (launch_resume_graph(c, c->graphs_xchg[k], stream)) return -1;  // (ends with an exchanged PREP update)
                left -= 2ull << k;
            }
    }
    for (; left > 0; --left)
        if (enqueue_exchanged_step(c, stream, left == 1, left == 1 ? n_steps : 0)) return -1;
    HIP_OK(hipGetLastError());
    return 0;
}

extern "C" uint64_t bean_hip_step_bytes(const bean_hip_ctx* c) {
    if (!c) return 0;
    const bean_hip_shape& s = c->shape;
    const uint64_t R = s.n_reps, B = s.n_condits, G = s.n_guides, T = s.n_targets, A = s.n_max_alleles;
    const bool bc = s.flags & BEAN_FLAG_USE_BCMATCH;
    // counts (f32) once per likelihood, repguide mask, per-guide a0 / a0_bc / g2t
    uint64_t bytes = G * (4 * R * B * (bc ? 2 : 1) + R + 8 * (bc ? 2 : 1) + 4);
    if (is_mixture(s)) bytes += G * (4 * R * s.n_ctrl * A + 8 /*pi_a0*/ + 3 * 4 * A * 2 /*alpha_pi, m, v r+w*/);
    if (is_tiling(s)) bytes += G * A /*allele_mask*/ + 4 * (G * (A - 1) + 1) + 2 * 4 * (uint64_t)s.n_a2e_nnz + 4 * (T + 1);
    if (s.flags & BEAN_FLAG_SCALE_BY_ACC) bytes += G * (8 + ((s.flags & BEAN_FLAG_FIT_NOISE) ? 2 * 3 * 4 * 2 : 0));
    bytes += T * ((is_survival(s) ? 2 : 4) * 3 * 4 * 2);  // per-target params with moments, read + written
    if (is_survival(s) && s.family == BEAN_FAMILY_MIXTURE_NORMAL) bytes += G * (3 * 4 * 2 /*q0*/ + 8 * R /*log_obs0*/);
    return bytes;
}

extern "C" const char* bean_hip_dominant_kernel(const bean_hip_ctx* c) {
    if (c && c->d.family == kMultiMixture)
        return c->tiling_wide ? "k_guide_tiling_wide"
                              : (c->tiling_rep ? "k_guide_tiling_rep" : (c->tiling_wave ? "k_guide_tiling_wave" : "k_guide_tiling"));
    if (c && c->d.survival) return c->surv_wave ? "k_guide_survival_wave" : "k_guide_survival";
    if (c && c->wave_guide && c->tile_svi && c->tile_ready) return "k_svi_tile";
    if (c && c->wave_guide && c->wave2 && c->async_step) return "k_svi_async";
    if (c && c->wave_guide) return c->wave2 ? (c->fused_step ? "k_step_wave2" : "k_guide_wave2") : "k_guide_wave";
    return "k_lik";
}

extern "C" const char* bean_hip_dominant_kernel_variant(const bean_hip_ctx* c) {
    static thread_local std::string name;
    if (!c) return "";
    const DevArgs& d = c->d;
    const char* acc = (d.flags & kAcc) ? "true" : "false";
    const char* surv = d.survival ? "true" : "false";
    const std::string base = bean_hip_dominant_kernel(c);
    if (d.family == kMultiMixture) name = base + "<" + acc + ", " + surv + ">";
    else if (base == "k_guide_wave2" || base == "k_guide_survival_wave" || base == "k_step_wave2" || base == "k_svi_tile" ||
             base == "k_guide_wave" || base == "k_svi_async")
        name = base + "<" + (d.family == kMixture ? "2" : "0") + ", " + (d.family == kMixture ? acc : "false") + ">";
    else name = base;
    return name.c_str();
}

extern "C" uint64_t bean_hip_dominant_lds_bytes(const bean_hip_ctx* c) {
    if (!c) return 0;
    const DevArgs& d = c->d;
    const bool acc = (d.flags & kAcc) != 0;
    if (d.family == kMultiMixture) {
        if (c->tiling_wide) return 0;
        const uint64_t nt = c->tiling_rep ? 64ull * c->tiling_rep_w : 64ull;
        return guide_tiling_lds(d.B, acc, (size_t)nt, !c->tiling_rep);
    }
    if (d.survival) return c->surv_wave ? guide_survival_wave_lds(d.B) : 0;
    if (c->wave_guide && c->wave2) return guide_wave2_lds(d.B, d.tile_targets);
    return 0;
}

extern "C" int bean_hip_set_profile(bean_hip_ctx* c, int32_t enable) {
    if (!c) return fail("bean_hip_set_profile: null handle");
    c->profile = enable != 0;
    c->profile_param = enable == 2;
    if (c->profile) drop_graph(c);
    return 0;
}

extern "C" int bean_hip_get_profile(bean_hip_ctx* c, double* avg_ms, uint64_t* launches) {
    if (!c || !avg_ms || !launches) return fail("bean_hip_get_profile: null argument");
    double total = 0.0;
    uint64_t n = 0;
    for (size_t i = 0; i + 1 < c->ev.size(); i += 2) {
        HIP_OK(hipEventSynchronize(c->ev[i + 1]));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        total += ms;
        // a launch of k_svi_tile covers all the steps of its call: the average is per SVI step
        n += (c->ev_steps.size() * 2 == c->ev.size()) ? c->ev_steps[i / 2] : 1;
    }
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    c->ev.clear();
    c->ev_steps.clear();
    *avg_ms = n ? total / (double)n : 0.0;
    *launches = n;
    return 0;
}

#ifdef BEAN_ASYNC_STAMP
// diagnostic builds only: the item timeline of the last k_svi_async call (bean_async_v2.hpp)
extern "C" int64_t bean_hip_async_stamps(bean_hip_ctx* c, unsigned long long* host, uint64_t n_words) {
    HIP_OK(hipDeviceSynchronize());
    if (!c->async_stamps) return 0;
    const uint64_t n = n_words < c->async_stamp_words ? n_words : c->async_stamp_words;
    HIP_OK(hipMemcpy(host, c->async_stamps, n * 8, hipMemcpyDeviceToHost));
    return (int64_t)c->async_stamp_words;
}
#endif

#ifdef BEAN_STAMP
// diagnostic builds only: copy the per-wave cycle stamps of the last k_lik launch to the host
extern "C" int bean_hip_debug_stamps(bean_hip_ctx* c, unsigned long long* host, uint64_t n_words) {
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(host, c->d.dbg, n_words * 8, hipMemcpyDeviceToHost));
    return 0;
}
#endif

extern "C" int bean_hip_test_special(int32_t op, uint64_t n, const double* a, const double* x,
                                     const double* b, double* out0, double* out1, void* stream_) {
    if (op < 0 || op > 9) return fail("bean_hip_test_special: unknown op");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_test_special, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       (int)op, (long)n, a, x, b, out0, out1);
    HIP_OK(hipGetLastError());
    return 0;
}

/*
 * bean_hip.h - C ABI of the MI355X-native BEAN SVI step (libbean_hip.so).
 *
 * The reference has no FFI boundary for this path: the seam is the Python call
 *   run_inference(model, guide, data, initial_lr, gamma, num_steps)
 *       (bean/model/run.py:347-396)
 * which drives pyro.infer.SVI over the model/guide pairs of
 * bean/model/model.py and bean/model/survival_model.py.  This header is the
 * C-ABI a maintainer would bind (ctypes) to replace the body
 * of that loop.  Entry points and what they replace:
 *
 *   bean_hip_elbo_grad   one Trace_ELBO.loss_and_grads evaluation
 *                        (svi.step minus the optimiser; run.py:377)
 *   bean_hip_adam        pyro.optim.ClippedAdam update (run.py:371)
 *   bean_hip_svi_run     the whole "for t in range(num_steps): svi.step(data)"
 *                        loop (run.py:376-380) with the loss history kept on
 *                        the device
 *
 * Conventions
 *   - plain C: opaque handle, int status codes (0 = ok, < 0 = error; the message
 *     is available from bean_hip_last_error()); nothing throws across the ABI.
 *   - every data / parameter / optimiser-state / gradient buffer is a
 *     caller-owned DEVICE pointer bound to a slot with bean_hip_bind(); the
 *     library never frees or retains them beyond the handle's lifetime.  The
 *     handle owns only scratch workspace.
 *   - all launches go to the hipStream_t passed in (void* here so that the
 *     header needs no HIP include); no call synchronises the host unless noted.
 *   - one handle per (device, stream); calls on one handle are serialised by
 *     the caller.
 *   - random draws come from a counter-based generator keyed by
 *     (seed, site, element, step): results do not depend on grid shape or on
 *     how guides are sharded across GPUs.
 *
 * Tensor layouts (R reps, B conditions, G guides, T targets, A alleles,
 * C control conditions) follow the reference's ScreenData contract
 * (bean/preprocessing/data_class.py): counts are
 * (R, B, G) float32 with G contiguous.
 */
#ifndef BEAN_HIP_H
#define BEAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bean_hip_ctx bean_hip_ctx;

/* model families: bean/model/run.py:399-474 (identify_model_guide) */
enum bean_hip_family {
    BEAN_FAMILY_NORMAL = 0,          /* NormalModel/NormalGuide            model.py:19,754  */
    BEAN_FAMILY_CONTROL_NORMAL = 1,  /* ControlNormalModel/Guide           model.py:168,861 */
    BEAN_FAMILY_MIXTURE_NORMAL = 2,  /* MixtureNormalModel/Guide           model.py:378,785 */
    BEAN_FAMILY_MULTI_MIXTURE = 3    /* MultiMixtureNormalModel/Guide      model.py:550,878 */
};

enum bean_hip_selection { BEAN_SELECTION_SORTING = 0, BEAN_SELECTION_SURVIVAL = 1 };

enum bean_hip_flags {
    BEAN_FLAG_USE_BCMATCH = 1,     /* second DirMult site on X_bcmatch (model.py:536-547) */
    BEAN_FLAG_SCALE_BY_ACC = 2,    /* scale_pi_by_accessibility (utils.py:106-178)         */
    BEAN_FLAG_FIT_NOISE = 4,       /* guide learns noise_loc/noise_scale (utils.py:144-155) */
    BEAN_FLAG_PRIOR_NORMAL_MU = 8, /* --prior-params: Normal instead of Laplace (model.py:408-420) */
    BEAN_FLAG_DUMP_PI = 16,        /* write the Dirichlet draws to BEAN_BUF_PI_OUT (tests)  */
    BEAN_FLAG_NOT_LOSS_OWNER = 32  /* guide-sharded fit of a family with replicated per-target
                                      parameters: another rank counts their prior/entropy terms */
};

typedef struct bean_hip_shape {
    int32_t family;        /* enum bean_hip_family */
    int32_t selection;     /* enum bean_hip_selection */
    int32_t flags;         /* OR of enum bean_hip_flags */
    int32_t n_reps;        /* R */
    int32_t n_condits;     /* B: sort bins incl. the control pseudo-bin, or timepoints (<= 8 in libbean_hip.so,
                              <= 64 in libbean_hip_a16.so / _a32.so) */
    int32_t n_guides;      /* G */
    int32_t n_targets;     /* T (variant); = n_edits for tiling: the per-target parameters are per edit */
    int32_t n_max_alleles; /* A (2 in variant mode) */
    int32_t n_edits;       /* E (tiling) */
    int32_t n_ctrl;        /* C: control conditions in allele_counts_control */
    int32_t mask_thres;    /* guides with sum_b x <= mask_thres are masked (10) */
    int32_t max_target_len;/* longest target (guides); very long targets get one block each */
    /* position of this shard inside the whole screen: the random streams are
     * keyed by GLOBAL guide / target indices, so a guide-sharded multi-GPU fit
     * reproduces the single-GPU fit bit for bit */
    int32_t guide_offset;  /* index of this shard's first guide in the whole screen */
    int32_t target_offset; /* index of this shard's first target */
    int32_t n_guides_total;/* guides in the whole screen (0 = n_guides) */
    int32_t n_a2e_nnz;     /* tiling: total number of (allele, edit) pairs */
    double sd_prior_scale; /* LogNormal prior scale of sd_targets (0.01; 1.0 for ControlNormal) */
    double initial_lr;     /* ClippedAdam lr (0.01) */
    double lrd;            /* per-step lr decay gamma ** (1 / num_steps) */
    double clip_norm;      /* 10 */
    /* survival MixtureNormal: prior of the per-guide baseline growth mu_negctrl
     * (survival_model.py:225,271-274); fed from the neg-ctrl fit by bean run */
    double negctrl_loc;    /* 0.0 */
    double negctrl_scale;  /* 0.1 */
    /* sorting NormalModel with sample covariates (bean/model/model.py:73-91, 771-783): number of
     * covariates (0 = none).  The mu_cov_loc / mu_cov_scale parameters (n_sample_covariates each) use
     * the NOISE_LOC / NOISE_SCALE parameter slots and the eps draws the EPS_NOISE_IN / _OUT slots,
     * which this family does not otherwise use. */
    int32_t n_sample_covariates;
    int32_t reserved_;
    /* survival NormalModel with prior_params["initial_abundance"] (survival_model.py:38-49): sum of that
     * per-guide prior concentration over the WHOLE screen (0 = the default prior ones / G); the per-guide
     * values of this shard go in BEAN_BUF_PRIOR_IA */
    double prior_ia_total;
} bean_hip_shape;

/* Buffer slots.  dtype / shape in brackets; "opt" = only for some families. */
enum bean_hip_buf {
    /* ---- data (read-only) */
    BEAN_BUF_X = 0,           /* f32 (R,B,G)   X_masked                               */
    BEAN_BUF_X_BC,            /* f32 (R,B,G)   X_bcmatch_masked                  opt  */
    BEAN_BUF_ALLELE_CTRL,     /* f32 (R,C,G,A) allele_counts_control             opt  */
    BEAN_BUF_REPGUIDE,        /* u8  (R,G)     repguide_mask                          */
    BEAN_BUF_SIZE_FACTOR,     /* f64 (R,B)                                            */
    BEAN_BUF_SIZE_FACTOR_BC,  /* f64 (R,B)                                       opt  */
    BEAN_BUF_SAMPLE_MASK,     /* f64 (R,B)     0/1                                    */
    BEAN_BUF_A0,              /* f64 (G)                                              */
    BEAN_BUF_A0_BC,           /* f64 (G)                                         opt  */
    BEAN_BUF_PI_A0,           /* f64 (G)                                         opt  */
    BEAN_BUF_Z_HI,            /* f64 (B)  Phi^-1(upper quantile), +inf where it is 1  */
    BEAN_BUF_Z_LO,            /* f64 (B)  Phi^-1(lower quantile), -inf where it is 0  */
    BEAN_BUF_TARGET_OFFSETS,  /* i32 (T+1) exclusive prefix sum of target_lengths     */
    BEAN_BUF_GUIDE_TO_TARGET, /* i32 (G)                                              */
    BEAN_BUF_ACCESSIBILITY,   /* f64 (G)                                         opt  */
    BEAN_BUF_PRIOR_MU_LOC,    /* f64 (T)  --prior-params                         opt  */
    BEAN_BUF_PRIOR_MU_SCALE,  /* f64 (T)                                         opt  */
    BEAN_BUF_PRIOR_SD_LOC,    /* f64 (T)                                         opt  */
    BEAN_BUF_PRIOR_SD_SCALE,  /* f64 (T)                                         opt  */
    /* tiling: allele slot s = g*(A-1) + (a-1) -> its edits (CSR), and the transpose */
    BEAN_BUF_A2E_PTR,         /* i32 (G*(A-1)+1)                                 opt  */
    BEAN_BUF_A2E_IDX,         /* i32 (nnz)  edit ids                             opt  */
    BEAN_BUF_E2A_PTR,         /* i32 (E+1)                                       opt  */
    BEAN_BUF_E2A_IDX,         /* i32 (nnz)  allele slots containing the edit     opt  */
    BEAN_BUF_ALLELE_MASK,     /* u8  (G,A)  allele_mask                          opt  */
    /* survival: timepoints divided by the last one (data_class.py:1034-1053) */
    BEAN_BUF_TIMEPOINTS,      /* f64 (B)                                         opt  */
    BEAN_BUF_CONTROL_TIME,    /* f64 (C)   timepoint(s) of the control condition opt  */
    BEAN_BUF_LOG_OBS0,        /* f64 (R,G) log((X[:,0,:]+1)/sum): observed initial abundance opt */
    BEAN_BUF_NEGCTRL_MASK,    /* u8  (G)   survival NormalModel: 1 where the guide is a negative control
                                 (mu forced to 0, survival_model.py:59-60)                   opt  */
    /* exchange buffers of guide-sharded fits (bean_hip_sharded_*): the library writes this rank's
       sums, the caller all-reduces them over the ranks, the library reads the totals */
    BEAN_BUF_XCHG_GSUM,       /* f64 (R+1) survival MixtureNormal: sum_g of the Dirichlet(q0) site's gamma
                                 draws per replicate, sum_g q0                               opt  */
    BEAN_BUF_XCHG_TGRAD,      /* f64 (2,T) ControlNormal / tiling: per-target likelihood gradient opt  */
    BEAN_BUF_XCHG_SQ,         /* f64 (R)   survival NormalModel: sum_g q_0[r,g] * d loss / d q_0[r,g], the
                                 projection term of the Dirichlet-over-guides pathwise gradient opt  */
    /* ---- sorting NormalModel with sample covariates */
    BEAN_BUF_REP_BY_COV = 31, /* f64 (R, n_sample_covariates) design matrix rep_by_cov (data_class.py:83-90) opt */
    /* ---- parameters: unconstrained values as Pyro's param store keeps them */
    BEAN_BUF_P_MU_LOC = 32,   /* f32 (T)                                              */
    BEAN_BUF_P_MU_SCALE,      /* f32 (T)   log mu_scale                               */
    BEAN_BUF_P_SD_LOC,        /* f32 (T)                                              */
    BEAN_BUF_P_SD_SCALE,      /* f32 (T)   log sd_scale                               */
    BEAN_BUF_P_ALPHA_PI,      /* f32 (G,A) log alpha_pi                          opt  */
    BEAN_BUF_P_NOISE_LOC,     /* f32 (G)                                         opt  */
    BEAN_BUF_P_NOISE_SCALE,   /* f32 (G)   log noise_scale                       opt  */
    BEAN_BUF_P_Q0,            /* f32 (G)   log q0 (survival MixtureNormal)       opt  */
    /* ---- gradients w.r.t. the unconstrained parameters (bean_hip_elbo_grad) */
    BEAN_BUF_G_MU_LOC = 48,
    BEAN_BUF_G_MU_SCALE,
    BEAN_BUF_G_SD_LOC,
    BEAN_BUF_G_SD_SCALE,
    BEAN_BUF_G_ALPHA_PI,
    BEAN_BUF_G_NOISE_LOC,
    BEAN_BUF_G_NOISE_SCALE,
    BEAN_BUF_G_Q0,
    /* ---- ClippedAdam first / second moments, same shapes as the parameters */
    BEAN_BUF_M_MU_LOC = 64,
    BEAN_BUF_M_MU_SCALE,
    BEAN_BUF_M_SD_LOC,
    BEAN_BUF_M_SD_SCALE,
    BEAN_BUF_M_ALPHA_PI,
    BEAN_BUF_M_NOISE_LOC,
    BEAN_BUF_M_NOISE_SCALE,
    BEAN_BUF_M_Q0,
    BEAN_BUF_V_MU_LOC = 80,
    BEAN_BUF_V_MU_SCALE,
    BEAN_BUF_V_SD_LOC,
    BEAN_BUF_V_SD_SCALE,
    BEAN_BUF_V_ALPHA_PI,
    BEAN_BUF_V_NOISE_LOC,
    BEAN_BUF_V_NOISE_SCALE,
    BEAN_BUF_V_Q0,
    /* ---- injected / exported noise (parity tests) */
    BEAN_BUF_EPS_MU_IN = 96,  /* f64 (T)   standard-normal draws for mu_targets  opt  */
    BEAN_BUF_EPS_SD_IN,       /* f64 (T)                                         opt  */
    BEAN_BUF_PI_IN,           /* f64 (R,G,A) Dirichlet draws                     opt  */
    BEAN_BUF_EPS_NOISE_IN,    /* f64 (G)                                         opt  */
    BEAN_BUF_EPS_MU_OUT,      /* f64 (T)   draws actually used                   opt  */
    BEAN_BUF_EPS_SD_OUT,      /* f64 (T)                                         opt  */
    BEAN_BUF_PI_OUT,          /* f64 (R,G,A)                                     opt  */
    BEAN_BUF_EPS_NOISE_OUT,   /* f64 (G)                                         opt  */
    BEAN_BUF_X0_IN,           /* f64 (R,G) survival: Dirichlet(q0) draws         opt  */
    BEAN_BUF_EPS_U_IN,        /* f64 (G)   survival: mu_negctrl standard normals opt  */
    BEAN_BUF_X0_OUT,          /* f64 (R,G)                                       opt  */
    BEAN_BUF_EPS_U_OUT,       /* f64 (G)                                         opt  */
    BEAN_BUF_PRIOR_IA,        /* f64 (G)   survival NormalModel: prior concentration of the Dirichlet-over-guides
                                 site, prior_params["initial_abundance"]                      opt  */
    BEAN_BUF_XCHG_COV,        /* f64 (R)   guide-sharded sorting NormalModel with sample covariates: sum over this
                                 rank's guides of each replicate's d nll / d mu row (the likelihood gradient of
                                 the shared mu_cov site), all-reduced by the caller between
                                 bean_hip_sharded_guide and bean_hip_sharded_update                opt  */
    BEAN_BUF_GUIDE_IDS,       /* i32 (G)   tiling: each guide's index in the caller's whole screen.  The per-guide
                                 random streams (pi draws, accessibility noise) are keyed by it instead of
                                 guide_offset + position, so a caller may hand the guides over in ANOTHER ORDER
                                 - crispr-bean_amd orders them by their number of alleles, which makes the lanes
                                 of a wave take the same branches - and still draw what the screen order draws  opt  */
    /* ---- loss */
    BEAN_BUF_LOSS_HIST = 112, /* f64 (capacity) one entry per SVI step                */
    BEAN_BUF_COUNT = 128
};

const char* bean_hip_version(void);
const char* bean_hip_last_error(void);

/* Create / destroy a handle for one screen shape on the current HIP device.
 * Allocates the scratch workspace (O(G + B*T) doubles). */
int bean_hip_create(const bean_hip_shape* shape, bean_hip_ctx** out);
int bean_hip_destroy(bean_hip_ctx* ctx);

/* Bind a caller-owned device buffer to a slot; nbytes is checked against the
 * size the shape implies (a mismatch is an error, nothing is launched). */
int bean_hip_bind(bean_hip_ctx* ctx, int slot, void* device_ptr, uint64_t nbytes);

/* Validate that every slot the family needs is bound, and run the one-off
 * data-only precomputation (masks, log-factorial constants of the likelihoods;
 * tiling: the list of allele slots that hold an allele, for which the allele
 * mask and the CSR row pointers are read back to the host once).  Synchronises
 * the stream.  Must be called after the data slots are bound and before any step. */
int bean_hip_prepare(bean_hip_ctx* ctx, void* stream);

/* One ELBO evaluation with gradients: draws the step's noise (or reads the
 * *_IN slots when bound), writes d loss / d param to the BEAN_BUF_G_* slots and
 * the loss to loss_hist[loss_index].  Parameters are not modified. */
int bean_hip_elbo_grad(bean_hip_ctx* ctx, uint64_t seed, uint64_t step,
                       uint64_t loss_index, void* stream);

/* ClippedAdam update of every bound parameter from the BEAN_BUF_G_* slots;
 * t is the 1-based update count (lr_t = initial_lr * lrd ** t). */
int bean_hip_adam(bean_hip_ctx* ctx, uint64_t t, void* stream);

/* Fused SVI loop: n_steps x {draw, ELBO, gradient, ClippedAdam}, steps
 * first_step .. first_step + n_steps - 1, loss of step s written to
 * loss_hist[s].  Kernels are enqueued through a captured hipGraph of
 * graph_chunk steps when graph_chunk > 0 (eager launches when 0). */
int bean_hip_svi_run(bean_hip_ctx* ctx, uint64_t seed, uint64_t first_step,
                     uint64_t n_steps, int32_t graph_chunk, void* stream);

/* The same loop for a fit that is stepped in windows (run_inference reports every 100 steps,
 * bean/model/run.py:378): results are those of bean_hip_svi_run, bit for bit, but the call ends with the
 * draw and the tables of step first_step + n_steps already on the device, and a call that continues exactly
 * there - same seed, same stream, first_step = the previous call's first_step + n_steps, nothing bound,
 * prepared or stepped through another entry point in between - starts stepping at once instead of with the
 * two preparing launches.  The caller asserts that it has not written the bound parameter / moment buffers
 * since the previous call returned (the library cannot see such writes; use bean_hip_svi_run after one).
 * Falls back to bean_hip_svi_run when per-step noise is injected or dumped. */
int bean_hip_svi_resume(bean_hip_ctx* ctx, uint64_t seed, uint64_t first_step,
                        uint64_t n_steps, int32_t graph_chunk, void* stream);

/* The same loop for guide-sharded fits of families in which something is shared across the
 * shards (the reference is single-process, so there is no interface to
 * mirror).  The step is cut at its exchange points; after each call that names a buffer the
 * caller all-reduces (sum) that caller-owned buffer over the ranks, on the same stream:
 *
 *   bean_hip_sharded_begin(first_step, n_steps)     once per run: zero the loss window, draw step 0
 *   per step:
 *     bean_hip_sharded_sums      -> BEAN_BUF_XCHG_GSUM   (survival MixtureNormal / NormalModel: normalisers
 *                                                         of the Dirichlet-over-guides draw; no-op otherwise)
 *     bean_hip_sharded_guide     -> BEAN_BUF_XCHG_TGRAD  (ControlNormal, tiling: per-target likelihood
 *                                                         gradients; not written when the slot is unbound)
 *                                -> BEAN_BUF_XCHG_SQ     (survival NormalModel: projection sums)
 *                                -> BEAN_BUF_XCHG_COV    (sorting NormalModel with sample covariates: the
 *                                                         replicates' gradient sums of the shared mu_cov site)
 *     bean_hip_sharded_update(last)                       gradients, ClippedAdam, draws of the next step;
 *                                                         last != 0 on the run's final step: no further draw, and
 *                                                         loss_hist of ALL the run's steps is written (once, here)
 *
 * Families whose parameters are all per-target or per-guide with target-aligned shards (sorting
 * Normal / MixtureNormal) need none of this: bean_hip_svi_run on every rank is the sharded fit. */
int bean_hip_sharded_begin(bean_hip_ctx* ctx, uint64_t seed, uint64_t first_step, uint64_t n_steps, void* stream);
int bean_hip_sharded_sums(bean_hip_ctx* ctx, void* stream);
int bean_hip_sharded_guide(bean_hip_ctx* ctx, void* stream);
int bean_hip_sharded_update(bean_hip_ctx* ctx, int32_t last, void* stream);

/* The same per-step exchange without the host in the loop: the library owns an RCCL communicator (one
 * process per GPU; RCCL is resolved at run time from the shared object `rccl_path` names - the one
 * PyTorch has already loaded - so that the library itself has no link-time dependency on it) and
 * bean_hip_svi_run_exchanged enqueues, per step,
 *     [all-reduce GSUM] guide kernels [all-reduce TGRAD (+ SQ, grouped)] update
 * on `stream`: ncclAllReduce(sum, float64, in place) on the bound BEAN_BUF_XCHG_* buffers between the
 * kernels, no host synchronisation, no Python.  graph_chunk > 0 additionally captures 2^k steps,
 * collectives included, into hipGraphs (opt-in: whether RCCL's kernels can be captured depends on the
 * RCCL build; on refusal the call falls back to eager launches and says so in bean_hip_last_error).
 *
 *   bean_hip_comm_unique_id   rank 0 only: fills id[128] (ncclGetUniqueId); the caller broadcasts it
 *   bean_hip_comm_init        every rank, collectively (ncclCommInitRank on the current device)
 *   bean_hip_comm_all_reduce  one ncclAllReduce(sum, float64, in place) of `n` doubles on that communicator and `stream`:
 *                             what bean_hip_svi_run_exchanged issues between its kernels, callable by itself so that a
 *                             caller can CHECK a fresh communicator against a known sum before a fit depends on it
 *   bean_hip_comm_destroy     ncclCommDestroy
 * Returns 0, or -1 with bean_hip_last_error() (RCCL missing, refused, ...): callers then keep stepping
 * with bean_hip_sharded_* and their own all-reduce. */
#define BEAN_HIP_COMM_ID_BYTES 128
int bean_hip_comm_unique_id(const char* rccl_path, uint8_t* id);
int bean_hip_comm_init(bean_hip_ctx* ctx, const char* rccl_path, const uint8_t* id, int32_t rank, int32_t world);
int bean_hip_comm_all_reduce(bean_hip_ctx* ctx, double* buf, uint64_t n, void* stream);
int bean_hip_comm_destroy(bean_hip_ctx* ctx);
int bean_hip_svi_run_exchanged(bean_hip_ctx* ctx, uint64_t seed, uint64_t first_step, uint64_t n_steps,
                               int32_t graph_chunk, void* stream);

/* Introspection for bench.py / DESIGN.md: algorithmic bytes one step moves
 * (each input read once, each parameter and moment read and written once) and
 * the name of the dominant kernel. */
uint64_t bean_hip_step_bytes(const bean_hip_ctx* ctx);
const char* bean_hip_dominant_kernel(const bean_hip_ctx* ctx);
/* Dynamic LDS (bytes per workgroup) the library requests when it launches that kernel for this shape -
 * the figure that, with the code object's register counts, decides how many waves a CU holds (profilers
 * report the static segment only). */
uint64_t bean_hip_dominant_lds_bytes(const bean_hip_ctx* ctx);
/* ... and the template instantiation that is launched for this shape, as it is spelt in the code object
 * (e.g. "k_guide_wave2<2, false>"), to look its register counts up there. */
const char* bean_hip_dominant_kernel_variant(const bean_hip_ctx* ctx);

/* Time (ms, HIP events on `stream`) of the dominant kernel averaged over the
 * launches issued since the previous call; enable with profile != 0 in
 * bean_hip_set_profile before running steps (eager mode only). */
int bean_hip_set_profile(bean_hip_ctx* ctx, int32_t enable);
int bean_hip_get_profile(bean_hip_ctx* ctx, double* avg_ms, uint64_t* launches);

/* Unit-test hooks for the device special functions (n elements, device ptrs):
 *   op 0: out0 = lgamma(a+x)-lgamma(a), out1 = digamma(a+x)-digamma(a)
 *   op 1: out0 = lgamma(a),             out1 = digamma(a)
 *   op 2: out0 = dirichlet_grad(x, a, total=b)
 *   op 3: out0 = Phi(a)
 *   op 4: out0, out1 = Dirichlet(a, b) draw (seed = x[0] bits, element index)
 *   op 5: as op 0 through the two-chain evaluation (element i paired with element i ^ 1)
 *   op 6: out0, out1 = max((float)Gamma(a), FLT_MIN), max((float)Gamma(b), FLT_MIN) from the plain pair sampler
 *   op 7: the same from the float32-floor sampler of the survival q0 site (must equal op 6 draw for draw)
 *   op 8: out0, out1 = max(Gamma(a), DBL_MIN), max(Gamma(b), DBL_MIN) from the plain pair sampler
 *   op 9: the same from the double-floor sampler of the wide tiling kernel (must equal op 8 draw for draw) */
int bean_hip_test_special(int32_t op, uint64_t n, const double* a, const double* x,
                          const double* b, double* out0, double* out1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BEAN_HIP_H */

// HIP kernels of the BEAN SVI step for gfx950 (MI355X).
//
// One SVI step of the variant sorting families is two kernels:
//
//   k_param  (per target + per guide, light)
//       FINISH: reduce the previous step's per-guide partials into parameter
//               gradients (guide -> target segmented reduce; priors and
//               entropies; Dirichlet normalisers), then ClippedAdam in place;
//       PREP:   draw eps for the next step, mu_t / sd_t, and tabulate the
//               edited-component bin probabilities P[b, t] and their
//               derivatives (the Normal-CDF work depends only on the target).
//
//   k_guide_wave  (per (rep, guide), heavy: ~80 % of the time)
//       one single-wave workgroup = 64 consecutive guides of one replicate
//       (coalesced reads of the (R, B, G) count tensors).  Per thread:
//       Dirichlet draw, mixture, both Dirichlet-Multinomial likelihoods with
//       analytic gradients, Multinomial on control allele counts, implicit-
//       reparameterisation gradient.  Per-replicate rows go to wrow; k_param
//       sums them over replicates in fixed order.
//   (survival and tiling screens: k_guide_survival / k_allele + k_guide_tiling,
//    blocks of R waves with an LDS reduction over replicates.)
//
// Reference semantics: bean/model/model.py (models/guides), bean/model/utils.py
// (get_alpha, get_std_normal_prob), SURVEY.md Appendix A/D for the algebra.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bean_special.hpp"

namespace bean {

// In-kernel cycle stamps for diagnostic builds only (-DBEAN_STAMP=1: the guide kernels, =2: k_param;
// they share one buffer); the shipped kernels contain none.
#ifdef BEAN_STAMP
#define BEAN_STAMP_WRITE(slot)                                                                   \
    do {                                                                                         \
        unsigned long long t_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        if (lane == 0) c.dbg[wave_gid * 8 + (slot)] = t_;                                        \
    } while (0)
#endif
#if defined(BEAN_STAMP) && BEAN_STAMP == 1
#define BEAN_STAMP_AT(slot) BEAN_STAMP_WRITE(slot)
#else
#define BEAN_STAMP_AT(slot) do {} while (0)
#endif
#if defined(BEAN_STAMP) && BEAN_STAMP == 4  // in-kernel clock: shader-clock and 100 MHz real-time stamps at both ends
#define BEAN_STAMP_CLK(slot)                                                                     \
    do {                                                                                         \
        unsigned long long t_, u_;                                                               \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(u_)::"memory"); \
        if (lane == 0) {                                                                         \
            c.dbg[wave_gid * 8 + (slot)] = t_;                                                   \
            c.dbg[wave_gid * 8 + (slot) + 1] = u_;                                               \
        }                                                                                        \
    } while (0)
#else
#define BEAN_STAMP_CLK(slot) do {} while (0)
#endif
#if defined(BEAN_STAMP) && BEAN_STAMP == 3  // the tail of the fused step kernel (one record per tile)
#define BEAN_STAMP_TL(slot) BEAN_STAMP_WRITE(slot)
#else
#define BEAN_STAMP_TL(slot) do {} while (0)
#endif
#if defined(BEAN_STAMP) && BEAN_STAMP == 2
#define BEAN_STAMP_KP(slot) BEAN_STAMP_WRITE(slot)
#else
#define BEAN_STAMP_KP(slot) do {} while (0)
#endif
#if defined(BEAN_STAMP) && BEAN_STAMP == 5  // k_param's block roles on the 100 MHz real-time clock (one record per block)
#define BEAN_STAMP_RT(rec, slot)                                                                 \
    do {                                                                                         \
        unsigned long long u_;                                                                   \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u_)::"memory");          \
        if (threadIdx.x == 0) c.dbg[(long)(rec) * 8 + (slot)] = u_;                              \
    } while (0)
#else
#define BEAN_STAMP_RT(rec, slot) do {} while (0)
#endif

constexpr double kEps = 1e-5;         // epsilon of get_alpha (utils.py:11)
// utils.py:133; the reference builds Normal(0, 0.655) from Python floats, i.e. float32 tensors
constexpr double kPiNoiseSd = (double)0.655f;
constexpr double kAccA = 0.2513;      // utils.py:82
constexpr float kAccBf = -1.9458f;    // utils.py:83 (exp taken in float32 there)
constexpr double kHalfLog2PiC = 0.91893853320467274178;
constexpr double kLog2 = 0.69314718055994530942;
constexpr double kDblMin = 2.2250738585072014e-308;
constexpr double kOneMinus = 0.99999999999999988898;  // nextafter(1, 0)
constexpr double kProbEps = 2.220446049250313e-16;    // torch clamp_probs eps (f64)

enum Family { kNormal = 0, kControlNormal = 1, kMixture = 2, kMultiMixture = 3 };
enum Flags { kUseBc = 1, kAcc = 2, kFitNoise = 4, kPriorNormalMu = 8, kDumpPi = 16 };

struct StepCtr {
    unsigned long long step;  // SVI step (RNG offset, Adam t = step + 1)
    unsigned long long slot;  // index into loss_hist
    float step_size;          // ClippedAdam step size of update t = step + 1 (publish_ctr)
    float pad_;
};

// partial-sum rows written by k_guide, (kNumPart, G) doubles
// (survival reuses row kPGy for the q0 gradient: it has no sd latent)
enum Part { kPGmu = 0, kPGy = 1, kPQ0 = 1, kPGnoise = 2, kPNrg = 3, kPPath = 4, kPLp = 6, kPLq = 8, kNumPart = 10 };

struct DevArgs {
    int R, B, G, T, A, C;
    int family, flags, mask_thres, wide_targets;
    int g_off, t_off, G_tot;  // shard position (RNG streams use global indices)
    double sd_prior_scale, lr0, log_lrd, clip;
    unsigned long long seed;
    // data
    const float* X;
    const float* Xbc;
    const float* allele;
    const uint8_t* rg;
    const double *sf, *sf_bc, *smask, *a0, *a0_bc, *pi_a0, *z_hi, *z_lo, *acc;
    const int *toff, *g2t;
    const double *pr_mu_loc, *pr_mu_scale, *pr_sd_loc, *pr_sd_scale;
    // parameters / grads / moments: mu_loc, mu_scale, sd_loc, sd_scale, alpha_pi, noise_loc, noise_scale, q0
    float* p[8];
    float* g[8];
    float* m[8];
    float* v[8];
    // noise
    const double *eps_mu_in, *eps_sd_in, *pi_in, *eps_noise_in;
    double *eps_mu_out, *eps_sd_out, *pi_out, *eps_noise_out;
    double* loss_hist;
    // order-independent loss accumulators (loss_add): kLossSub lines of kLossWords int64 words per
    // loss_hist slot, and one line for the data-only constant
    long long *loss_acc, *const_acc;
    // (6, G): lgamma / digamma of the guide-side Dirichlet concentrations c_q of the CURRENT alpha_pi:
    // rows lgamma(c_q0 + c_q1), lgamma(c_q0), lgamma(c_q1), then the three digammas.  Written by k_param
    // PREP (after the update), read by the next guide kernel (implicit-gradient calls) and by the next
    // k_param FINISH instead of being recomputed there.  Null: every user computes its own.
    // sorting NormalModel with sample covariates (model.py:73-91, 771-783): mu_cov ~ N(0, 1) per covariate,
    // guide N(mu_cov_loc, mu_cov_scale) in parameter slots 5 / 6; replicate r's means are shifted by
    // rep_by_cov[r, 0] * mu_cov[0] (only the first column enters the likelihood there), so the Phi tables
    // are per (replicate, target): (R, B, T)
    int n_cov;
    const double* rbc;                 // (R, n_cov) design matrix
    double *cov_mu, *cov_eps;          // (n_cov) current draw
    double *cov_shift, *cov_sum;       // (R) shift of replicate r; sum_g of its d nll / d mu rows
    // survival NormalModel: prior_params["initial_abundance"] (survival_model.py:38-49): per-guide prior
    // concentration of the Dirichlet-over-guides site and its sum over the WHOLE screen; null: ones / G
    const double* prior_ia;
    double* kacc;    // +Acc: exp(b) acc[g]^a of utils.py:106-131, per guide - data, formed once by bean_hip_prepare
    const int* gid;  // tiling, optional: the guide's index in the caller's whole screen (keys its random streams)
    double prior_ia_total;
    int trow_summed;                   // k_sum_trow has reduced trow into part
    int wide_alleles;                  // tiling with more alleles per guide than kAMax: bean_tiling_wide.hpp
    double* dgq;
    double* dgq_t;                     // tiling: (kAMax + 1, G) digamma(c_q[a]) rows and digamma(sum c_q), same contract
    long long* lpart;                  // (n_lpart, 3) per-wave loss parts of the wave-form guide kernels, or null
    int n_lpart;
    int rows_v2;                       // wrow holds the five rows of k_guide_wave2 (bean_guide_v2.hpp)
    // 1: the guide kernel leaves the total term lgamma(A0 + n) - lgamma(A0) of a Dirichlet-Multinomial
    // site to the constant wherever no bin sits on its floor: get_alpha normalises, so there
    // A0 = sum_b alpha_b = a0[g] is data (k_prepare adds the term once) and its derivative multiplies
    // sum_b d alpha_b = 0.  A (replicate, guide) with a floored bin adds the difference to that constant.
    int tot_const;
    int *tile_ctr, *bnd_ctr;           // fused step kernel: arrivals per 64-guide tile / per tile boundary
    int n_arrival_ctr;                 // ... their number (tile_ctr[0 .. n) covers both arrays; k_set_step zeroes them)
    // distinct finite bin edges (k_prepare): ue_z[n] their z values, n in ue_idx[2 B]; ue_idx[b] /
    // ue_idx[B + b]: the upper / lower edge of bin b in that list, -1 where the edge is infinite.  The
    // four sort bins + bulk of a standard screen have 10 edges, 4 of them distinct and finite.
    double* ue_z;
    int* ue_idx;
    // workspace
    double *tabP, *tabPmu, *tabPy;     // (B, T)
    double* P0;                        // (B)
    double *mu_t, *y_t, *eps_mu, *eps_sd;  // (T)
    double* part;                      // (kNumPart, G)
    double *lpn, *eps_noise;           // (G)
    double* loss_const;                // (1)
    double *pi_ws, *gpi_ws;            // (R, G, 2) split-kernel hand-off: draws, d nll / d pi
    unsigned long long* dbg;           // diagnostic builds (-DBEAN_STAMP): per-wave cycle stamps
    double* rrow;                      // (3, R, G) split form: d/dmu_t, d/dy_t, d/dnoise per (rep, guide)
    double* wrow;                      // (kNumPart, R, G) wave form: every per-guide row, per replicate
    int tile_targets;                  // wave form: most targets spanned by any 64-guide tile (<= 64)
    double* trow;                      // (kTNumPart, R, G) tiling wave form: per-replicate rows
    double* nobs;                      // (2, R, G) wave form: count totals of X / X_bcmatch, -1 where masked
    StepCtr *ctrA, *ctrB;
    // tiling (MultiMixtureNormal): CSR allele slot -> edits and its transpose
    int E;
    const int *a2e_ptr, *a2e_idx;  // (G*(A-1)+1), (nnz): slot = g*(A-1) + (a-1)
    const int *e2a_ptr, *e2a_idx;  // (E+1), (nnz): slots containing each edit
    const uint8_t* amask;          // (G, A)
    const int* live_slots;         // tiling: the n_live_slots allele slots k_allele fills, as a1 * G + g (bean_hip_prepare)
    int n_live_slots;
    double *mu_a, *sig_a;          // (A-1, G) allele mean / scale of the current draw
    // survival (exp(mu t) growth instead of Normal-CDF bins)
    int survival;
    const double *time, *ctrl_time;  // (B), (C)
    double neg_loc, neg_scale;       // prior of the per-guide baseline mu_negctrl
    const double *x0_in, *eps_u_in;
    const double* log_obs0;          // (R, G) log of the normalised t0 counts (data only)
    double *x0_out, *eps_u_out;
    double *u_g, *eps_u;             // (G) baseline draw of the current step
    double *gam;                     // (R, G) gamma draws of the Dirichlet(q0) site
    double *gpart;                   // (n_gamma_blocks, R + 1) block sums: gammas per rep, q0
    double *gsum;                    // (R + 1) totals: sum_g gamma[r, g], sum_g q0
    // survival NormalModel: q_0 ~ Dirichlet(initial_abundance) over all guides enters the likelihood
    int surv_q0lik;
    const uint8_t* negctrl;          // (G) guides whose mu is forced to 0
    // sharded runs with replicated per-target parameters
    const double* tgrad;             // (2, T) all-reduced likelihood gradients, or null: reduce locally
    int not_loss_owner;              // 1: another rank adds the loss terms of the replicated parameters
    double *gq;                      // (R, G) d loss / d q_0[r, g]
    double *sq;                      // (R) sum_g q_0[r, g] * gq[r, g]
    int n_gamma_blocks;
    int* q0_ctr;                     // arrivals of k_param's q0 blocks (last one forms gsum), zero between launches
    int q0_blocks;                   // survival: k_param has q0 blocks (n_gamma_blocks of them, kParamBlock guides each)
    int q0_blk0;                     // survival: k_param's guide-part blocks ahead of its q0 blocks (MixtureNormal: the alpha_pi blocks)
    int lpt;                         // k_param, thin mode: lanes per target (kLanesPerTarget; kLanesPerTargetNarrow: survival and tiling families)
    // k_guide_wave2 (bean_guide_v2.hpp): tiles are aligned to the GLOBAL guide index, and the wave leaves d/dmu_t,
    // d/dy_t summed per target part instead of per guide
    int g_sh;                        // g_off % 64: empty lanes at the head of this shard's first tile
    int n_tiles;                     // (g_sh + G + 63) / 64
    int seg_steps;                   // ceil(log2(min(longest target, 64))): steps of the in-wave segmented scan
    double* tsum;                    // (2, R, n_tiles * tile_targets): sums per (replicate, tile, target of the tile); null: per-guide rows
    int tsum_direct;                 // 1 (no target longer than a tile, thin mode): (2, R, 2 T) instead - target t's part in the tile it
                                     // starts in at slot 2 t, its continuation in the next tile at 2 t + 1 (zero if there is none): the
                                     // reader needs no descriptor, i.e. one dependent memory round trip less at the head of k_param
    const int2* tdesc;               // (T): {slot of the target's first part = tile * tile_targets + index in the tile, number of parts};
                                     //      part i >= 1 is the first target of the i-th next tile: slot (tile + i) * tile_targets
};

// rows of the per-guide partials written by k_guide_tiling, (kTNumPart, G)
#ifndef BEAN_PARAM_BLOCK
#define BEAN_PARAM_BLOCK 256
#endif
#ifndef BEAN_KP_DIAG
#define BEAN_KP_DIAG 0
#endif
constexpr int kParamBlock = BEAN_PARAM_BLOCK;  // threads per block of k_param / k_target_reduce / k_q0_draws
constexpr int kLanesPerTarget = 16;  // k_param: lanes sharing one target: its (guide, replicate) rows and the 2 B bin edges of its Phi table
#ifndef BEAN_AMAX
#define BEAN_AMAX 8
#endif
constexpr int kAMax = BEAN_AMAX;  // alleles per guide the tiling kernels hold (8 in libbean_hip.so, 16 in libbean_hip_a16.so)
constexpr int kWaveMisc = 6;  // k_guide_wave: per-guide values staged in LDS (count totals, a0, allele counts)
#ifndef BEAN_BMAX
#define BEAN_BMAX 8
#endif
constexpr int kBMax = BEAN_BMAX;  // conditions whose counts a wave loads in one register batch: 8 in libbean_hip.so, 16 in libbean_hip_a16.so
// n_condits <= kBCap (bean_hip_create).  The default build stays at its batch (the fast path); the 16-condition
// build stages further conditions one by one (every per-condition value is a thread-private LDS column and the
// loops over conditions are rolled), up to 64 conditions - or what a workgroup's 160 KB of LDS hold (the launch asks
// for the attribute beyond 64 KB; bean_hip_prepare refuses a shape that needs more than the CU has).
constexpr int kBCap = BEAN_BMAX <= 8 ? 8 : 64;
constexpr size_t kLdsPerWorkgroupMax = 160 * 1024;
enum TPart { kTGnoise = 0, kTNrg = 1, kTPath = 2, kTL = 2 + kAMax, kTGmu = 2 + 2 * kAMax,
             kTGsig = 2 + 2 * kAMax + (kAMax - 1), kTNumPart = 2 + 2 * kAMax + 2 * (kAMax - 1) };

// Wave-uniform read of kernel-invariant data through the scalar cache (s_load): the constant
// address space tells the compiler that nothing in this launch writes the location.
__device__ __forceinline__ double uniform_ld(const double* p, int i) {
    typedef const double __attribute__((address_space(4))) * cptr;
    return ((cptr)(unsigned long long)p)[i];
}

// Layout of the per-allele-slot tables (tabP, tabPmu, tabPy: `b`-th plane of slot (a1, g)) and of mu_a / sig_a.
// The register-resident tiling kernels run one lane per GUIDE, so the guide index is the contiguous one:
// (B, A - 1, G) and (A - 1, G).  The allele-parallel kernels (bean_tiling_wide.hpp, DevArgs::wide_alleles) run
// one lane per ALLELE of one guide: there the allele index is contiguous - (B, G, A - 1) and (G, A - 1) - so
// that a wave's 64 lanes read consecutive doubles instead of 64 cache lines.
__device__ __forceinline__ long tab_off(const DevArgs& c, int b, int a1, long g) {
    const long A1 = c.A - 1;
    return c.wide_alleles ? ((long)b * c.G + g) * A1 + a1 : ((long)b * A1 + a1) * c.G + g;
}
__device__ __forceinline__ long slot_off(const DevArgs& c, int a1, long g) {
    return c.wide_alleles ? g * (long)(c.A - 1) + a1 : (long)a1 * c.G + g;
}

__device__ __forceinline__ int uniform_ld_i(const int* p, int i) {
    typedef const int __attribute__((address_space(4))) * cptr;
    return ((cptr)(unsigned long long)p)[i];
}

// The index that keys a guide's random streams: its position in the whole screen - guide_offset + g, or, where the
// caller handed the guides over in another order (tiling: by allele count), what it says it is.
__device__ __forceinline__ unsigned long long guide_stream_id(const DevArgs& c, int g) {
    return (unsigned long long)(c.gid ? c.gid[g] : c.g_off + g);
}

// ---------------------------------------------------------------- loss accumulation
// Every wave / block adds its part of the step's loss with INTEGER atomics: the part is split exactly
// into a multiple of 2^-10 and a remainder rounded to 2^-40, so the sum does not depend on the order
// in which the parts arrive and the loss history is bitwise reproducible (a float64 atomicAdd is not).
// Range: |part| < 2^33 (kLossPartMax), |loss| < 2^52; resolution 2^-40.  One address would serialise the chip's ~4000 waves at ~12 ns
// per atomic (tens of microseconds per launch - measured: it dominated the guide kernel), so a slot is
// kLossSub accumulators in separate 64-byte lines and a part goes to the one its block id selects;
// integer sums make the choice irrelevant for the result.  k_loss_finalize adds them up into the
// double in loss_hist.
constexpr int kLossWords = 8;   // int64 words per accumulator line: 2^-10 units, 2^-40 units, poison count
// Largest part a wave / block may add: its 2^-10-unit word is < 2^43, so 2^20 parts (a 13 M (replicate, guide)
// screen has that many waves) cannot wrap the int64 sum; a part beyond it - a diverging fit - poisons the slot
// and the reported loss is NaN (the host halts the fit at the next report), instead of a finite wrong value.
constexpr double kLossPartMax = 8589934592.0;  // 2^33
constexpr int kLossSub = 64;    // accumulator lines per loss_hist slot
__device__ __forceinline__ void fixed_add(long long* acc, double v) {
    if (!(fabs(v) < kLossPartMax)) {  // NaN / inf / out of range: poison the slot (k_loss_finalize reports NaN)
        atomicAdd((unsigned long long*)acc + 2, 1ull);
        return;
    }
    const double hi = rint(v * 1024.0);
    const double lo = rint((v - hi * (1.0 / 1024.0)) * 1099511627776.0);
    atomicAdd((unsigned long long*)acc, (unsigned long long)(long long)hi);
    atomicAdd((unsigned long long*)acc + 1, (unsigned long long)(long long)lo);
}
// sum of `n` accumulator lines as a double (NaN when poisoned)
__device__ __forceinline__ double fixed_value(const long long* acc, int n) {
    long long hi = 0, lo = 0, bad = 0;
    for (int i = 0; i < n; ++i) {
        hi += acc[i * kLossWords];
        lo += acc[i * kLossWords + 1];
        bad |= acc[i * kLossWords + 2];
    }
    if (bad != 0) return __builtin_nan("");
    return (double)hi * (1.0 / 1024.0) + (double)lo * (1.0 / 1099511627776.0);
}
__device__ __forceinline__ void loss_add(const DevArgs& c, unsigned long long slot, double v) {
#ifdef BEAN_NO_LOSS  // diagnostic builds: what do the loss atomics cost?
    return;
#endif
    const unsigned sub = (blockIdx.x + blockIdx.y * gridDim.x) & (kLossSub - 1);
    fixed_add(c.loss_acc + ((long)slot * kLossSub + sub) * kLossWords, v);
}

// A wave-form guide kernel leaves its part of the loss as three int64 words (the split of fixed_add)
// in lpart[wave]; the k_param launch that follows adds them up (integers: any grouping gives the same
// sum) and issues two atomics per BLOCK instead of two per wave - the ~4000 device-scope atomics of a
// guide launch cost 3.6 us of its 48.
__device__ __forceinline__ void wave_loss_out(const DevArgs& c, unsigned long long slot, long wave, double v) {
    if (!c.lpart) {
        loss_add(c, slot, v);
        return;
    }
    long long* o = c.lpart + 3 * wave;
    if (!(fabs(v) < kLossPartMax)) {
        o[0] = 0;
        o[1] = 0;
        o[2] = 1;
        return;
    }
    const double hi = rint(v * 1024.0);
    o[0] = (long long)hi;
    o[1] = (long long)rint((v - hi * (1.0 / 1024.0)) * 1099511627776.0);
    o[2] = 0;
}

// ---------------------------------------------------------------- reductions
// Wave-wide sums by DPP row shifts / broadcasts (the GFX9 scan: row_shr 1, 2, 4, 8, row_bcast 15, 31; the
// total ends in lane 63 and is read back as a scalar): twelve 32-bit DPP moves for a 64-bit value where
// the shuffle form needed twelve trips through the LDS crossbar (~1 000 cycles at the end of every wave).
// Fixed order; the result is valid in every lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_pair(int lo, int hi, int& olo, int& ohi) {
    olo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    ohi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    int lo, hi;
    dpp_pair<CTRL, ROW_MASK>((int)b, (int)(b >> 32), lo, hi);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_i64(long long b) {
    int lo, hi;
    dpp_pair<CTRL, ROW_MASK>((int)b, (int)(b >> 32), lo, hi);
    return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    v += dpp_i64<0x111, 0xf>(v);  // row_shr:1
    v += dpp_i64<0x112, 0xf>(v);  // row_shr:2
    v += dpp_i64<0x114, 0xf>(v);  // row_shr:4
    v += dpp_i64<0x118, 0xf>(v);  // row_shr:8
    v += dpp_i64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_i64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane((int)v, 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((long long)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0x111, 0xf>(v);
    v += dpp_f64<0x112, 0xf>(v);
    v += dpp_f64<0x114, 0xf>(v);
    v += dpp_f64<0x118, 0xf>(v);
    v += dpp_f64<0x142, 0xa>(v);
    v += dpp_f64<0x143, 0xc>(v);
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Sum over the block; result valid in thread 0.  `scratch` holds >= 16 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) tot += scratch[i];
    return tot;
}

// likelihood-gradient row q (kPGmu, kPGy or kPGnoise) of guide g: the fused kernels reduce
// over replicates in LDS, the split form leaves one value per replicate (fixed-order sum)
__device__ __forceinline__ double part_row(const DevArgs& c, int q, int g) {
    if (!c.wrow) return c.part[(long)q * c.G + g];
    double s = 0.0;
    for (int r = 0; r < c.R; ++r) s += c.wrow[((long)q * c.R + r) * c.G + g];
    return s;
}
// tiling: row q of guide g summed over the replicates in fixed order: from `part` (block form, or the
// wave form after k_sum_trow - measured faster than R strided reads per row inside k_param), or straight
// from the per-replicate rows (the allele-parallel path, which has no separate reduction launch)
__device__ __forceinline__ double trow_sum(const DevArgs& c, int q, long g) {
    if (!c.trow || c.trow_summed) return c.part[(long)q * c.G + g];
    double s = 0.0;
    if (c.wide_alleles) {
        // allele-parallel path: the rows of a (replicate, guide) are contiguous, (R, G, Q)
        const long Q = 2 + 2 * (long)c.A + 2 * ((long)c.A - 1);
        for (int r = 0; r < c.R; ++r) s += c.trow[((long)r * c.G + g) * Q + q];
        return s;
    }
    for (int r = 0; r < c.R; ++r) s += c.trow[((long)q * c.R + r) * c.G + g];
    return s;
}
__device__ __forceinline__ double lik_row(const DevArgs& c, int q, int g) {
    if (!c.rrow) return part_row(c, q, g);
    double s = 0.0;
    for (int r = 0; r < c.R; ++r) s += c.rrow[((long)q * c.R + r) * c.G + g];
    return s;
}

// ------------------------------------------------------------- ClippedAdam
// pyro.optim.ClippedAdam on one float32 element.
struct AdamCoef {
    float step_size, clip;
};
__device__ __forceinline__ AdamCoef adam_coef(const DevArgs& c, unsigned long long t) {
    const double td = (double)t;
    const double lr = c.lr0 * exp(td * c.log_lrd);
    const double bc1 = 1.0 - pow(0.9, td), bc2 = 1.0 - pow(0.999, td);
    AdamCoef k;
    k.step_size = (float)(lr * sqrt(bc2) / bc1);
    k.clip = (float)c.clip;
    return k;
}
// The guide kernel of step s hands the step counter back to k_param (ctrA) together with the
// ClippedAdam step size of the update that follows: exp + two pow in float64 are ~500 instructions,
// which every k_param thread used to evaluate for itself at the head of its dependency chain; one
// lane of the (long) guide kernel computes them instead.
__device__ __forceinline__ void publish_ctr(const DevArgs& c, StepCtr ctr) {
    ctr.step_size = adam_coef(c, ctr.step + 1).step_size;
    *c.ctrA = ctr;
}

__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float grad, AdamCoef k) {
    // every rounding is pinned (no implicit contraction) so that the stand-alone
    // k_adam and the fused update inside k_param produce identical bits
#pragma clang fp contract(off)
    const float gc = fminf(fmaxf(grad, -k.clip), k.clip);
    const float m9 = m * 0.9f;
    m = fmaf(gc, 0.1f, m9);            // exp_avg.mul_(b1).add_(grad, alpha=1-b1)
    const float v9 = v * 0.999f;
    const float g2 = gc * gc;
    v = fmaf(g2, 0.001f, v9);          // exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2)
    const float denom = sqrtf(v) + 1e-8f;
    const float q = m / denom;
    p = fmaf(-k.step_size, q, p);      // p.addcdiv_(exp_avg, denom, value=-step_size)
}

// Accesses to state that ANOTHER wave of the same launch wrote at an earlier step or will read at a later one
// (k_svi_async, bean_async_v2.hpp: one launch runs many steps, and the wave that finishes a tile differs from step
// to step): COH == 2 makes them agent-scope (global_load / global_store ... sc1: the store is written through,
// the load bypasses this CU's L1, which no other CU's store ever refreshes).  Any other mode: plain.
template <int COH, typename T>
__device__ __forceinline__ T coh_ld(const T* p) {
    if (COH == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <int COH, typename T>
__device__ __forceinline__ void coh_st(T* p, T v) {
    if (COH == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

template <bool ADAM, int COH = 0>
__device__ __forceinline__ void emit_grad(const DevArgs& c, int which, long idx, double grad, AdamCoef k) {
    const float gf = (float)grad;
    if (ADAM) {
        float p = coh_ld<COH>(c.p[which] + idx), m = coh_ld<COH>(c.m[which] + idx), v = coh_ld<COH>(c.v[which] + idx);
        adam_update(p, m, v, gf, k);
        coh_st<COH>(c.p[which] + idx, p);
        coh_st<COH>(c.m[which] + idx, m);
        coh_st<COH>(c.v[which] + idx, v);
    } else {
        c.g[which][idx] = gf;
    }
}

// Same with the parameter and its moments already in registers (loaded before the gradient sums,
// so that no load sits behind them); the updated parameter stays in `p` for the next draw.
template <bool ADAM, int COH = 0>
__device__ __forceinline__ void emit_grad_pre(const DevArgs& c, int which, long idx, double grad, AdamCoef k,
                                              float& p, float m, float v) {
    const float gf = (float)grad;
    if (ADAM) {
        adam_update(p, m, v, gf, k);
        coh_st<COH>(c.p[which] + idx, p);
        coh_st<COH>(c.m[which] + idx, m);
        coh_st<COH>(c.v[which] + idx, v);
    } else {
        c.g[which][idx] = gf;
    }
}

// ---- per-target scalar math of the sorting families' FINISH / PREP.  Shared by k_param and by the
// fused step kernel (bean_step_v2.hpp), whose lanes evaluate one parameter each: every rounding is
// pinned (no implicit contraction), so both produce the same bits.
struct TgtPrior {
    double l0, var0, logs0;
};
__device__ __forceinline__ TgtPrior tgt_sd_prior(const DevArgs& c, int t) {
    TgtPrior pr;
    pr.l0 = c.pr_sd_loc ? c.pr_sd_loc[t] : 0.0;
    // LogNormal(sd_loc, sd_scale) prior: the default scale lives in a float32
    // tensor in the reference (model.py:405-406), so torch forms scale**2 and
    // log(scale) in float32; user-supplied --prior-params are taken as float64
    if (c.pr_sd_scale) {
        const double s0 = c.pr_sd_scale[t];
        pr.var0 = s0 * s0;
        pr.logs0 = log(s0);
    } else {
        const float s0f = (float)c.sd_prior_scale;
        pr.var0 = (double)(s0f * s0f);
        pr.logs0 = (double)logf(s0f);
    }
    return pr;
}
// d log p / d mu and d log p / d y of the priors (Laplace(0, 1) or Normal on mu, LogNormal on sd), and
// - log p + log q of the target's two sites for the reported loss (p1, p3: log of the guide scales)
__device__ __forceinline__ void tgt_prior_terms(const DevArgs& c, int t, const TgtPrior& pr, double mu, double y,
                                                double eps1, double eps2, float p1, float p3, double& dlogp_mu,
                                                double& dlogp_dy, double& loss) {
#pragma clang fp contract(off)
    double logp_mu;
    if (c.flags & kPriorNormalMu) {
        const double pl = c.pr_mu_loc ? c.pr_mu_loc[t] : 0.0;
        const double ps = c.pr_mu_scale ? c.pr_mu_scale[t] : 1.0;
        const double zz = (mu - pl) / ps;
        logp_mu = -0.5 * zz * zz - log(ps) - kHalfLog2PiC;
        dlogp_mu = -zz / ps;
    } else {
        logp_mu = -kLog2 - fabs(mu);
        dlogp_mu = mu > 0.0 ? -1.0 : (mu < 0.0 ? 1.0 : 0.0);
    }
    const double dy0 = y - pr.l0;
    const double logp_sd = -y - pr.logs0 - kHalfLog2PiC - dy0 * dy0 / (2.0 * pr.var0);
    dlogp_dy = -1.0 - dy0 / pr.var0;
    const double logq_mu = -0.5 * eps1 * eps1 - (double)p1 - kHalfLog2PiC;
    const double logq_sd = -y - 0.5 * eps2 * eps2 - (double)p3 - kHalfLog2PiC;
    loss = -logp_mu - logp_sd + logq_mu + logq_sd;
}
// d loss / d (unconstrained parameter j) of a target: j = 0 mu_loc, 1 mu_scale, 2 sd_loc, 3 sd_scale;
// G = d loss / d (drawn mu or y), eps the draw's standard normal, s = exp(parameter j) for the scales
__device__ __forceinline__ double tgt_grad(int j, double G, double eps, double s) {
#pragma clang fp contract(off)
    if (j == 0) return G;
    if (j == 2) return G - 1.0;
    const double ges = G * eps * s;
    if (j == 1) return ges - 1.0;
    return ges - 1.0 - eps * s;
}
// the reparameterised draw loc + eps * exp(log scale)
__device__ __forceinline__ double tgt_draw(float loc, double eps, float log_scale) {
#pragma clang fp contract(off)
    return (double)loc + eps * exp((double)log_scale);
}
// One bin edge of a target's Phi tables (a2): even e = upper edge, odd e = lower edge of bin e >> 1,
// on adjacent lanes; the pair is combined with one shuffle.  mu_r: the target's drawn mean (plus the
// replicate's covariate shift), inv = 1 / sigma.
__device__ __forceinline__ void phi_edge(const DevArgs& c, int t, bool live_t, int e, double mu_r, double inv,
                                         double dsig_dy, long off) {
#pragma clang fp contract(off)
    const int b = e >> 1;
    const bool upper = (e & 1) == 0, live = live_t && b < c.B;
    const double z = live ? (upper ? c.z_hi[b] : c.z_lo[b]) : 0.0;
    double cdf = upper ? 1.0 : 0.0, pdf = 0.0, upd = 0.0;
    if (live && !isinf(z)) {
        const double u = (z - mu_r) * inv;
        cdf = norm_cdf(u);
        pdf = norm_pdf(u);
        upd = u * pdf;
    }
    // the other edge of the bin sits on the neighbouring lane: quad_perm [1, 0, 3, 2]
    const double cl = dpp_f64<0xB1, 0xf>(cdf), fl = dpp_f64<0xB1, 0xf>(pdf), ufl = dpp_f64<0xB1, 0xf>(upd);
    if (live && upper) {
        const long o = off + (long)b * c.T + t;
        c.tabP[o] = cdf - cl;
        c.tabPmu[o] = -(pdf - fl) * inv;
        c.tabPy[o] = -(upd - ufl) * inv * dsig_dy;
    }
}

// --------------------------------------------------- tiling: per-guide finish
// Guide part of k_param for MultiMixtureNormal: Dirichlet normalisers of the
// A-component pi site, chain to alpha_pi through the two concentration maps the
// reference uses (guide: alpha/sum * pi_a0, model.py:938; model:
// (alpha + eps/A)/(sum + eps) * pi_a0 floored at eps, model.py:646-651).
// sum over the kAMax lanes that share one guide (fixed shuffle tree)
__device__ __forceinline__ double allele_group_sum(double v) {
#pragma unroll
    for (int off = kAMax / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kAMax);
    return v;
}

// Guide part of k_param for the tiling families: kAMax lanes per guide, lane a owns allele a (its
// two lgamma/digamma pairs and its alpha_pi update); lane 0 also owns the guide's noise parameters.
template <bool FINISH, bool ADAM, bool PREP>
__device__ __forceinline__ void param_guide_tiling(const DevArgs& c, int guide_block,
                                                   unsigned long long s_prep, AdamCoef ak,
                                                   double& loss_fin) {
    const int tid = guide_block * blockDim.x + threadIdx.x;
    const int g = tid / kAMax, a = tid % kAMax;
    const bool acc_on = (c.flags & kAcc) != 0;
    const bool fit_noise = acc_on && (c.flags & kFitNoise);
    const bool in = g < c.G;
    const int A = c.A;
    const bool lead = in && a == 0;
    float nl = 0.f, ns_u = 0.f;
    if (lead && fit_noise) {
        nl = c.p[5][g];
        ns_u = c.p[6][g];
    }
    if (FINISH) {
        const bool live = in && a < A;
        const bool am = live && c.amask[(long)g * A + a] != 0;
        const double alpha = live ? (am ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
        const double S = allele_group_sum(alpha);
        const double pa0 = in ? c.pi_a0[g] : 1.0;
        const double rS = frcp(in ? S : 1.0), rSe = frcp((in ? S : 1.0) + kEps);
        // the survival tiling guide clamps its concentration at 1e-5 (survival_model.py:813-821);
        // the sorting one does not (model.py:942-950)
        const bool clampq = c.survival != 0;
        const double nrg = in ? trow_sum(c, kTNrg, g) : 0.0;
        const double cqr = alpha * rS * pa0;
        const bool cqc = clampq && cqr < 1e-5;
        const double cq = cqc ? 1e-5 : cqr;
        const double cpr = (alpha + kEps / A) * rSe * pa0;
        const bool cpc = cpr < kEps;
        const double cp = cpc ? kEps : cpr;
        const double sq = allele_group_sum(live ? cq : 0.0), sp = allele_group_sum(live ? cp : 0.0);
        double lgS_q = 0.0, dgS_q = 0.0, lgS_p = 0.0, dgS_p = 0.0, gq = 0.0, gp = 0.0;
        if (in) {
            lgamma_digamma(sq, lgS_q, dgS_q);
            lgamma_digamma(sp, lgS_p, dgS_p);
        }
        if (live) {
            double lg, dg;
            lgamma_digamma(cq, lg, dg);
            const double L = trow_sum(c, kTL + a, g);
            const double lq = -nrg * lg + (cq - 1.0) * L;
            gq = cqc ? 0.0 : L + nrg * (dgS_q - dg) + trow_sum(c, kTPath + a, g);
            lgamma_digamma(cp, lg, dg);
            const double lp = -nrg * lg + (cp - 1.0) * L;
            gp = cpc ? 0.0 : -(L + nrg * (dgS_p - dg));
            loss_fin = lq - lp;
        }
        if (lead) loss_fin += nrg * (lgS_q - lgS_p);
        const double dq = allele_group_sum(gq * alpha) * rS * rS;
        const double dp = allele_group_sum(live ? gp * (alpha + kEps / A) : 0.0) * rSe * rSe;
        if (live) {
            const double ga = pa0 * (gq * rS - dq + gp * rSe - dp);
            emit_grad<ADAM>(c, 4, (long)g * A + a, am ? ga * alpha : 0.0, ak);
        }
        if (lead && acc_on) {
            const double lpn = c.lpn[g], eps = c.eps_noise[g];
            const double gl = trow_sum(c, kTGnoise, g);
            const double ns = fit_noise ? exp((double)ns_u) : kPiNoiseSd;
            const float nsf = 0.655f;
            const double nvar = (double)(nsf * nsf);
            const double logp = -lpn * lpn / (2.0 * nvar) - (double)logf(nsf) - kHalfLog2PiC;
            const double logq = -0.5 * eps * eps - log(ns) - kHalfLog2PiC;
            loss_fin += -logp + logq;
            if (fit_noise) {
                const double Gl = gl + lpn / nvar;
                emit_grad<ADAM>(c, 5, g, Gl, ak);
                emit_grad<ADAM>(c, 6, g, Gl * eps * ns - 1.0, ak);
                if (ADAM) {
                    nl = c.p[5][g];
                    ns_u = c.p[6][g];
                }
            }
        }
    }
    if (PREP && acc_on && lead) {
        double eps;
        if (c.eps_noise_in) {
            eps = c.eps_noise_in[g];
        } else {
            eps = (double)normal2_at(c.seed, ((unsigned long long)kSiteNoise << 48) + guide_stream_id(c, g),
                                         s_prep * 4ull).x;
        }
        const double ns = fit_noise ? exp((double)ns_u) : kPiNoiseSd;
        c.eps_noise[g] = eps;
        c.lpn[g] = (fit_noise ? (double)nl : 0.0) + eps * ns;
        if (c.eps_noise_out) c.eps_noise_out[g] = eps;
    }
    if (PREP && c.dgq_t) {
        // digamma of the guide-side concentrations of the (updated) alpha_pi for the next guide kernel's
        // A implicit-gradient calls per (replicate, guide), which otherwise each evaluate two digammas
        const bool live = in && a < A;
        const bool am = live && c.amask[(long)g * A + a] != 0;
        const double alpha = live ? (am ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
        const double S = allele_group_sum(alpha);
        const double pa0 = in ? c.pi_a0[g] : 1.0;
        double cq = alpha * frcp(in ? S : 1.0) * pa0;
        if (c.survival && live && cq < 1e-5) cq = 1e-5;
        const double tot = allele_group_sum(live ? cq : 0.0);
        if (live) c.dgq_t[(long)a * c.G + g] = digamma(cq);
        if (lead) c.dgq_t[(long)kAMax * c.G + g] = digamma(tot);
    }
}

// One entry of the per-target Phi tables (a2): P[b, t] = Phi(u_hi) - Phi(u_lo) and its
// derivatives in mu_t and y_t = log sd_t, for the draw (mu, y) of target t.
__device__ __forceinline__ void write_phi_entry(const DevArgs& c, int t, int b, double mu, double y, long off = 0) {
    // NormalModel uses sqrt(sd) as the scale (model.py:92-98)
    const double sigma = c.family == kNormal ? exp(0.5 * y) : exp(y);
    const double dsig_dy = c.family == kNormal ? 0.5 * sigma : sigma;
    const double inv = 1.0 / sigma;
    const double zh = c.z_hi[b], zl = c.z_lo[b];
    double ch = 1.0, cl = 0.0, fh = 0.0, fl = 0.0, ufh = 0.0, ufl = 0.0;
    if (!isinf(zh)) {
        const double u = (zh - mu) * inv;
        ch = norm_cdf(u);
        fh = norm_pdf(u);
        ufh = u * fh;
    }
    if (!isinf(zl)) {
        const double u = (zl - mu) * inv;
        cl = norm_cdf(u);
        fl = norm_pdf(u);
        ufl = u * fl;
    }
    const long o = off + (long)b * c.T + t;
    c.tabP[o] = ch - cl;
    c.tabPmu[o] = -(fh - fl) * inv;
    c.tabPy[o] = -(ufh - ufl) * inv * dsig_dy;
}

// Thin mode (the usual one): a target block of kParamBlock threads holds kParamBlock / DevArgs::lpt targets
// and works in three phases with two thread -> work maps:
//   A  sums      lpt lanes per target (group map): the target's (guide, replicate) rows
//   B  scalar    ONE lane per target, packed into the block's first lanes (owner map): priors,
//                entropies, ClippedAdam, the next draw.  Packing matters: this is a long serial chain,
//                and a wave issues it whether one or all of its lanes are active
//   C  tables    group map again: one lane per bin edge
// with the hand-over through LDS.  Wide mode (few or very long targets): one target per block.
// The survival families and the tiling families have no table to fill and few rows per target (~15
// (guide, replicate) rows; the handful of alleles that carry an edit): with kLanesPerTargetNarrow lanes
// per target a block takes 64 targets instead of 16.  BASELINE config 5: 313 target blocks instead of
// 1 250, and k_param's three kinds of blocks are resident together; config 3: k_param 32 -> 27 us.
constexpr int kLanesPerTargetNarrow = 4;
constexpr int kTargetsPerBlockMax = kParamBlock / kLanesPerTargetNarrow;

// owner map: the target whose parameters this thread updates (phase B)
__device__ __forceinline__ void target_of_thread(const DevArgs& c, int& t, bool& active, unsigned bid) {
    if (c.wide_targets) {
        t = bid;
        active = threadIdx.x == 0;
    } else {
        const int tpb = kParamBlock / c.lpt;
        t = bid * tpb + threadIdx.x;
        active = (int)threadIdx.x < tpb && t < c.T;
    }
}
// group map: the target this thread's lane group sums rows / tabulates edges for (phases A and C)
__device__ __forceinline__ void target_of_group(const DevArgs& c, int& t, bool& lead, unsigned bid) {
    if (c.wide_targets) {
        t = bid;
        lead = threadIdx.x == 0;
    } else {
        t = (bid * blockDim.x + threadIdx.x) / c.lpt;
        lead = t < c.T && (threadIdx.x & (c.lpt - 1)) == 0;
    }
}

// Sum of two values over each aligned group of 16 lanes, in every lane: the xor tree 8, 4, 2, 1 as DPP
// row rotations (a row IS 16 lanes; after the step with offset 8 the values repeat with period 8, after 4
// with period 4, ..., so rotating by the offset reads the same operand the xor partner holds: same
// additions, same bits as the shuffle tree, without the LDS crossbar).
static_assert(kLanesPerTarget == 16, "group16_allsum: one DPP row per target");
__device__ __forceinline__ void group16_allsum(double& a, double& b) {
    a += dpp_f64<0x128, 0xf>(a);  // row_ror:8
    b += dpp_f64<0x128, 0xf>(b);
    a += dpp_f64<0x124, 0xf>(a);  // row_ror:4
    b += dpp_f64<0x124, 0xf>(b);
    a += dpp_f64<0x122, 0xf>(a);  // row_ror:2
    b += dpp_f64<0x122, 0xf>(b);
    a += dpp_f64<0x121, 0xf>(a);  // row_ror:1
    b += dpp_f64<0x121, 0xf>(b);
}

// ... or of kLanesPerTargetNarrow = 4 lanes: xor 2, xor 1 as quad permutations
__device__ __forceinline__ void group_allsum(int lpt, double& a, double& b) {
    static_assert(kLanesPerTargetNarrow == 4, "one DPP quad per target");
    if (lpt == kLanesPerTarget) {
        group16_allsum(a, b);
    } else {
        a += dpp_f64<0x4e, 0xf>(a);  // quad_perm:[2,3,0,1]
        b += dpp_f64<0x4e, 0xf>(b);
        a += dpp_f64<0xb1, 0xf>(a);  // quad_perm:[1,0,3,2]
        b += dpp_f64<0xb1, 0xf>(b);
    }
}

// Likelihood gradient of one target w.r.t. its drawn mu_t / y_t: the guide -> target segmented
// sum (a8), in a fixed order.  Valid in the `active` thread.
__device__ __forceinline__ void target_grad_sums(const DevArgs& c, int t, bool active, double* scratch,
                                                 double& gmu, double& gy) {
    gmu = 0.0;
    gy = 0.0;
    if (c.wide_targets && c.tsum) {
        // few or very long targets, one per block: entry i = part * R + r of the target's partial sums
        const int2 dsc = c.tdesc[t];
        const int R = c.R, n = dsc.y * R, ntm = c.tile_targets;
        const long S = (long)c.n_tiles * ntm;
        const int tile0 = dsc.x / ntm;
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int part = i / R, r = i - part * R;
            const long o = (long)r * S + (part == 0 ? dsc.x : (tile0 + part) * ntm);
            a += c.tsum[o];
            b += c.tsum[(long)R * S + o];
        }
        gmu = block_sum(a, scratch);
        gy = block_sum(b, scratch);
        return;
    }
    if (c.wide_targets) {
        const int g0 = c.toff[t], g1 = c.toff[t + 1];
        double a = 0.0, b = 0.0;
        for (int g = g0 + threadIdx.x; g < g1; g += blockDim.x) {
            a += lik_row(c, kPGmu, g);
            b += lik_row(c, kPGy, g);
        }
        gmu = block_sum(a, scratch);
        gy = block_sum(b, scratch);
        return;
    }
    if (c.family == kMultiMixture) {
        // edit <- alleles containing it (transposed CSR): the backward of
        // allele_to_edit @ mu_edits and ||allele_to_edit * sd_edits|| (model.py:618-622).  The
        // slots of the edit are spread over its lane group, summed by a fixed shuffle tree.
        const int A1 = c.A - 1;
        // rows of the per-allele-slot gradients: fixed layout of the register-resident kernels, or the
        // runtime layout of the wide path (bean_tiling_wide.hpp: 2 + 2 A, 2 + 2 A + (A - 1))
        const int q_gmu = c.wide_alleles ? 2 + 2 * c.A : (int)kTGmu;
        const int q_gsig = c.wide_alleles ? 2 + 2 * c.A + A1 : (int)kTGsig;
        const int lpt = c.lpt;
        const int lg = threadIdx.x & (lpt - 1);
        double a = 0.0, b = 0.0;
        if (t < c.T) {
            const double sd = c.survival ? 0.0 : exp(c.y_t[t]);  // survival: no sd latent
            // four slots of the lane at a time: their indices asked for together, then their rows together, then the
            // additions in the order of the plain loop (same bits).  One slot per pass was two dependent round trips per
            // slot - index, then rows - and an edit's lane has three or four slots: this loop is the head of the launch's
            // longest chain (edit blocks -> allele blocks)
            const int k1 = c.e2a_ptr[t + 1];
            for (int kb = c.e2a_ptr[t] + lg; kb < k1; kb += 4 * lpt) {
                int sl[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = kb + u * lpt;
                    sl[u] = k < k1 ? c.e2a_idx[k] : -1;
                }
                double xa[4], xb[4], sg[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    xa[u] = 0.0;
                    xb[u] = 0.0;
                    sg[u] = 1.0;
                    if (sl[u] >= 0) {
                        const int a1 = sl[u] % A1, gs = sl[u] / A1;
                        xa[u] = trow_sum(c, q_gmu + a1, gs);
                        if (!c.survival) {
                            xb[u] = trow_sum(c, q_gsig + a1, gs);
                            sg[u] = c.sig_a[slot_off(c, a1, gs)];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (sl[u] >= 0) {
                        a += xa[u];
                        // d sigma_a / d y_e = sd_e^2 / sigma_a
                        if (!c.survival) b += xb[u] * sd * sd / sg[u];
                    }
            }
        }
        group_allsum(lpt, a, b);
        gmu = a;
        gy = b;
        return;
    }
    if (active) {
        if (!c.wrow) {
            const int g0 = c.toff[t], g1 = c.toff[t + 1];
            for (int g = g0; g < g1; ++g) {
                gmu += lik_row(c, kPGmu, g);
                gy += lik_row(c, kPGy, g);
            }
        }
    }
    if (c.wrow && c.tsum) {
        // k_guide_wave2 has summed the target's guides inside its waves: entry i = part * R + r of the
        // target's (part, replicate) sums - R of them, 2 R where the target straddles two tiles - goes to lane
        // i mod lpt of its group; fixed shuffle tree.  A target is cut at multiples of 64 of the GLOBAL guide
        // index: the same parts, the same bits, whatever the shard.
        const int lpt = c.lpt;
        const int lg = threadIdx.x & (lpt - 1);
        double a = 0.0, b = 0.0;
        if (t < c.T && c.tsum_direct) {
            // both slots of the target, 2 R entries: part-major, as below (an absent continuation holds zeros)
            const int R = c.R, n = 2 * R;
            const long S = 2 * (long)c.T;
            for (int i = lg; i < n; i += lpt) {
                const int part = i >= R ? 1 : 0, r = i - part * R;
                const long o = (long)r * S + 2 * (long)t + part;
                a += c.tsum[o];
                b += c.tsum[(long)R * S + o];
            }
        } else if (t < c.T) {
            const int2 dsc = c.tdesc[t];
            const int R = c.R, n = dsc.y * R, ntm = c.tile_targets;
            const long S = (long)c.n_tiles * ntm;
            const float rR = 1.0f / (float)R;  // i / R below: exact for i < 2^20
            for (int i = lg; i < n; i += lpt) {
                const int part = (int)(((float)i + 0.5f) * rR), r = i - part * R;
                const long o = (long)r * S + (part == 0 ? dsc.x : (dsc.x / ntm + part) * ntm);
                a += c.tsum[o];
                b += c.tsum[(long)R * S + o];
            }
        }
        group_allsum(lpt, a, b);
        gmu = a;
        gy = b;
    } else if (c.wrow) {
        // wave form: the (guide, replicate) rows of the target are spread over its lane
        // group and summed by a fixed shuffle tree (deterministic, shard independent)
        const int lpt = c.lpt;
        const int lg = threadIdx.x & (lpt - 1);
        double a = 0.0, b = 0.0;
        if (t < c.T) {
            const int g0 = c.toff[t], ng = c.toff[t + 1] - g0;
            const int n = ng * c.R;
            // four entries' loads in flight per lane before the first add (a loop of load / add pairs is
            // a chain of memory round trips, and k_param is made of those); i / ng without the integer
            // division (exact for i < 2^20); the order of the additions is unchanged
            const float rng = 1.0f / (float)ng;
            for (int i0 = lg; i0 < n; i0 += 4 * lpt) {
                double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * lpt;
                    xa[u] = xb[u] = 0.0;
                    if (i < n) {
                        const int r = (int)(((float)i + 0.5f) * rng), g = g0 + (i - r * ng);
                        xa[u] = c.wrow[((long)kPGmu * c.R + r) * c.G + g];
                        if (!c.survival) xb[u] = c.wrow[((long)kPGy * c.R + r) * c.G + g];  // (survival: no sd latent)
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * lpt < n) {
                        a += xa[u];
                        b += xb[u];
                    }
            }
        }
        group_allsum(lpt, a, b);
        gmu = a;
        gy = b;
    }
}

// Sharded runs of families whose per-target parameters are replicated on every rank
// (ControlNormal, tiling per-edit parameters): this rank's part of every target's likelihood
// gradient, (2, T) doubles, for the host to all-reduce before k_param.
__global__ __launch_bounds__(kParamBlock) void k_target_reduce(DevArgs c, double* out) {
    __shared__ double scratch[16];
    int t;
    bool lead;
    target_of_group(c, t, lead, blockIdx.x);
    double gmu, gy;
    target_grad_sums(c, t, lead, scratch, gmu, gy);
    if (lead) {
        out[t] = gmu;
        out[c.T + t] = gy;
    }
}

}  // namespace bean
#include "bean_tiling_wide.hpp"  // needs everything above; k_param below dispatches to it
namespace bean {

// Coherent read of a row another wave of the SAME launch has written (fused step kernel): agent-scope
// atomic load (global_load ... sc1); k_param reads the rows of the previous launch with plain loads.
template <bool COH>
__device__ __forceinline__ double row_ld(const double* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
__device__ __forceinline__ double noise_row_coherent(const DevArgs& c, int g) {
    double s = 0.0;
    for (int r = 0; r < c.R; ++r) s += row_ld<true>(c.wrow + ((long)kPGnoise * c.R + r) * c.G + g);
    return s;
}

// Guide part of k_param for the variant MixtureNormal families, one guide per lane: Dirichlet
// normalisers, chain to alpha_pi, ClippedAdam, (+Acc: the noise site), and the lgamma / digamma table of
// the updated concentrations.  Shared with the fused step kernel (COH: see row_ld); roundings pinned.
// COH: 0 = rows of the previous launch (plain loads); 1 = rows another wave of this launch has written (fused step
// kernel: agent-scope loads); 2 = k_svi_async: as 1, and the guide's own state - alpha_pi, the noise site, their
// moments, the tabulated digammas, the draw - is read and written at agent scope too (coh_ld / coh_st).
template <bool FINISH, bool ADAM, bool PREP, int COH>
__device__ __forceinline__ void param_guide_mix(const DevArgs& c, int g, AdamCoef ak, unsigned long long s_prep,
                                                double& loss_fin) {
#pragma clang fp contract(off)
    const bool acc_on = (c.flags & kAcc) != 0;
    const bool fit_noise = acc_on && (c.flags & kFitNoise);
    float nl = 0.f, ns_u = 0.f;
    if (fit_noise) {
        nl = coh_ld<COH>(c.p[5] + g);
        ns_u = coh_ld<COH>(c.p[6] + g);
    }
    // alpha_pi: parameters and moments loaded once, with everything else the guide needs (k_param is a
    // chain of memory round trips); PREP takes the updated values from registers
    float up[2] = {coh_ld<COH>(c.p[4] + 2 * g), coh_ld<COH>(c.p[4] + 2 * g + 1)}, um[2] = {0.f, 0.f}, uv[2] = {0.f, 0.f};
    if (FINISH && ADAM) {
        um[0] = coh_ld<COH>(c.m[4] + 2 * g);
        um[1] = coh_ld<COH>(c.m[4] + 2 * g + 1);
        uv[0] = coh_ld<COH>(c.v[4] + 2 * g);
        uv[1] = coh_ld<COH>(c.v[4] + 2 * g + 1);
    }
    if (FINISH) {
        const float u0 = up[0], u1 = up[1];
        const double al0 = (double)expf(u0), al1 = (double)expf(u1);
        const double s = al0 + al1, pa0 = c.pi_a0[g];
        const double cp[2] = {al0 / s * pa0, al1 / s * pa0};
        const bool cl[2] = {cp[0] < 1e-5, cp[1] < 1e-5};
        const double cq[2] = {cl[0] ? 1e-5 : cp[0], cl[1] ? 1e-5 : cp[1]};
        double lgS_p, dgS_p, lg_p[2], dg_p[2];
        double lgS_q, dgS_q, lg_q[2], dg_q[2];
        if (c.dgq) {
            // the guide side (c_q) was tabulated by the previous PREP for this alpha_pi
            const long Gl = c.G;
            lgS_q = coh_ld<COH>(c.dgq + g);
            lg_q[0] = coh_ld<COH>(c.dgq + Gl + g);
            lg_q[1] = coh_ld<COH>(c.dgq + 2 * Gl + g);
            dgS_q = coh_ld<COH>(c.dgq + 3 * Gl + g);
            dg_q[0] = coh_ld<COH>(c.dgq + 4 * Gl + g);
            dg_q[1] = coh_ld<COH>(c.dgq + 5 * Gl + g);
            lgS_p = lgS_q, dgS_p = dgS_q, lg_p[0] = lg_q[0], lg_p[1] = lg_q[1], dg_p[0] = dg_q[0], dg_p[1] = dg_q[1];
            if (cl[0] || cl[1]) {  // model side (c_p, unclamped) differs
                lgamma_digamma(cp[0] + cp[1], lgS_p, dgS_p);
                lgamma_digamma(cp[0], lg_p[0], dg_p[0]);
                lgamma_digamma(cp[1], lg_p[1], dg_p[1]);
            }
        } else {
            lgamma_digamma(cp[0] + cp[1], lgS_p, dgS_p);
            lgamma_digamma(cp[0], lg_p[0], dg_p[0]);
            lgamma_digamma(cp[1], lg_p[1], dg_p[1]);
            lgS_q = lgS_p, dgS_q = dgS_p, lg_q[0] = lg_p[0], lg_q[1] = lg_p[1], dg_q[0] = dg_p[0], dg_q[1] = dg_p[1];
            if (cl[0] || cl[1]) {
                lgamma_digamma(cq[0] + cq[1], lgS_q, dgS_q);
                lgamma_digamma(cq[0], lg_q[0], dg_q[0]);
                lgamma_digamma(cq[1], lg_q[1], dg_q[1]);
            }
        }
        // per-replicate rows of the wave form: independent loads, four replicates in flight
        double Lp_[2] = {0.0, 0.0}, Lq_[2] = {0.0, 0.0}, path_[2] = {0.0, 0.0}, nrg = 0.0;
        double GA_[2] = {0.0, 0.0};
        if (c.wrow && c.rows_v2) {
            // k_guide_wave2 rows: GA_a = sum_r of everything d loss / d c_a owes to the draws;
            // the (c - 1) log pi terms of the loss were added by the guide kernel
            const long RG = (long)c.R * c.G;
            const double* w = c.wrow + g;
            nrg = c.part[(long)kPNrg * c.G + g];  // data only (k_prepare)
            if (COH != 0) {
                // atomic loads stay in program order and are waited for where they are used: load
                // eight replicates' rows first, add afterwards (same order of additions)
                for (int r0 = 0; r0 < c.R; r0 += 8) {
                    double x0[8], x1[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        x0[u] = x1[u] = 0.0;
                        if (r0 + u < c.R) {
                            const double* wr = w + (long)(r0 + u) * c.G;
                            x0[u] = row_ld<true>(wr + 3 * RG);
                            x1[u] = row_ld<true>(wr + 4 * RG);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (r0 + u < c.R) {
                            GA_[0] += x0[u];
                            GA_[1] += x1[u];
                        }
                }
            } else {
#pragma unroll 4
                for (int r = 0; r < c.R; ++r) {
                    const double* wr = w + (long)r * c.G;
                    GA_[0] += wr[3 * RG];
                    GA_[1] += wr[4 * RG];
                }
            }
        } else if (c.wrow) {
            const long RG = (long)c.R * c.G;
            const double* w = c.wrow + g;
            nrg = c.part[(long)kPNrg * c.G + g];  // data only (k_prepare)
#pragma unroll 4
            for (int r = 0; r < c.R; ++r) {
                const double* wr = w + (long)r * c.G;
                Lp_[0] += wr[kPLp * RG];
                Lp_[1] += wr[(kPLp + 1) * RG];
                Lq_[0] += wr[kPLq * RG];
                Lq_[1] += wr[(kPLq + 1) * RG];
                path_[0] += wr[kPPath * RG];
                path_[1] += wr[(kPPath + 1) * RG];
            }
        } else {
            nrg = c.part[(long)kPNrg * c.G + g];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                Lp_[a] = c.part[(long)(kPLp + a) * c.G + g];
                Lq_[a] = c.part[(long)(kPLq + a) * c.G + g];
                path_[a] = c.part[(long)(kPPath + a) * c.G + g];
            }
        }
        const double Rf = (double)c.R;
        double gc[2];
        double lp = nrg * (lgS_p - lg_p[0] - lg_p[1]);
        double lq = Rf * (lgS_q - lg_q[0] - lg_q[1]);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const double Lp = Lp_[a], Lq = Lq_[a], gpath = path_[a];
            lp += (cp[a] - 1.0) * Lp;
            lq += (cq[a] - 1.0) * Lq;
            const double g_cp = -(Lp + nrg * (dgS_p - dg_p[a]));
            const double g_cq = cl[a] ? 0.0 : (Lq + Rf * (dgS_q - dg_q[a]) + gpath);
            gc[a] = g_cp + g_cq;
        }
        if (c.wrow && c.rows_v2) {
            lp = nrg * (lgS_p - lg_p[0] - lg_p[1]);
            lq = Rf * (lgS_q - lg_q[0] - lg_q[1]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
                gc[a] = -nrg * (dgS_p - dg_p[a]) + (cl[a] ? 0.0 : Rf * (dgS_q - dg_q[a])) + GA_[a];
        }
        loss_fin += -lp + lq;
        const double dot = (gc[0] * al0 + gc[1] * al1) / s;
        emit_grad_pre<ADAM, COH>(c, 4, 2 * g, pa0 / s * (gc[0] - dot) * al0, ak, up[0], um[0], uv[0]);
        emit_grad_pre<ADAM, COH>(c, 4, 2 * g + 1, pa0 / s * (gc[1] - dot) * al1, ak, up[1], um[1], uv[1]);
        if (acc_on) {
            const double lpn = coh_ld<COH>(c.lpn + g), eps = coh_ld<COH>(c.eps_noise + g);
            const double gl = COH != 0 ? noise_row_coherent(c, g) : lik_row(c, kPGnoise, g);
            const double ns = fit_noise ? exp((double)ns_u) : kPiNoiseSd;
            // Normal(0, 0.655) prior held in float32 by the reference (utils.py:158-161)
            const float nsf = 0.655f;
            const double nvar = (double)(nsf * nsf);
            const double logp = -lpn * lpn / (2.0 * nvar) - (double)logf(nsf) - kHalfLog2PiC;
            const double logq = -0.5 * eps * eps - log(ns) - kHalfLog2PiC;
            loss_fin += -logp + logq;
            if (fit_noise) {
                const double Gl = gl + lpn / nvar;
                emit_grad<ADAM, COH>(c, 5, g, Gl, ak);
                emit_grad<ADAM, COH>(c, 6, g, Gl * eps * ns - 1.0, ak);
                if (ADAM) {
                    nl = coh_ld<COH>(c.p[5] + g);
                    ns_u = coh_ld<COH>(c.p[6] + g);
                }
            }
        }
    }
    if (PREP && acc_on) {
        double eps;
        if (c.eps_noise_in) {
            eps = c.eps_noise_in[g];
        } else {
            eps = (double)normal2_at(c.seed, ((unsigned long long)kSiteNoise << 48) + (unsigned long long)(c.g_off + g),
                                         s_prep * 4ull).x;
        }
        const double ns = fit_noise ? exp((double)ns_u) : kPiNoiseSd;
        coh_st<COH>(c.eps_noise + g, eps);
        coh_st<COH>(c.lpn + g, (fit_noise ? (double)nl : 0.0) + eps * ns);
        if (c.eps_noise_out) c.eps_noise_out[g] = eps;
    }
    if (PREP && c.dgq) {
        // lgamma / digamma of the guide-side concentrations of the (updated) alpha_pi, for
        // the next guide kernel and the next FINISH
        const double al0 = (double)expf(up[0]), al1 = (double)expf(up[1]);
        const double s = al0 + al1, pa0 = c.pi_a0[g];
        const double c0 = al0 / s * pa0, c1 = al1 / s * pa0;  // as FINISH forms c_p
        const double q0 = c0 < 1e-5 ? 1e-5 : c0, q1 = c1 < 1e-5 ? 1e-5 : c1;
        double lgS, dgS, lg0, dg0, lg1, dg1;
        lgamma_digamma(q0 + q1, lgS, dgS);
        lgamma_digamma(q0, lg0, dg0);
        lgamma_digamma(q1, lg1, dg1);
        const long Gl = c.G;
        coh_st<COH>(c.dgq + g, lgS);
        coh_st<COH>(c.dgq + Gl + g, lg0);
        coh_st<COH>(c.dgq + 2 * Gl + g, lg1);
        coh_st<COH>(c.dgq + 3 * Gl + g, dgS);
        coh_st<COH>(c.dgq + 4 * Gl + g, dg0);
        coh_st<COH>(c.dgq + 5 * Gl + g, dg1);
    }
}

// Survival, k_param's q0 blocks (PREP): the Gamma draws of the Dirichlet-over-all-guides site
// (survival MixtureNormal q0, survival NormalModel initial_abundance) of the step just prepared, from
// the concentration the lane has just updated, and their normalisers.
//   gam[r, g]          one lane per guide draws its R gammas, two per rejection loop;
//   gpart[block, j]    block sums (j < R: gammas of replicate j; j = R: the concentrations);
//   gsum[j]            the q0 block that arrives LAST adds the partials of all blocks, in a fixed
//                      order (strided partials, wave tree).
// These were two more launches per step (k_q0_draws 13 us, k_sum_parts 6 us at BASELINE config 5) whose
// only product is R + 1 sums.  Hand-over as in the fused step kernel (bean_step_v2.hpp): agent-scope
// stores, s_waitcnt, one relaxed agent-scope atomic per block; no fence.
// torch draws this site in float32 (the concentration is a float32 parameter): the gamma underflows
// to 0 and is floored at FLT_MIN (ATen _s_dirichlet_cpu); sample_gamma_pair_floor32 uses that floor to
// leave the rejection loop out where U^(1/alpha) has already decided the result (with the
// concentrations of 1 / n_guides the site starts from, ~90 % of the waves).
__device__ __forceinline__ void q0_draws_and_totals(const DevArgs& c, int gb, int g, bool in, double conc,
                                                    unsigned long long step, unsigned rec) {
    (void)rec;
    constexpr int kChunk = 16;                 // values reduced per barrier pair
    constexpr int kWaves = kParamBlock / 64;
    __shared__ double qs[kChunk][kWaves];      // wave sums of the current chunk of values
    __shared__ int is_last;
    const int R = c.R, np1 = c.R + 1;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    BEAN_STAMP_RT(rec, 1);
    double* mine = c.gpart + (long)gb * np1;
    // value j of the block: j < R the gammas of replicate j, j = R the concentrations; wave sums go to
    // LDS, and once per chunk (or at the end) thread k adds the waves' parts of value k in fixed order
    int n_in_chunk = 0, j0 = 0;
    auto flush = [&]() {
        __syncthreads();
        if ((int)threadIdx.x < n_in_chunk) {
            double t = 0.0;
            for (int i = 0; i < kWaves; ++i) t += qs[threadIdx.x][i];
            __hip_atomic_store(mine + j0 + threadIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        j0 += n_in_chunk;
        n_in_chunk = 0;
    };
    auto put = [&](double v) {
        const double t = wave_sum(v);
        if (lane == 0) qs[n_in_chunk][w] = t;
        if (++n_in_chunk == kChunk) flush();
    };
    for (int r0 = 0; r0 < R; r0 += 2) {
        const int r1 = r0 + 1;
        const bool two = r1 < R;
        double gm0 = 0.0, gm1 = 0.0;
        if (in) {
            if (c.x0_in) {
                gm0 = c.x0_in[(long)r0 * c.G + g];  // injected draws: already normalised
                if (two) gm1 = c.x0_in[(long)r1 * c.G + g];
            } else {
                Rng rng(c.seed, kSiteQ0, (unsigned long long)r0 * c.G_tot + (c.g_off + g), step * 256ull);
                const GammaPair gp = sample_gamma_pair_floor32(conc, two ? conc : 0.0, rng);
                gm0 = (double)fmaxf((float)gp.g0, 1.17549435e-38f);
                gm1 = two ? (double)fmaxf((float)gp.g1, 1.17549435e-38f) : 0.0;
            }
            c.gam[(long)r0 * c.G + g] = gm0;
            if (two) c.gam[(long)r1 * c.G + g] = gm1;
        }
        put(gm0);
        if (two) put(gm1);
    }
    BEAN_STAMP_RT(rec, 3);
    put(in ? conc : 0.0);
    if (n_in_chunk) flush();
    // this block's partials are out; count in, and the block that arrives last forms the totals:
    // wave i takes values i, i + 4, ...; lanes stride over the blocks' partials, fixed tree
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial stores (all issued by wave 0) are acknowledged
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(c.q0_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = old == c.n_gamma_blocks - 1;
        if (is_last) __hip_atomic_store(c.q0_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    BEAN_STAMP_RT(rec, 4);
    if (is_last) {
        for (int j = w; j < np1; j += kWaves) {
            double v = 0.0;
            for (int b = lane; b < c.n_gamma_blocks; b += 64)
                v += __hip_atomic_load(c.gpart + (long)b * np1 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double tot = wave_sum(v);
            if (lane == 0) c.gsum[j] = tot;
        }
    }
}

// ------------------------------------------------- allele_slot_tables (k_allele, k_param's allele blocks)
// Tiling: per allele slot (g, a >= 1): mu_a = sum of its edits' mu, sigma_a =
// l2 norm of their sd (model.py:618-622) as a CSR gather, then the bin
// probabilities and their derivatives.  Tables are laid out (B, A-1, G).
// One allele slot (g, a1): shared by k_allele (one thread per slot of the screen) and by the head of
// k_guide_tiling_rep (the slots of the workgroup's own guides).
// COH == 2: the edits' draws are loaded at agent scope - the allele blocks of k_param (round 5) read what the edit blocks
// of the SAME launch have just stored from other CUs (see coh_ld).
template <int COH = 0>
__device__ __forceinline__ void allele_slot_tables(const DevArgs& c, int a1, int g) {
    const int A1 = c.A - 1;
    const long idx = slot_off(c, a1, g);
    const long slot = (long)g * A1 + a1;
    const bool valid = c.amask[(long)g * c.A + a1 + 1] != 0;
    if (c.survival) {
        // growth of the allele over the timepoints, exp((u_g + sum_e mu_e) t_b); masked alleles get
        // probability 0 (survival_model.py:484-488, 561-567)
        double mu = 0.0;
        for (int k = c.a2e_ptr[slot]; k < c.a2e_ptr[slot + 1]; ++k) mu += c.mu_t[c.a2e_idx[k]];
        c.mu_a[idx] = mu;
        const double full = c.u_g[g] + mu;
        for (int b = 0; b < c.B; ++b) {
            const double tb = c.time[b];
            const double P = valid ? exp(full * tb) : 0.0;
            const long o = tab_off(c, b, a1, g);
            c.tabP[o] = P;
            c.tabPmu[o] = tb * P;
            c.tabPy[o] = 0.0;
        }
        return;
    }
    double mu = 0.0, var = 0.0;
    {
        // four edits of the allele at a time: their indices asked for together, then their draws together, then the sums
        // in the plain loop's order (same bits; an allele has two edits on average, and one per pass was two dependent
        // round trips each - at agent scope in k_param's allele blocks)
        const int k1 = c.a2e_ptr[slot + 1];
        for (int kb = c.a2e_ptr[slot]; kb < k1; kb += 4) {
            int ei[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ei[u] = kb + u < k1 ? c.a2e_idx[kb + u] : -1;
            double xm[4], xy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xm[u] = 0.0;
                xy[u] = 0.0;
                if (ei[u] >= 0) {
                    xm[u] = coh_ld<COH>(c.mu_t + ei[u]);
                    xy[u] = coh_ld<COH>(c.y_t + ei[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ei[u] >= 0) {
                    mu += xm[u];
                    const double sd = exp(xy[u]);
                    var += sd * sd;
                }
        }
    }
    const double sigma = sqrt(var);
    c.mu_a[idx] = mu;
    c.sig_a[idx] = sigma;
    const bool live = valid && var > 0.0;  // masked alleles: probability 0, no gradient (utils.py:56-59,73-74)
    const double inv = live ? 1.0 / sigma : 0.0;
    // bins are sorted by their bounds (data_class.py:948-964), so a bin's lower edge is very often the
    // previous bin's upper edge: its Phi / phi are reused instead of evaluated twice (same values)
    double pz = 0.0, pch = 0.0, pfh = 0.0, pufh = 0.0;
    bool have_prev = false;
    for (int b = 0; b < c.B; ++b) {
        double P = 0.0, dmu = 0.0, dsig = 0.0;
        if (live) {
            const double zh = c.z_hi[b], zl = c.z_lo[b];
            double ch = 1.0, cl = 0.0, fh = 0.0, fl = 0.0, ufh = 0.0, ufl = 0.0;
            if (!isinf(zl)) {
                if (have_prev && zl == pz) {
                    cl = pch;
                    fl = pfh;
                    ufl = pufh;
                } else {
                    const double u = (zl - mu) * inv;
                    cl = norm_cdf(u);
                    fl = norm_pdf(u);
                    ufl = u * fl;
                }
            }
            if (!isinf(zh)) {
                const double u = (zh - mu) * inv;
                ch = norm_cdf(u);
                fh = norm_pdf(u);
                ufh = u * fh;
                pz = zh;
                pch = ch;
                pfh = fh;
                pufh = ufh;
                have_prev = true;
            }
            P = ch - cl;
            dmu = -(fh - fl) * inv;
            dsig = -(ufh - ufl) * inv;
        }
        const long o = tab_off(c, b, a1, g);
        c.tabP[o] = P;
        c.tabPmu[o] = dmu;
        c.tabPy[o] = dsig;  // d/d sigma_a here (chain to y_e in k_param)
    }
}


// -------------------------------------------------------------------- k_param
// grid = n_target_blocks + n_guide_blocks (+ the q0 blocks of the survival families with a
// Dirichlet-over-all-guides site), 256 threads; the roles' dispatch order: see `bid` below.
// KIND 1: the variant sorting families on the wave-form path in thin mode (what a `bean run ... variant`
// fit of a sorting screen launches 2 000 times).  The launch conditions are stated to the compiler, which
// drops the other families' code: 121 -> <= 96 VGPRs without scratch (five instead of four resident
// waves per SIMD, so the 1 026 blocks of a 62.5k-guide shard are one round, not one round and two blocks)
// and less than half the instructions to fetch.  KIND 0: everything.
// Allele blocks (round 5; KIND 3, sorting, PREP; n_allele_blocks > 0): the LAST n_allele_blocks blocks of the grid do
// k_allele's work for the step this launch prepares - one thread per live allele slot - instead of a launch of their own
// behind this one.  They need the draws of ALL the edit blocks: every edit block counts in (after its stores of mu_t /
// y_t - agent scope - have completed), the edit block that counts in last raises the go flags, an allele block's first
// thread polls one of them (bounded; the edit blocks are the first blocks of the grid, i.e. resident before any allele
// block is dispatched) and the block then reads the draws at agent scope.  The allele block that leaves last zeroes
// counters and flags for the next launch.  Same arithmetic as k_allele: same bits.
// The words live in DevArgs::tile_ctr (tiling: kAlleleCtrLines lines of 128 bytes; k_set_step zeroes them): line 0 the edit
// blocks' arrivals, line 1 the allele blocks', lines 2 ... the go flags - SIXTEEN copies in lines of their own, because
// 753 blocks polling one word kept one memory channel busy for everybody (the first form: every block of the launch ran
// ~1.5 x longer while the allele blocks polled; scripts/stamps_kp_tiling.py).
constexpr int kAlleleGoFlags = 16;
constexpr int kAlleleCtrLines = 2 + kAlleleGoFlags;
constexpr int kAlleleSpinMax = 1 << 20;  // polls (~0.5 us each) before an allele block gives up: the step's loss reports NaN
template <bool FINISH, bool ADAM, bool PREP, int KIND = 0>
#ifndef BEAN_KP3_WAVES
#define BEAN_KP3_WAVES 1
#endif
// (-DBEAN_KP3_WAVES=7 holds KIND 3 to seven waves per SIMD, 72 VGPRs: what it takes by itself in the shipped build; an
// intermediate form of the allele blocks took 74 - 1 536 places instead of 1 792 at BASELINE config 3 - and measured the
// same step time held to 72 (5 spilled) or not)
__global__ __launch_bounds__(kParamBlock) __attribute__((amdgpu_waves_per_eu(KIND == 3 ? BEAN_KP3_WAVES : 1)))
void k_param(DevArgs c, int n_target_blocks) {
    // (KIND 3: what the grid holds beyond the edit blocks and the guide blocks - kAMax lanes per guide - are allele blocks)
    const int n_allele_blocks =
        KIND == 3 ? (int)gridDim.x - n_target_blocks - (int)(((long)c.G * kAMax + kParamBlock - 1) / kParamBlock) : 0;
    if (KIND == 1) {
        __builtin_assume(c.lpt == kLanesPerTarget);
        __builtin_assume(!c.survival);
        __builtin_assume(c.family != kMultiMixture);
        __builtin_assume(!c.wide_targets);
        __builtin_assume(c.tgrad == nullptr);
        __builtin_assume(c.n_cov == 0);
        __builtin_assume(c.wrow != nullptr);
        __builtin_assume(c.rows_v2 != 0);
        __builtin_assume(c.rrow == nullptr);
        __builtin_assume(!c.surv_q0lik);
        __builtin_assume(!c.not_loss_owner);
        __builtin_assume(c.lpart != nullptr);
        __builtin_assume(c.dgq != nullptr || c.family != kMixture);
        __builtin_assume(c.tsum != nullptr);
    }
    if (KIND == 2) {  // survival variant MixtureNormal on the wave-form path (thin mode, unsharded parameters)
        __builtin_assume(c.lpt == kLanesPerTargetNarrow);
        __builtin_assume(c.survival != 0);
        __builtin_assume(c.family == kMixture);
        __builtin_assume(!c.surv_q0lik);
        __builtin_assume(c.dgq != nullptr);
        __builtin_assume(c.tsum == nullptr);
        __builtin_assume(!c.wide_targets);
        __builtin_assume(c.tgrad == nullptr);
        __builtin_assume(c.n_cov == 0);
        __builtin_assume(c.wrow != nullptr);
        __builtin_assume(c.rows_v2 != 0);
        __builtin_assume(c.rrow == nullptr);
        __builtin_assume(c.lpart != nullptr);
    }
    if (KIND == 3) {  // tiling (MultiMixtureNormal) in the register-resident wave form, thin mode
        __builtin_assume(c.lpt == kLanesPerTargetNarrow);
        __builtin_assume(c.family == kMultiMixture);
        __builtin_assume(!c.wide_targets);
        __builtin_assume(!c.wide_alleles);
        // (c.tgrad: either - a guide-sharded fit's exchanged update is this build too, with its allele blocks)
        __builtin_assume(c.n_cov == 0);
        __builtin_assume(c.lpart == nullptr);
        __builtin_assume(c.trow_summed != 0);
        __builtin_assume(!c.surv_q0lik);
    }
    // (dispatch order: edit blocks, c.q0_blk0 guide blocks, the allele blocks, the other guide blocks - launch_param)
    const int allele_blk0 = n_target_blocks + c.q0_blk0;
    if (KIND == 3 && PREP && n_allele_blocks > 0 && (int)blockIdx.x >= allele_blk0 &&
        (int)blockIdx.x < allele_blk0 + n_allele_blocks) {
        const long idx = (long)((int)blockIdx.x - allele_blk0) * blockDim.x + threadIdx.x;
        const int stamp_rec = (int)gridDim.x - n_allele_blocks + ((int)blockIdx.x - allele_blk0);  // behind the guide blocks' records
        (void)stamp_rec;
        BEAN_STAMP_RT(stamp_rec, 0);
        const bool in = idx < c.n_live_slots;
        const int sl = in ? c.live_slots[idx] : 0;  // a1 * G + g
        {
            // data the slot's chain begins with, asked for now (the loads behind the poll then find it in the caches)
            const int a1 = sl / c.G, g = sl - a1 * c.G;
            const long slot = (long)g * (c.A - 1) + a1;
            const int k0 = c.a2e_ptr[slot], k1 = c.a2e_ptr[slot + 1];
            const int e0 = k0 < k1 ? c.a2e_idx[k0] : 0;
            const int am = c.amask[(long)g * c.A + a1 + 1];
            asm volatile("" ::"v"(e0), "v"(am), "v"(k1));
        }
        if (threadIdx.x == 0) {
            const int* const go = c.tile_ctr + 32 * (2 + ((int)blockIdx.x & (kAlleleGoFlags - 1)));
            int spins = 0;
            while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                if (++spins > kAlleleSpinMax) {
                    const StepCtr cs = *c.ctrA;
                    atomicAdd((unsigned long long*)(c.loss_acc + ((long)cs.slot * kLossSub) * kLossWords) + 2, 1ull);
                    break;
                }
                __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
        asm volatile("" ::: "memory");  // (nothing below is loaded before the poll has matched)
        BEAN_STAMP_RT(stamp_rec, 1);
        if (in) allele_slot_tables<2>(c, sl / c.G, sl % c.G);
        __syncthreads();
        BEAN_STAMP_RT(stamp_rec, 7);
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(c.tile_ctr + 32, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == n_allele_blocks - 1) {
                for (int k = 0; k < kAlleleCtrLines; ++k)
                    __hip_atomic_store(c.tile_ctr + 32 * k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // Block roles, in role order: target blocks, guide blocks, (survival q0 site) q0 blocks.  The q0
    // blocks hold the kernel's longest chain (parameter update -> gamma draws -> block sums -> the last
    // one's totals, ~15 us at BASELINE config 5 against ~10 us of an alpha_pi guide block and ~5 us of a
    // target block), so they are dispatched first, the guide blocks next and the target blocks last.
    // (As a tail of the guide blocks - this round's first form - the draws began when the alpha_pi update
    // ended: k_param 33 us.  As blocks of their own that waited for a guide block's flag: 35 us.  Without a
    // q0 site the order of guide and target blocks makes no difference: measured on configs 1 and 3.)
    unsigned bid = blockIdx.x;
    if (KIND == 3 && PREP && n_allele_blocks > 0 && (int)blockIdx.x >= allele_blk0) bid -= (unsigned)n_allele_blocks;
    if (c.q0_blocks) {
        const unsigned ntb = (unsigned)n_target_blocks, nq0 = (unsigned)c.n_gamma_blocks, ngd = (unsigned)c.q0_blk0;
        if (blockIdx.x < nq0) bid = ntb + ngd + blockIdx.x;
        else if (blockIdx.x < nq0 + ngd) bid = ntb + (blockIdx.x - nq0);
        else bid = blockIdx.x - nq0 - ngd;
    }
    __shared__ double scratch[16];
    __shared__ double hand[4][kTargetsPerBlockMax];  // phase hand-over: gmu, gy (A -> B), mu, y (B -> C)
#if BEAN_KP_DIAG == 1  // diagnostic builds (wrong results): time the target part alone ...
    if ((int)bid >= n_target_blocks) return;
#elif BEAN_KP_DIAG == 2  // ... or the guide part alone
    if ((int)bid < n_target_blocks && bid != 0) return;
#endif
#if defined(BEAN_STAMP) && BEAN_STAMP == 2
    const int lane = threadIdx.x & 63;
    const long wave_gid = (long)bid * (kParamBlock / 64) + (threadIdx.x >> 6);
#endif
    BEAN_STAMP_KP(0);
    if (PREP && FINISH) BEAN_STAMP_RT(bid, 0);
    const StepCtr ctr = *c.ctrA;
    const unsigned long long s_prep = FINISH ? ctr.step + 1 : ctr.step;
    const unsigned long long slot_prep = FINISH ? ctr.slot + 1 : ctr.slot;
    AdamCoef ak;
    ak.step_size = 0.f;
    ak.clip = 0.f;
    if (FINISH && ADAM) {
        ak.step_size = ctr.step_size;  // of update t = ctr.step + 1, computed by the guide kernel (publish_ctr)
        ak.clip = (float)c.clip;
    }
    double loss_fin = 0.0, loss_prep = 0.0;
    const bool mixture = c.family == kMixture;
    // The guide kernel's per-wave loss parts (wave_loss_out).  Thin mode: every target block takes its
    // share; the block's LAST wave issues the loads now and adds them up while it would otherwise idle at
    // the barrier behind phase B (summed by the first blocks at the end of the kernel, they were the last
    // ~2 us of its critical path).  Other modes: the strided pass at the end.
    __shared__ long long lp3[3];
    const bool lp_early = FINISH && c.lpart != nullptr && !c.wide_targets && !c.tgrad;
    long long lw0 = 0, lw1 = 0, lw2 = 0;
    if (lp_early && (int)bid < n_target_blocks && threadIdx.x >= blockDim.x - 64) {
        const long per = (c.n_lpart + n_target_blocks - 1) / n_target_blocks;
        const long e0 = (long)bid * per, e1 = e0 + per < c.n_lpart ? e0 + per : c.n_lpart;
        for (long i = e0 + (threadIdx.x & 63); i < e1; i += 64) {
            lw0 += c.lpart[3 * i];
            lw1 += c.lpart[3 * i + 1];
            lw2 += c.lpart[3 * i + 2];
        }
    }

    if (c.survival && mixture && (int)bid >= n_target_blocks + c.q0_blk0) {
        // survival MixtureNormal: the Dirichlet(q0) site over ALL guides and the per-guide
        // baseline growth draw (survival_model.py:259-274,306-311,660-669)
        const int gb = (int)bid - n_target_blocks - c.q0_blk0;
        const int g = gb * kParamBlock + threadIdx.x;
        const bool in = g < c.G;
        float q0u = in ? c.p[7][g] : 0.f;
        if (FINISH && in) {
            float q0m = 0.f, q0v = 0.f;
            if (ADAM) {
                q0m = c.m[7][g];
                q0v = c.v[7][g];
            }
            const double q0 = (double)expf(q0u);
            emit_grad_pre<ADAM>(c, 7, g, part_row(c, kPQ0, g) * q0, ak, q0u, q0m, q0v);
            // - log p(mu_negctrl): Normal(m0, s0) built from Python floats => float32 tensors
            const float s0f = (float)c.neg_scale;
            const double du = c.u_g[g] - (double)(float)c.neg_loc;
            loss_fin += du * du / (2.0 * (double)(s0f * s0f)) + (double)logf(s0f) + kHalfLog2PiC;
        }
        if (PREP) {
            double q0 = 0.0;
            if (in) {
                q0 = (double)expf(q0u);
                double eps;
                if (c.eps_u_in) {
                    eps = c.eps_u_in[g];
                } else {
                    eps = (double)normal2_at(c.seed, ((unsigned long long)kSiteAux << 48) + (unsigned long long)(c.g_off + g),
                                                 s_prep * 4ull).x;
                }
                c.eps_u[g] = eps;
                c.u_g[g] = (double)(float)c.neg_loc + eps * (double)(float)c.neg_scale;
                if (c.eps_u_out) c.eps_u_out[g] = eps;
            }
            q0_draws_and_totals(c, gb, g, in, q0, s_prep, bid);
        }
    }
    if ((int)bid < n_target_blocks) {
        // ------------------------------------------------ target part
        // (the edit blocks head the launch's longest chain when allele blocks follow; a raised issue priority for them -
        // s_setprio 2 - measured nothing: 142.1 against 142.1 us per step)
        int t;
        bool active;
        double gmu = 0.0, gy = 0.0, tab_mu = 0.0, tab_y = 0.0;
        target_of_thread(c, t, active, bid);
        // sorting families: the target's parameters, moments and last draw are loaded BEFORE the
        // gradient sums (independent of them), not after
        float pf[4] = {0.f, 0.f, 0.f, 0.f}, mf[4] = {0.f, 0.f, 0.f, 0.f}, vf[4] = {0.f, 0.f, 0.f, 0.f};
        double eps1_f = 0.0, eps2_f = 0.0, mu_f = 0.0, y_f = 0.0;
        if (active) {
            const int n_lat = c.survival ? 2 : 4;  // survival: mu only (loc, scale)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < n_lat) {
                    pf[i] = c.p[i][t];
                    if (FINISH && ADAM) {
                        mf[i] = c.m[i][t];
                        vf[i] = c.v[i][t];
                    }
                }
            }
            if (FINISH) {
                eps1_f = c.eps_mu[t];
                mu_f = c.mu_t[t];
                if (!c.survival) {
                    eps2_f = c.eps_sd[t];
                    y_f = c.y_t[t];
                }
            }
        }
        // the standard normals of the NEXT step's draw depend on (seed, target, step) only: formed here, under
        // the latency of the loads above and of the gradient sums below, not behind the update they will be
        // scaled by (Philox + Box-Muller are ~250 instructions of this kernel's one-lane-per-target chain)
        float2 nrm_next = make_float2(0.f, 0.f);
        if (PREP && active && !c.eps_mu_in) {
            nrm_next = normal2_at(c.seed, ((unsigned long long)kSiteTarget << 48) + (unsigned long long)(c.t_off + t),
                                  s_prep * 4ull);
        }
        if (FINISH) {
            if (c.tgrad) {
                // sharded run of a family whose per-target parameters are shared across shards: the
                // sums were formed by k_target_reduce and all-reduced by the host
                if (active) {
                    gmu = c.tgrad[t];
                    gy = c.tgrad[c.T + t];
                }
            } else if (c.wide_targets) {
                target_grad_sums(c, t, active, scratch, gmu, gy);
            } else {
                // phase A (group map) -> phase B (owner map) through LDS
                int tg;
                bool lead;
                target_of_group(c, tg, lead, bid);
                double a, b;
                target_grad_sums(c, tg, lead, scratch, a, b);
                if (lead) {
                    hand[0][threadIdx.x / c.lpt] = a;
                    hand[1][threadIdx.x / c.lpt] = b;
                }
                __syncthreads();
                if (active) {
                    gmu = hand[0][threadIdx.x];
                    gy = hand[1][threadIdx.x];
                }
                if (lp_early && threadIdx.x >= blockDim.x - 64) {  // idle until the next barrier
                    lw0 = wave_sum_i64(lw0);
                    lw1 = wave_sum_i64(lw1);
                    lw2 = wave_sum_i64(lw2);
                    if ((threadIdx.x & 63) == 0) {
                        lp3[0] = lw0;
                        lp3[1] = lw1;
                        lp3[2] = lw2;
                    }
                }
            }
        }
        BEAN_STAMP_KP(1);
        if (active && c.survival) {
            // survival families: mu only (no sd latent), growth tables are computed in k_guide_survival
            // (parameters, moments and last draw loaded before the gradient sums, like the sorting families')
            float pl = pf[0], psu = pf[1];
            if (FINISH) {
                const double eps1 = eps1_f, mu = mu_f;
                const double s_mu = exp((double)psu);
                double logp_mu, dlogp_mu;
                if (c.flags & kPriorNormalMu) {
                    const double ploc = c.pr_mu_loc ? c.pr_mu_loc[t] : 0.0;
                    const double ps = c.pr_mu_scale ? c.pr_mu_scale[t] : 1.0;
                    const double zz = (mu - ploc) / ps;
                    logp_mu = -0.5 * zz * zz - log(ps) - kHalfLog2PiC;
                    dlogp_mu = -zz / ps;
                } else {
                    logp_mu = -kLog2 - fabs(mu);
                    dlogp_mu = mu > 0.0 ? -1.0 : (mu < 0.0 ? 1.0 : 0.0);
                }
                const double logq_mu = -0.5 * eps1 * eps1 - (double)psu - kHalfLog2PiC;
                loss_fin = -logp_mu + logq_mu;
                const double Gmu = gmu - dlogp_mu;
                emit_grad_pre<ADAM>(c, 0, t, Gmu, ak, pl, mf[0], vf[0]);
                emit_grad_pre<ADAM>(c, 1, t, Gmu * eps1 * s_mu - 1.0, ak, psu, mf[1], vf[1]);
            }
            if (PREP) {
                double eps1;
                if (c.eps_mu_in) {
                    eps1 = c.eps_mu_in[t];
                } else {
                    eps1 = (double)nrm_next.x;
                }
                c.eps_mu[t] = eps1;
                c.mu_t[t] = (double)pl + eps1 * exp((double)psu);
                if (c.eps_mu_out) c.eps_mu_out[t] = eps1;
            }
        } else if (active) {
            if (FINISH) {
                const double eps1 = eps1_f, eps2 = eps2_f;
                const double s_mu = exp((double)pf[1]), s_sd = exp((double)pf[3]);
                double dlogp_mu, dlogp_dy;
                tgt_prior_terms(c, t, tgt_sd_prior(c, t), mu_f, y_f, eps1, eps2, pf[1], pf[3], dlogp_mu, dlogp_dy,
                                loss_fin);
                const double Gmu = gmu - dlogp_mu;
                const double Gy = gy - dlogp_dy;
                emit_grad_pre<ADAM>(c, 0, t, tgt_grad(0, Gmu, eps1, s_mu), ak, pf[0], mf[0], vf[0]);
                emit_grad_pre<ADAM>(c, 1, t, tgt_grad(1, Gmu, eps1, s_mu), ak, pf[1], mf[1], vf[1]);
                emit_grad_pre<ADAM>(c, 2, t, tgt_grad(2, Gy, eps2, s_sd), ak, pf[2], mf[2], vf[2]);
                emit_grad_pre<ADAM>(c, 3, t, tgt_grad(3, Gy, eps2, s_sd), ak, pf[3], mf[3], vf[3]);
            }
            BEAN_STAMP_KP(2);
            if (PREP) {
                double eps1, eps2;
                if (c.eps_mu_in) {
                    eps1 = c.eps_mu_in[t];
                    eps2 = c.eps_sd_in[t];
                } else {
                    eps1 = (double)nrm_next.x;
                    eps2 = (double)nrm_next.y;
                }
                const double mu = tgt_draw(pf[0], eps1, pf[1]);
                const double y = tgt_draw(pf[2], eps2, pf[3]);
                c.eps_mu[t] = eps1;
                c.eps_sd[t] = eps2;
                // (KIND 3: written through - the allele blocks of this launch read them from other CUs)
                coh_st<KIND == 3 ? 2 : 0>(c.mu_t + t, mu);
                coh_st<KIND == 3 ? 2 : 0>(c.y_t + t, y);
                if (c.eps_mu_out) {
                    c.eps_mu_out[t] = eps1;
                    c.eps_sd_out[t] = eps2;
                }
                tab_mu = mu;
                tab_y = y;
            }
        }
        BEAN_STAMP_KP(3);
        // ---- Phi tables: the B entries of a target are spread over the lanes of its group
        // (thin mode: kLanesPerTarget consecutive lanes; wide mode: the block's first threads)
        if (PREP && !c.survival && c.family != kMultiMixture) {
            if (c.wide_targets) {
                if (threadIdx.x == 0) {
                    scratch[0] = tab_mu;
                    scratch[1] = tab_y;
                }
                __syncthreads();
                if ((int)threadIdx.x < c.B) {
                    if (c.n_cov)
                        for (int r = 0; r < c.R; ++r)
                            write_phi_entry(c, t, threadIdx.x, scratch[0] + c.cov_shift[r], scratch[1], (long)r * c.B * c.T);
                    else
                        write_phi_entry(c, t, threadIdx.x, scratch[0], scratch[1]);
                }
                __syncthreads();
            } else {
                // one lane per bin EDGE: even lane = upper edge, odd lane = lower edge of bin e >> 1, so
                // the erf / exp chain of a target is one edge deep (it was 2 B / 4 edges deep with four
                // lanes per target); the pair is combined with one shuffle.  Same formulas, same bits as
                // write_phi_entry.
                if (active) {
                    hand[2][threadIdx.x] = tab_mu;
                    hand[3][threadIdx.x] = tab_y;
                }
                __syncthreads();
                const int grp = threadIdx.x / kLanesPerTarget;
                t = (bid * blockDim.x + threadIdx.x) / kLanesPerTarget;  // group map from here on
                const double mu = hand[2][grp], y = hand[3][grp];
                const int j = threadIdx.x & (kLanesPerTarget - 1);
                const double sigma = c.family == kNormal ? exp(0.5 * y) : exp(y);
                const double dsig_dy = c.family == kNormal ? 0.5 * sigma : sigma;
                const double inv = 1.0 / sigma;
                const int n_tab = c.n_cov ? c.R : 1;  // sample covariates: one table per replicate
                for (int rt = 0; rt < n_tab; ++rt) {
                    const double mu_r = c.n_cov ? mu + c.cov_shift[rt] : mu;
                    const long off = (long)rt * c.B * c.T;
                    for (int e0 = 0; e0 < 2 * c.B; e0 += kLanesPerTarget)
                        phi_edge(c, t, t < c.T, e0 + j, mu_r, inv, dsig_dy, off);
                }
            }
        }
        BEAN_STAMP_KP(4);
        if (KIND == 3 && PREP && n_allele_blocks > 0) {
            // this edit block's draws are out: count in for the allele blocks
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const int old = __hip_atomic_fetch_add(c.tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old == n_target_blocks - 1)  // the last edit block: every draw is out
                    for (int k = 0; k < kAlleleGoFlags; ++k)
                        __hip_atomic_store(c.tile_ctr + 32 * (2 + k), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            BEAN_STAMP_RT(bid, 4);
        }
    } else if (c.family == kMultiMixture) {
        const int guide_block = (int)bid - n_target_blocks;
        if (c.wide_alleles) param_guide_tiling_wide<FINISH, ADAM, PREP>(c, guide_block, s_prep, ak, loss_fin);
        else param_guide_tiling<FINISH, ADAM, PREP>(c, guide_block, s_prep, ak, loss_fin);
        if (c.survival) {
            // per-guide baseline growth mu_negctrl ~ N(m0, s0): sampled in the model only
            // (survival_model.py:479-483), i.e. a fresh prior draw each step
            const int tid = ((int)bid - n_target_blocks) * blockDim.x + threadIdx.x;
            // kAMax lanes per guide (param_guide_tiling) or one wave per guide (wide path): lane 0 acts
            const int lpg = c.wide_alleles ? 64 : kAMax;
            const int g = tid / lpg;
            if (g < c.G && tid % lpg == 0) {
                if (FINISH) {
                    const float s0f = (float)c.neg_scale;
                    const double du = c.u_g[g] - (double)(float)c.neg_loc;
                    loss_fin += du * du / (2.0 * (double)(s0f * s0f)) + (double)logf(s0f) + kHalfLog2PiC;
                }
                if (PREP) {
                    double eps;
                    if (c.eps_u_in) {
                        eps = c.eps_u_in[g];
                    } else {
                        eps = (double)normal2_at(c.seed, ((unsigned long long)kSiteAux << 48) + (unsigned long long)(c.g_off + g),
                                         s_prep * 4ull).x;
                    }
                    c.eps_u[g] = eps;
                    c.u_g[g] = (double)(float)c.neg_loc + eps * (double)(float)c.neg_scale;
                    if (c.eps_u_out) c.eps_u_out[g] = eps;
                }
            }
        }
    } else if (mixture) {
        // ------------------------------------------------- guide part
        // (survival: the q0 blocks follow the alpha_pi blocks; their g is out of range here)
        const int g = ((int)bid - n_target_blocks) * (int)blockDim.x + threadIdx.x;
        if (g < c.G) param_guide_mix<FINISH, ADAM, PREP, false>(c, g, ak, s_prep, loss_fin);
    }
    if (c.surv_q0lik && (int)bid >= n_target_blocks) {
        // survival NormalModel: Dirichlet(initial_abundance) site over ALL guides, drawn per
        // replicate and used by the likelihood (survival_model.py:62-67, 629-639).  The prior is
        // Dirichlet(1 / G), so unlike the MixtureNormal q0 site nothing cancels.
        const int gb = (int)bid - n_target_blocks;
        const int g = gb * kParamBlock + threadIdx.x;
        const bool in = g < c.G;
        float iau = in ? c.p[7][g] : 0.f;
        if (FINISH && in) {
            const double ia = (double)expf(iau);
            const double tot = c.gsum[c.R];
            double lg_tot, dg_tot, lg_a, dg_a;
            lgamma_digamma(tot, lg_tot, dg_tot);
            lgamma_digamma(ia, lg_a, dg_a);
            const double Rf = (double)c.R;
            // d/d ia of log q: direct term R (psi(tot) - psi(ia)) + sum_r log x, and the pathwise term
            double grad = Rf * (dg_tot - dg_a) + part_row(c, kPQ0, g);
            for (int r = 0; r < c.R; ++r) {
                const double gm = c.gam[(long)r * c.G + g];
                const double x = c.x0_in ? gm
                                         : (double)fminf(fmaxf((float)(gm * frcp(c.gsum[r])), 1.17549435e-38f),
                                                         0.99999994f);
                grad += dirichlet_grad_one(x, ia, tot) * (c.gq[(long)r * c.G + g] - c.sq[r]);
            }
            emit_grad<ADAM>(c, 7, g, grad * ia, ak);
            if (ADAM) iau = c.p[7][g];
            // normalisers: + log q: R (lgamma(tot) - sum lgamma(ia)); - log p: the prior
            // concentration is the float32 value of 1 / G on every guide (torch.ones(G) / G)
            const double pr = c.prior_ia ? c.prior_ia[g] : (double)(1.0f / (float)c.G_tot);
            double lg_p, dg_p;
            lgamma_digamma(pr, lg_p, dg_p);
            loss_fin += Rf * (lg_p - lg_a);
            if (c.g_off + g == 0) {  // once per screen (guide 0 of the whole screen)
                double lg_ps, dg_ps;
                lgamma_digamma(c.prior_ia ? c.prior_ia_total : pr * (double)c.G_tot, lg_ps, dg_ps);
                loss_fin += Rf * (lg_tot - lg_ps);
            }
        }
        if (PREP) {
            q0_draws_and_totals(c, gb, g, in, in ? (double)expf(iau) : 0.0, s_prep, bid);
        }
    }
    if (FINISH) {
        // replicated per-target parameters (sharded ControlNormal / tiling): their prior and entropy
        // terms are counted by one rank only
        // (sorting NormalModel with sample covariates: the replicated parameters are mu_cov's, handled by
        // k_cov_step; its per-target parameters are shard-local and count on every rank)
        if ((int)bid < n_target_blocks && c.not_loss_owner && !c.n_cov) loss_fin = 0.0;
        const double tot = block_sum(loss_fin, scratch);
        if (threadIdx.x == 0) {
            if (lp_early && (int)bid < n_target_blocks) {
                // this block's prior / entropy terms and its share of the guide kernel's loss parts in
                // one set of integer atomics
                long long a = lp3[0], b = lp3[1], d = lp3[2];
                if (fabs(tot) < kLossPartMax) {
                    const double hi = rint(tot * 1024.0);
                    a += (long long)hi;
                    b += (long long)rint((tot - hi * (1.0 / 1024.0)) * 1099511627776.0);
                } else {
                    d += 1;
                }
                long long* acc = c.loss_acc + ((long)ctr.slot * kLossSub + (bid & (kLossSub - 1))) * kLossWords;
                atomicAdd((unsigned long long*)acc, (unsigned long long)a);
                atomicAdd((unsigned long long*)acc + 1, (unsigned long long)b);
                if (d) atomicAdd((unsigned long long*)acc + 2, (unsigned long long)d);
            } else {
                loss_add(c, ctr.slot, tot);
            }
        }
        // other modes: the first blocks take 256 loss parts each
        if (c.lpart && !lp_early && (long)bid * blockDim.x < c.n_lpart) {
            __shared__ long long isum[3][16];
            long long ph = 0, pl = 0, pb = 0;
            for (long i = (long)bid * blockDim.x + threadIdx.x; i < c.n_lpart; i += (long)gridDim.x * blockDim.x) {
                ph += c.lpart[3 * i];
                pl += c.lpart[3 * i + 1];
                pb += c.lpart[3 * i + 2];
            }
            ph = wave_sum_i64(ph);
            pl = wave_sum_i64(pl);
            pb = wave_sum_i64(pb);
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
            if (lane == 0) {
                isum[0][w] = ph;
                isum[1][w] = pl;
                isum[2][w] = pb;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                long long a = 0, b = 0, d = 0;
                for (int i = 0; i < nw; ++i) {
                    a += isum[0][i];
                    b += isum[1][i];
                    d += isum[2][i];
                }
                long long* acc = c.loss_acc + ((long)ctr.slot * kLossSub + (bid & (kLossSub - 1))) * kLossWords;
                atomicAdd((unsigned long long*)acc, (unsigned long long)a);
                atomicAdd((unsigned long long*)acc + 1, (unsigned long long)b);
                if (d) atomicAdd((unsigned long long*)acc + 2, (unsigned long long)d);
            }
        }
    }
    (void)loss_prep;
    BEAN_STAMP_KP(7);
    if (PREP && FINISH) BEAN_STAMP_RT(bid, 7);
    if (bid == 0 && threadIdx.x == 0) {
        StepCtr nxt;
        nxt.step = s_prep;
        nxt.slot = slot_prep;
        nxt.step_size = 0.f;
        nxt.pad_ = 0.f;
        *c.ctrB = nxt;
    }
}

#ifndef BEAN_WAVE_EU
#define BEAN_WAVE_EU 4
#endif
#ifndef BEAN_GUIDE_WAVES_PER_EU
#define BEAN_GUIDE_WAVES_PER_EU 2
#endif
#ifdef BEAN_AB_KERNELS  // superseded forms, kept as A/B references: libbean_hip_ab.so only (-DBEAN_AB_KERNELS)
// --------------------------------------------------------------------- k_guide
// Dirichlet-Multinomial observation term of one (rep, guide):
// returns -log p and accumulates d(-log p)/d e[b] into ge.
template <int B>
__device__ __forceinline__ double dirmult_nll(const float* __restrict__ xp, long stride,
                                              const double* __restrict__ sf,
                                              const double* __restrict__ sm, double a0,
                                              const double (&e)[B], double (&ge)[B]) {
    double araw[B], ga[B], S = 0.0, n = 0.0;
    float x[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        x[b] = xp[b * stride];
        araw[b] = e[b] * sf[b];  // p[b] for now
        S += araw[b];
        n += (double)x[b];
    }
    const double inv = frcp(S + kEps);
    double A0 = 0.0;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        araw[b] = (araw[b] + kEps / B) * inv * a0 * sm[b];
        A0 += araw[b] < kEps ? kEps : araw[b];
    }
    const DD d0 = lgamma_digamma_diff(A0, n);
    double nll = d0.d, W = 0.0;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const bool clamped = araw[b] < kEps;
        const DD db = lgamma_digamma_diff(clamped ? kEps : araw[b], (double)x[b]);
        nll -= db.d;
        ga[b] = clamped ? 0.0 : (d0.dp - db.dp);  // d nll / d alpha_b
        W += ga[b] * araw[b];
    }
    W *= inv;
#pragma unroll
    for (int b = 0; b < B; ++b) ge[b] += (ga[b] * a0 * sm[b] * inv - W) * sf[b];
    return nll;
}

#ifndef BEAN_GUIDE_WAVES_PER_EU
#define BEAN_GUIDE_WAVES_PER_EU 2
#endif

// ------------------------------------------------------------ k_guide_wave (sorting, variant)
// One single-wave workgroup per (64-guide tile, replicate): the whole per-(rep, guide) chain
// (Dirichlet draw, both Dirichlet-Multinomial terms, pi terms, implicit
// reparameterisation gradient) with the likelihood written as rolled loops over the bins
// whose state is a handful of scalars (the d nll / d e[b] vector is never materialised:
// everything downstream is linear in it, see k_lik).  No LDS, no barrier; table entries and
// counts of bin b + 1 are fetched while bin b is computed.  The per-replicate rows go to
// wrow[(q, r, g)]; k_param sums them over r in fixed order (part_row).
#ifndef BEAN_WAVE_EU
#define BEAN_WAVE_EU 4
#endif
template <int FAM, bool ACC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BEAN_WAVE_EU)))
void k_guide_wave(DevArgs c) {
    constexpr bool MIX = FAM == kMixture;
    // dynamic LDS: [3][B][ntm] doubles (P, dP/dmu, dP/dy columns of the tile's targets; ntm =
    // c.tile_targets, the largest number of targets any tile spans) + [kWaveMisc][64] doubles
    // (per-guide values needed after the draw) + [2][B][64] floats (counts)
    extern __shared__ double tabs[];
    const int lane = threadIdx.x;
    const int g = blockIdx.x * 64 + lane;
    const int r = blockIdx.y;
    const bool valid = g < c.G;
    const int G = c.G, T = c.T, B = c.B;
    const StepCtr ctr = *c.ctrB;
    double loss = 0.0;
#ifdef BEAN_STAMP
    const long wave_gid = (long)blockIdx.y * gridDim.x + blockIdx.x;
#endif
    BEAN_STAMP_AT(0);

    // Guides are target-sorted, so the tile's targets are one contiguous range of at most 64:
    // lane i stages the 3 B table entries of target t0 + i (coalesced); every lane then reads its
    // own target's column from LDS.  Indices are clamped instead of predicated so that all loads
    // are in flight before the first wait.
    const int g_first = blockIdx.x * 64;
    const int g_last = (g_first + 63 < G ? g_first + 63 : G - 1);
    const int t0 = __builtin_amdgcn_readfirstlane(c.g2t[g_first]);
    const int nt = __builtin_amdgcn_readfirstlane(c.g2t[g_last]) - t0 + 1;
    const int ntm = c.tile_targets;
    double* ms = tabs + 3 * B * ntm + lane;               // ms[q * 64], q < kWaveMisc
    float* xs = (float*)(tabs + 3 * B * ntm + kWaveMisc * 64);  // counts: xs[(lik * B + b) * 64 + lane]
    const bool use_bc = (c.flags & kUseBc) != 0;
    // per-guide scalars needed right after the barrier: loaded with the batch, kept in registers
    int tcol = 0;
    bool rgm = false;
    float api0 = 0.f, api1 = 0.f;
    double pa0 = 0.0;
    {
        // Everything the wave reads from global memory, issued as one batch before the first wait:
        // the counts of both likelihoods, the table columns, and the per-guide values that are only
        // needed after the draw (staged in LDS so that no load sits behind the sampler call).
        const int gc = valid ? g : G - 1;
        const long rgc = (long)r * G + gc;
        float xv[2][kBMax];
#pragma unroll
        for (int b = 0; b < kBMax; ++b) {
            const long xo = ((long)r * B + (b < B ? b : B - 1)) * G + gc;
            xv[0][b] = c.X[xo];
            xv[1][b] = use_bc ? c.Xbc[xo] : 0.f;
        }
        // table rows (which, b) are packed `per` rows to a load: P = pow2 >= nt lanes per row
        int lg2 = 0;
        while ((1 << lg2) < nt) ++lg2;
        const int per = 64 >> lg2;                       // rows per load instruction
        const int j = lane & ((1 << lg2) - 1), sub = lane >> lg2;
        const int n_rows = 3 * B;
        constexpr int kTabLoads = 6;                     // covers 3 B rows when per >= 4 (nt <= 16)
        double tv[kTabLoads];
        int wbv[kTabLoads];
#pragma unroll
        for (int q = 0; q < kTabLoads; ++q) {
            const int wb = q * per + sub;
            const bool ok = wb < n_rows && j < nt;
            const int wbc = ok ? wb : 0;
            const int which = (wbc >= B) + (wbc >= 2 * B), bb = wbc - which * B;
            const double* tab = which == 0 ? c.tabP : (which == 1 ? c.tabPmu : c.tabPy);
            tv[q] = tab[(long)bb * T + t0 + (ok ? j : 0)];
            wbv[q] = ok ? wb : -1;
        }
        tcol = c.g2t[gc] - t0;
        rgm = c.rg[rgc] != 0;
        if (MIX) {
            api0 = c.p[4][2 * gc];
            api1 = c.p[4][2 * gc + 1];
            pa0 = c.pi_a0[gc];
        }
        const double nn0 = c.nobs[rgc], nn1 = use_bc ? c.nobs[(long)c.R * G + rgc] : -1.0;
        const double a00 = c.a0[gc], a01 = use_bc ? c.a0_bc[gc] : 0.0;
        double cnt0 = 0.0, cnt1 = 0.0;
        if (MIX)
            for (int cc = 0; cc < c.C; ++cc) {
                const float* al = c.allele + (((long)r * c.C + cc) * G + gc) * 2;
                cnt0 += (double)al[0];
                cnt1 += (double)al[1];
            }
#pragma unroll
        for (int b = 0; b < kBMax; ++b) {
            const int bb = b < B ? b : B - 1;
            xs[(0 * B + bb) * 64 + lane] = xv[0][b];
            xs[(1 * B + bb) * 64 + lane] = xv[1][b];
        }
#pragma unroll
        for (int q = 0; q < kTabLoads; ++q)
            if (wbv[q] >= 0) tabs[wbv[q] * ntm + j] = tv[q];
        // wide tiles (more than 16 targets in 64 guides): the remaining rows, one load per pass
        for (int wb0 = kTabLoads * per; wb0 < n_rows; wb0 += per) {
            const int wb = wb0 + sub;
            if (wb < n_rows && j < nt) {
                const int which = (wb >= B) + (wb >= 2 * B), bb = wb - which * B;
                const double* tab = which == 0 ? c.tabP : (which == 1 ? c.tabPmu : c.tabPy);
                tabs[wb * ntm + j] = tab[(long)bb * T + t0 + j];
            }
        }
        ms[0 * 64] = nn0;
        ms[1 * 64] = nn1;
        ms[2 * 64] = a00;
        ms[3 * 64] = a01;
        ms[4 * 64] = cnt0;
        ms[5 * 64] = cnt1;
    }
    __syncthreads();

    if (valid) {
        const long rgi = (long)r * G + g;
        const long RG = (long)c.R * G;
        const double* tp = tabs + tcol;  // this guide's column: tp[(which * B + b) * ntm]
        double cp0 = 1.0, cp1 = 1.0, cq0 = 1.0, cq1 = 1.0;
        bool cl0 = false, cl1 = false;
        double pi0 = 0.0, pi1 = 1.0, pe1 = 1.0;
        double dpe1_dpi1 = 0.0, dpe1_dl = 0.0;
        if (MIX) {
            const double al0 = (double)expf(api0), al1 = (double)expf(api1);
            const double rs = frcp(al0 + al1) * pa0;
            cp0 = al0 * rs;
            cp1 = al1 * rs;
            cl0 = cp0 < 1e-5;
            cl1 = cp1 < 1e-5;
            cq0 = cl0 ? 1e-5 : cp0;
            cq1 = cl1 ? 1e-5 : cp1;
            if (c.pi_in) {
                pi0 = c.pi_in[rgi * 2];
                pi1 = c.pi_in[rgi * 2 + 1];
            } else {
                BEAN_STAMP_AT(1);
                Rng rng(c.seed, kSitePi, (unsigned long long)r * c.G_tot + (c.g_off + g), ctr.step * 256ull);
                const GammaPair gp = sample_gamma_pair_inl(cq0, cq1, rng);
                const double gm0 = fmax(gp.g0, kDblMin), gm1 = fmax(gp.g1, kDblMin);
                const double rs2 = frcp(gm0 + gm1);
                pi0 = fmin(fmax(gm0 * rs2, kDblMin), kOneMinus);
                pi1 = fmin(fmax(gm1 * rs2, kDblMin), kOneMinus);
            }
            if (c.flags & kDumpPi) {
                c.pi_out[rgi * 2] = pi0;
                c.pi_out[rgi * 2 + 1] = pi1;
            }
            pe1 = pi1;
            if (ACC) {
                // scale_pi_by_accessibility + add_noise_to_pi, A = 2 (utils.py:106-178)
                const double kacc = c.kacc[g];
                const double s1 = pi1 * kacc;
                const bool in1 = s1 > 1e-3 && s1 < 1.0 - 1e-3;
                const double p1c = fmin(fmax(s1, 1e-3), 1.0 - 1e-3);
                const double l = flog(p1c * frcp(1.0 - p1c)) + c.lpn[g];
                const double el = exp(l);
                const double pn = el * frcp(1.0 + el);
                const bool in2 = pn > 1e-3 && pn < 1.0 - 1e-3;
                pe1 = fmin(fmax(pn, 1e-3), 1.0 - 1e-3);
                dpe1_dl = in2 ? pn * (1.0 - pn) : 0.0;
                dpe1_dpi1 = in1 ? dpe1_dl * frcp(p1c * (1.0 - p1c)) * kacc : 0.0;
            }
        }
        const double w0 = MIX ? (ACC ? 1.0 - pe1 : pi0) : 0.0;  // weight of the wild-type component
        const double w1 = MIX ? (ACC ? pe1 : pi1) : 1.0;        // weight of the edited component
        const double* sm = c.smask + r * B;
        const double epsB = kEps / (double)B;
        double a_mu = 0.0, a_y = 0.0, g0 = 0.0, g1 = 0.0, nll = 0.0;
        // pass 1 of both likelihoods: S = sum_b e_b sf_b with the same e_b and each one's size factors
        double S_x = 0.0, S_bc = 0.0;
        {
            const double* sfx = c.sf + r * B;
            const double* sfb = use_bc ? c.sf_bc + r * B : sfx;
            double ev[kBMax], s0v[kBMax], s1v[kBMax];
#pragma unroll
            for (int b = 0; b < kBMax; ++b) {
                const int bb = b < B ? b : B - 1;
                ev[b] = w0 * (MIX ? uniform_ld(c.P0, bb) : 0.0) + w1 * tp[bb * ntm];
                s0v[b] = uniform_ld(sfx, bb);
                s1v[b] = uniform_ld(sfb, bb);
            }
#pragma unroll
            for (int b = 0; b < kBMax; ++b) {
                S_x += b < B ? ev[b] * s0v[b] : 0.0;
                S_bc += b < B ? ev[b] * s1v[b] : 0.0;
            }
        }
#pragma unroll 1
        for (int lik = 0; lik < 2; ++lik) {
            if (lik == 1 && !use_bc) break;
            if (lik == 0) BEAN_STAMP_AT(2);
            else BEAN_STAMP_AT(5);
            const float* xp = xs + lik * B * 64 + lane;  // xp[b * 64]
            const double* sf = (lik ? c.sf_bc : c.sf) + r * B;
            // n = sum x_b is data: k_prepare leaves it in nobs (-1 where the (rep, guide) is masked)
            const double nn = ms[lik * 64];
            if (nn < 0.0) continue;
            const double S = lik ? S_bc : S_x;  // pass 1 (S = sum e_b sf_b) was done for both at once
            if (lik == 0) BEAN_STAMP_AT(3);
            const double a0 = ms[(2 + lik) * 64];
            const double inv = frcp(S + kEps);
            // pass 2 (see k_lik): U_Q = sum k_b Q_b, V_Q = sum dpsi_b k_b Q_b, t_Q = sum sf_b Q_b
            double A0 = 0.0, lsum = 0.0, Ua = 0.0, Va = 0.0;
            double U_mu = 0.0, U_y = 0.0, U_0 = 0.0, U_1 = 0.0;
            double V_mu = 0.0, V_y = 0.0, V_0 = 0.0, V_1 = 0.0;
            double t_mu = 0.0, t_y = 0.0, t_0 = 0.0, t_1 = 0.0;
            double p0n = MIX ? uniform_ld(c.P0, 0) : 0.0, sfn = uniform_ld(sf, 0), smn = uniform_ld(sm, 0);
#pragma unroll 1
            for (int b = 0; b < B; ++b) {
                const double x = (double)xp[b * 64];
                const double p0 = p0n, sfb = sfn, smb = smn;
                const double p1 = tp[b * ntm], pmu = tp[(B + b) * ntm], py = tp[(2 * B + b) * ntm];
                if (b + 1 < B) {
                    if (MIX) p0n = uniform_ld(c.P0, b + 1);
                    sfn = uniform_ld(sf, b + 1);
                    smn = uniform_ld(sm, b + 1);
                }
                const double araw = ((w0 * p0 + w1 * p1) * sfb + epsB) * inv * a0 * smb;
                const bool floored = araw < kEps;
                const double alpha = floored ? kEps : araw;
                A0 += alpha;
                const DD db = lgamma_digamma_diff_inl(alpha, x);
                lsum += db.d;
                const double kb = floored ? 0.0 : a0 * smb * inv * sfb;
                const double kd = kb * db.dp;
                Ua += floored ? 0.0 : araw;
                Va += floored ? 0.0 : db.dp * araw;
                U_mu += kb * pmu;
                V_mu += kd * pmu;
                t_mu += sfb * pmu;
                U_y += kb * py;
                V_y += kd * py;
                t_y += sfb * py;
                U_1 += kb * p1;
                V_1 += kd * p1;
                t_1 += sfb * p1;
                if (MIX) {
                    U_0 += kb * p0;
                    V_0 += kd * p0;
                    t_0 += sfb * p0;
                }
            }
            if (lik == 0) BEAN_STAMP_AT(4);
            const DD d0 = lgamma_digamma_diff_inl(A0, nn);
            nll += d0.d - lsum;
            const double W = (d0.dp * Ua - Va) * inv;
            a_mu += w1 * (d0.dp * U_mu - V_mu - W * t_mu);
            a_y += w1 * (d0.dp * U_y - V_y - W * t_y);
            g0 += d0.dp * U_0 - V_0 - W * t_0;
            g1 += d0.dp * U_1 - V_1 - W * t_1;
        }
        BEAN_STAMP_AT(6);
        double* row = c.wrow + rgi;  // row q of this replicate at row[q * RG]
        row[kPGmu * RG] = a_mu;
        row[kPGy * RG] = a_y;
        if (MIX) {
            // d loss / d pi through the likelihood
            double gpi0 = g0, gpi1 = g1;
            if (ACC) {
                gpi0 = 0.0;
                gpi1 = (g1 - g0) * dpe1_dpi1;
            }
            if (ACC) row[kPGnoise * RG] = (g1 - g0) * dpe1_dl;
            const double lpi0 = flog(pi0), lpi1 = flog(pi1);
            const double rpi0 = frcp(pi0), rpi1 = frcp(pi1);
            if (rgm) {
                // Multinomial(probs = pi) on control allele counts (model.py:470-474):
                // torch renormalises the probabilities and clamps them to [eps, 1 - eps]
                const double s = pi0 + pi1;
                const double ls = s == 1.0 ? 0.0 : flog(s);
                const double rs = s == 1.0 ? 1.0 : frcp(s);
                const double cnt0 = ms[4 * 64], cnt1 = ms[5 * 64];
                const double pr0 = pi0 * rs, pr1 = pi1 * rs;
                const bool in0 = pr0 > kProbEps && pr0 < 1.0 - kProbEps;
                const bool in1 = pr1 > kProbEps && pr1 < 1.0 - kProbEps;
                const double lg0 = in0 ? lpi0 - ls : flog(fmin(fmax(pr0, kProbEps), 1.0 - kProbEps));
                const double lg1 = in1 ? lpi1 - ls : flog(fmin(fmax(pr1, kProbEps), 1.0 - kProbEps));
                nll -= cnt0 * lg0;
                nll -= cnt1 * lg1;
                if (in0) gpi0 -= cnt0 * rpi0;
                if (in1) gpi1 -= cnt1 * rpi1;
                gpi0 -= (cp0 - 1.0) * rpi0;
                gpi1 -= (cp1 - 1.0) * rpi1;
            }
            // (the number of unmasked replicates of the guide is data: k_prepare keeps it in part[kPNrg])
            row[kPLp * RG] = rgm ? lpi0 : 0.0;
            row[(kPLp + 1) * RG] = rgm ? lpi1 : 0.0;
            row[kPLq * RG] = lpi0;
            row[(kPLq + 1) * RG] = lpi1;
            gpi0 += (cq0 - 1.0) * rpi0;
            gpi1 += (cq1 - 1.0) * rpi1;
            const double proj = pi0 * gpi0 + pi1 * gpi1;
            const double total = cq0 + cq1;
            double path0 = 0.0, path1 = 0.0;
#pragma unroll 1
            for (int a = 0; a < 2; ++a) {
                if (a ? cl1 : cl0) continue;
                const double v = dirichlet_grad_one(a ? pi1 : pi0, a ? cq1 : cq0, total) *
                                 ((a ? gpi1 : gpi0) - proj);
                path0 = a ? path0 : v;
                path1 = a ? v : path1;
            }
            row[kPPath * RG] = path0;
            row[(kPPath + 1) * RG] = path1;
        }
        loss = nll;
    }
    const double tot = wave_sum(loss);
    if (lane == 0) {
        loss_add(c, ctr.slot, tot);
        if (blockIdx.x == 0 && blockIdx.y == 0) publish_ctr(c, ctr);
    }
    BEAN_STAMP_AT(7);
}

// ------------------------------------------------- split form (variant, diagnostic A/B)
// BEAN_HIP_GUIDE=split: the same work as k_guide_wave in three launches, each at
// its own occupancy:
//   k_sample_pi   per (rep, guide): Dirichlet draw                      -> pi_ws
//   k_lik         per (rep, likelihood, guide): one Dirichlet-Multinomial term
//                 with analytic gradients (X and X_bcmatch run as separate
//                 waves; everything downstream is linear in d nll / d e[b])
//                 -> per-guide d/dmu_t, d/dy_t rows, d nll / d pi per (rep, guide)
//   k_pi_terms    per (rep, guide): Multinomial on control allele counts,
//                 Dirichlet log-prob pieces, implicit-reparameterisation gradient
__global__ __launch_bounds__(256) void k_sample_pi(DevArgs c) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)c.R * c.G) return;
    const int r = (int)(idx / c.G), g = (int)(idx % c.G);
    double pi0, pi1;
    if (c.pi_in) {
        pi0 = c.pi_in[idx * 2];
        pi1 = c.pi_in[idx * 2 + 1];
    } else {
        const double al0 = (double)expf(c.p[4][2 * g]), al1 = (double)expf(c.p[4][2 * g + 1]);
        const double rs = frcp(al0 + al1) * c.pi_a0[g];
        const double cq0 = fmax(al0 * rs, 1e-5), cq1 = fmax(al1 * rs, 1e-5);
        Rng rng(c.seed, kSitePi, (unsigned long long)r * c.G_tot + (c.g_off + g), c.ctrB->step * 256ull);
        const GammaPair gp = sample_gamma_pair(cq0, cq1, rng);
        const double gm0 = fmax(gp.g0, kDblMin), gm1 = fmax(gp.g1, kDblMin);
        const double rsum = frcp(gm0 + gm1);
        pi0 = fmin(fmax(gm0 * rsum, kDblMin), kOneMinus);
        pi1 = fmin(fmax(gm1 * rsum, kDblMin), kOneMinus);
    }
    c.pi_ws[idx * 2] = pi0;
    c.pi_ws[idx * 2 + 1] = pi1;
    if (c.flags & kDumpPi) {
        c.pi_out[idx * 2] = pi0;
        c.pi_out[idx * 2 + 1] = pi1;
    }
}

// grid = (ceil(G / 64), R); blockDim.x = 64 * nlik: one wave per likelihood (X, X_bcmatch) of
// 64 consecutive guides of one replicate.  Small blocks keep the CUs evenly filled (a block of
// all replicates x likelihoods would be 10 waves and fit once per CU at 128 VGPRs).
//
// Register diet: the loops over conditions are ROLLED and keep scalar state only.
// alpha_b is recomputed from (pi, P[b], sf[b]) where needed, and because
// d nll / d alpha_b = psi-diff(A0, n) - psi-diff(alpha_b, x_b) enters every output
// linearly, the A0 term is factored out and applied after the loop; so one pass
// yields the likelihood and all gradients without per-condition arrays.
template <bool MIX, bool ACC>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4)))
void k_lik(DevArgs c) {
    __shared__ double ex[2][5][64];  // per wave: gpi0, gpi1, d/dmu, d/dy, d/dnoise
    __shared__ double scratch[16];
    // Every global read of the block is issued in one batch up front (the kernel is latency
    // bound: one exposed memory round trip instead of one per loop iteration) and staged in LDS,
    // from where the rolled loops read: the Phi tables of the 64 guides' targets, shared by
    // the likelihood waves, and each thread's own counts.
    __shared__ double tabs[3][kBMax][64];   // P, dP/dmu, dP/dy of this guide's target
    __shared__ float xs[2][kBMax][64];
    const int lane = threadIdx.x & 63;
    const int lik = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nlik = blockDim.x >> 6;
    const int g = blockIdx.x * 64 + lane;
    const int r = blockIdx.y;
    const bool valid = g < c.G;
    const int G = c.G, T = c.T, B = c.B;
    const long rgi = (long)r * G + g;
    double a_mu = 0.0, a_y = 0.0, a_noise = 0.0, loss = 0.0, gp0 = 0.0, gp1 = 0.0;
#ifdef BEAN_STAMP
    const long wave_gid = ((long)blockIdx.y * gridDim.x + blockIdx.x) * nlik + lik;
#endif
    BEAN_STAMP_AT(0);

    {
        const int t = valid ? c.g2t[g] : 0;
        const float* xq = (lik ? c.Xbc : c.X) + (long)r * B * G + g;
        // the 3 * B table entries of a guide are split between the likelihood waves
        const int nq = 3 * B, q0 = lik * ((nq + nlik - 1) / nlik);
        const int q1 = nlik == 1 ? nq : (lik == 0 ? (nq + 1) / 2 : nq);
        if (valid) {
            // fully unrolled and predicated so that all loads are in flight together
            float xv[kBMax];
            double tv[3 * kBMax];  // nlik == 1 stages all 3 B entries; two waves split them
#pragma unroll
            for (int b = 0; b < kBMax; ++b) xv[b] = b < B ? xq[(long)b * G] : 0.f;
#pragma unroll
            for (int k = 0; k < 3 * kBMax; ++k) {
                const int q = q0 + k;
                tv[k] = 0.0;
                if (q < q1) {
                    const int which = q / B, b = q - which * B;
                    const double* tab = which == 0 ? c.tabP : (which == 1 ? c.tabPmu : c.tabPy);
                    tv[k] = tab[(long)b * T + t];
                }
            }
#pragma unroll
            for (int b = 0; b < kBMax; ++b)
                if (b < B) xs[lik][b][lane] = xv[b];
#pragma unroll
            for (int k = 0; k < 3 * kBMax; ++k) {
                const int q = q0 + k;
                if (q < q1) {
                    const int which = q / B, b = q - which * B;
                    tabs[which][b][lane] = tv[k];
                }
            }
        }
    }
    BEAN_STAMP_AT(1);
    __syncthreads();
    BEAN_STAMP_AT(2);
    if (valid) {
        const double a0 = lik ? c.a0_bc[g] : c.a0[g];
        const double* sf = (lik ? c.sf_bc : c.sf) + r * B;
        const double* sm = c.smask + r * B;
        const double epsB = kEps / (double)B;
        double pi0 = 0.0, pi1 = 1.0, pe1 = 1.0, dpe1_dpi1 = 0.0, dpe1_dl = 0.0;
        if (MIX) {
            pi0 = c.pi_ws[rgi * 2];
            pi1 = c.pi_ws[rgi * 2 + 1];
            pe1 = pi1;
            if (ACC) {
                // scale_pi_by_accessibility + add_noise_to_pi, A = 2 (utils.py:106-178)
                const double kacc = c.kacc[g];
                const double s1 = pi1 * kacc;
                const bool in1 = s1 > 1e-3 && s1 < 1.0 - 1e-3;
                const double p1c = fmin(fmax(s1, 1e-3), 1.0 - 1e-3);
                const double l = flog(p1c * frcp(1.0 - p1c)) + c.lpn[g];
                const double el = exp(l);
                const double pn = el * frcp(1.0 + el);
                const bool in2 = pn > 1e-3 && pn < 1.0 - 1e-3;
                pe1 = fmin(fmax(pn, 1e-3), 1.0 - 1e-3);
                dpe1_dl = in2 ? pn * (1.0 - pn) : 0.0;
                dpe1_dpi1 = in1 ? dpe1_dl * frcp(p1c * (1.0 - p1c)) * kacc : 0.0;
            }
        }
        const double w0 = MIX ? (ACC ? 1.0 - pe1 : pi0) : 0.0;  // weight of the wild-type component
        const double w1 = MIX ? (ACC ? pe1 : pi1) : 1.0;        // weight of the edited component
        // pass 1: n = sum x_b, S = sum e_b sf_b
        double S = 0.0, nn = 0.0;
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const double p1 = tabs[0][b][lane];
            nn += (double)xs[lik][b][lane];
            S += (w0 * c.P0[b] + w1 * p1) * sf[b];
        }
        BEAN_STAMP_AT(3);
        const bool obs = c.rg[rgi] != 0 && nn > (double)c.mask_thres;
        double g0 = 0.0, g1 = 0.0;
        if (obs) {
            const double inv = frcp(S + kEps);
            // pass 2.  With k_b = a0 m_b inv sf_b (0 where alpha_b sits on its floor):
            //   U_Q = sum k_b Q_b, V_Q = sum dpsi_b k_b Q_b, t_Q = sum sf_b Q_b   (Q in {Pmu, Py, P0, P1})
            //   Ua = sum_unfloored alpha_b,           Va = sum dpsi_b alpha_b
            double A0 = 0.0, nll = 0.0, Ua = 0.0, Va = 0.0;
            double U_mu = 0.0, U_y = 0.0, U_0 = 0.0, U_1 = 0.0;
            double V_mu = 0.0, V_y = 0.0, V_0 = 0.0, V_1 = 0.0;
            double t_mu = 0.0, t_y = 0.0, t_0 = 0.0, t_1 = 0.0;
#pragma unroll 1
            for (int b = 0; b < B; ++b) {
                const double p1 = tabs[0][b][lane], pmu = tabs[1][b][lane], py = tabs[2][b][lane];
                const double p0 = MIX ? c.P0[b] : 0.0;
                const double x = (double)xs[lik][b][lane];
                const double sfb = sf[b], smb = sm[b];
                const double araw = ((w0 * p0 + w1 * p1) * sfb + epsB) * inv * a0 * smb;
                const bool floored = araw < kEps;
                const double alpha = floored ? kEps : araw;
                A0 += alpha;
                const DD db = lgamma_digamma_diff_inl(alpha, x);
                nll -= db.d;
                const double kb = floored ? 0.0 : a0 * smb * inv * sfb;
                const double kd = kb * db.dp;
                Ua += floored ? 0.0 : araw;
                Va += floored ? 0.0 : db.dp * araw;
                U_mu += kb * pmu;
                V_mu += kd * pmu;
                t_mu += sfb * pmu;
                U_y += kb * py;
                V_y += kd * py;
                t_y += sfb * py;
                U_1 += kb * p1;
                V_1 += kd * p1;
                t_1 += sfb * p1;
                if (MIX) {
                    U_0 += kb * p0;
                    V_0 += kd * p0;
                    t_0 += sfb * p0;
                }
            }
            BEAN_STAMP_AT(4);
            const DD d0 = lgamma_digamma_diff_inl(A0, nn);
            nll += d0.d;
            // ga_b = d0.dp - dpsi_b  =>  sum ga_b k_b Q_b = d0.dp U_Q - V_Q,  W = (d0.dp Ua - Va) inv
            const double W = (d0.dp * Ua - Va) * inv;
            a_mu = w1 * (d0.dp * U_mu - V_mu - W * t_mu);
            a_y = w1 * (d0.dp * U_y - V_y - W * t_y);
            g0 = d0.dp * U_0 - V_0 - W * t_0;
            g1 = d0.dp * U_1 - V_1 - W * t_1;
            loss = nll;
        }
        if (MIX) {
            gp0 = g0;
            gp1 = g1;
            if (ACC) {
                gp0 = 0.0;
                gp1 = (g1 - g0) * dpe1_dpi1;
                a_noise = (g1 - g0) * dpe1_dl;
            }
        }
    }
    BEAN_STAMP_AT(5);
    // ---- sum the likelihood waves (fixed order) and write this replicate's rows
    if (nlik == 2) {
        ex[lik][0][lane] = gp0;
        ex[lik][1][lane] = gp1;
        ex[lik][2][lane] = a_mu;
        ex[lik][3][lane] = a_y;
        ex[lik][4][lane] = a_noise;
        __syncthreads();
        if (lik == 0) {
            gp0 += ex[1][0][lane];
            gp1 += ex[1][1][lane];
            a_mu += ex[1][2][lane];
            a_y += ex[1][3][lane];
            a_noise += ex[1][4][lane];
        }
    }
    if (lik == 0 && valid) {
        if (MIX) {
            c.gpi_ws[rgi * 2] = gp0;
            c.gpi_ws[rgi * 2 + 1] = gp1;
        }
        c.rrow[((long)kPGmu * c.R + r) * G + g] = a_mu;
        c.rrow[((long)kPGy * c.R + r) * G + g] = a_y;
        if (MIX && ACC) c.rrow[((long)kPGnoise * c.R + r) * G + g] = a_noise;
    }
    BEAN_STAMP_AT(6);
    const double tot = block_sum(loss, scratch);
    BEAN_STAMP_AT(7);
    if (threadIdx.x == 0) {
        loss_add(c, c.ctrB->slot, tot);
        if (blockIdx.x == 0 && blockIdx.y == 0) publish_ctr(c, *c.ctrB);
    }
}

// blockDim.x = 64 * nw (replicates); dynamic LDS = nw * 7 * 64 doubles + 16.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
void k_pi_terms(DevArgs c) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int g = blockIdx.x * 64 + lane;
    const bool valid = g < c.G;
    const int G = c.G;
    // rows: nrg, path0, path1, Lp0, Lp1, Lq0, Lq1
    double acc[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double loss = 0.0;
    if (valid) {
        const double al0 = (double)expf(c.p[4][2 * g]), al1 = (double)expf(c.p[4][2 * g + 1]);
        const double rs = frcp(al0 + al1) * c.pi_a0[g];
        const double cp[2] = {al0 * rs, al1 * rs};
        const bool cl[2] = {cp[0] < 1e-5, cp[1] < 1e-5};
        const double cq[2] = {cl[0] ? 1e-5 : cp[0], cl[1] ? 1e-5 : cp[1]};
        const double total = cq[0] + cq[1];
        for (int r = w; r < c.R; r += nw) {
            const long rgi = (long)r * G + g;
            const bool rgm = c.rg[rgi] != 0;
            const double pi[2] = {c.pi_ws[rgi * 2], c.pi_ws[rgi * 2 + 1]};
            double gpi[2] = {c.gpi_ws[rgi * 2], c.gpi_ws[rgi * 2 + 1]};
            const double lpi[2] = {flog(pi[0]), flog(pi[1])};
            const double rpi[2] = {frcp(pi[0]), frcp(pi[1])};
            if (rgm) {
                const double s = pi[0] + pi[1];
                const double ls = s == 1.0 ? 0.0 : flog(s);
                const double rsum = s == 1.0 ? 1.0 : frcp(s);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const double pr = pi[a] * rsum;
                    const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
                    const double lg = inside ? lpi[a] - ls : flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                    double cnt = 0.0;
                    for (int cc = 0; cc < c.C; ++cc)
                        cnt += (double)c.allele[(((long)r * c.C + cc) * G + g) * 2 + a];
                    loss -= cnt * lg;
                    if (inside) gpi[a] -= cnt * rpi[a];
                }
                acc[0] += 1.0;
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (rgm) {
                    acc[3 + a] += lpi[a];
                    gpi[a] -= (cp[a] - 1.0) * rpi[a];
                }
                acc[5 + a] += lpi[a];
                gpi[a] += (cq[a] - 1.0) * rpi[a];
            }
            const double proj = pi[0] * gpi[0] + pi[1] * gpi[1];
            double path0 = 0.0, path1 = 0.0;
#pragma unroll 1
            for (int a = 0; a < 2; ++a) {
                const bool skip = a ? cl[1] : cl[0];
                if (skip) continue;
                const double v = dirichlet_grad_one(a ? pi[1] : pi[0], a ? cq[1] : cq[0], total) *
                                 ((a ? gpi[1] : gpi[0]) - proj);
                path0 = a ? path0 : v;
                path1 = a ? v : path1;
            }
            acc[1] += path0;
            acc[2] += path1;
        }
    }
    double* red = lds;  // [nw][7][64]
#pragma unroll
    for (int q = 0; q < 7; ++q) red[((long)w * 7 + q) * 64 + lane] = acc[q];
    __syncthreads();
    if (valid) {
        for (int q = w; q < 7; q += nw) {
            double s = 0.0;
            for (int ww = 0; ww < nw; ++ww) s += red[((long)ww * 7 + q) * 64 + lane];
            const int row = q == 0 ? kPNrg : (q < 3 ? kPPath + (q - 1) : (q < 5 ? kPLp + (q - 3) : kPLq + (q - 5)));
            c.part[(long)row * G + g] = s;
        }
    }
    double* scratch = lds + (long)nw * 7 * 64;
    const double tot = block_sum(loss, scratch);
    if (threadIdx.x == 0) loss_add(c, c.ctrB->slot, tot);
}

#endif  // BEAN_AB_KERNELS
// ------------------------------------------------------------ survival kernels
// survival NormalModel: sq[r] = sum_g q_0[r, g] * gq[r, g] (fixed order: strided partials, block tree)
__global__ __launch_bounds__(1024) void k_sum_q(DevArgs c) {
    __shared__ double scratch[16];
    const int r = blockIdx.x;
    double v = 0.0;
    for (int g = threadIdx.x; g < c.G; g += blockDim.x) {
        const double gm = c.gam[(long)r * c.G + g];
        const double x = c.x0_in ? gm
                                 : (double)fminf(fmaxf((float)(gm * frcp(c.gsum[r])), 1.17549435e-38f), 0.99999994f);
        v += x * c.gq[(long)r * c.G + g];
    }
    const double tot = block_sum(v, scratch);
    if (threadIdx.x == 0) c.sq[r] = tot;
}

#ifdef BEAN_AB_KERNELS
// Survival analogue of k_guide (survival_model.py:133-424): component "bin
// probabilities" are exp(mu_a * t_b) with mu = [u_g, u_g + mu_t]; the control
// Multinomial sees the alleles after selection up to the control timepoint; the
// Dirichlet(q0) site over all guides is handled per (rep, guide) as well.
template <int B, int FAM, bool ACC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BEAN_GUIDE_WAVES_PER_EU)))
void k_guide_survival(DevArgs c) {
    extern __shared__ double lds[];
    constexpr bool MIX = FAM == kMixture;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int g = blockIdx.x * 64 + lane;
    const bool valid = g < c.G;
    const StepCtr ctr = *c.ctrB;
    const int G = c.G;

    double acc[kNumPart];
#pragma unroll
    for (int q = 0; q < kNumPart; ++q) acc[q] = 0.0;
    double loss = 0.0;

    if (valid) {
        const bool use_bc = (c.flags & kUseBc) != 0;
        const bool q0lik = !MIX && c.surv_q0lik;
        // survival NormalModel: mu of negative-control guides is forced to 0 (survival_model.py:59-60)
        const bool negc = q0lik && c.negctrl && c.negctrl[g] != 0;
        const double mu_t = negc ? 0.0 : c.mu_t[c.g2t[g]];
        const double u = MIX ? c.u_g[g] : 0.0;
        const double mu1 = u + mu_t;
        const double ia = q0lik ? (double)expf(c.p[7][g]) : 0.0;
        double cp[2] = {1.0, 1.0}, cq[2] = {1.0, 1.0};
        bool cl[2] = {false, false};
        double kacc = 0.0, lpn = 0.0, q0 = 0.0;
        if (MIX) {
            const double al0 = (double)expf(c.p[4][2 * g]), al1 = (double)expf(c.p[4][2 * g + 1]);
            const double rs = frcp(al0 + al1) * c.pi_a0[g];
            cp[0] = al0 * rs;
            cp[1] = al1 * rs;
            cl[0] = cp[0] < 1e-5;
            cl[1] = cp[1] < 1e-5;
            cq[0] = cl[0] ? 1e-5 : cp[0];
            cq[1] = cl[1] ? 1e-5 : cp[1];
            q0 = (double)expf(c.p[7][g]);
            if (ACC) {
                kacc = c.kacc[g];
                lpn = c.lpn[g];
            }
        }
        for (int r = w; r < c.R; r += nw) {
            const bool rgm = c.rg[(long)r * G + g] != 0;
            double pi[2] = {0.0, 1.0}, pe1 = 1.0, dpe1_dpi1 = 0.0, dpe1_dl = 0.0;
            if (MIX) {
                if (c.pi_in) {
                    pi[0] = c.pi_in[((long)r * G + g) * 2];
                    pi[1] = c.pi_in[((long)r * G + g) * 2 + 1];
                } else {
                    Rng rng(c.seed, kSitePi, (unsigned long long)r * c.G_tot + (c.g_off + g), ctr.step * 256ull);
                    const GammaPair gp = sample_gamma_pair(cq[0], cq[1], rng);
                    const double gm0 = fmax(gp.g0, kDblMin), gm1 = fmax(gp.g1, kDblMin);
                    const double rs = frcp(gm0 + gm1);
                    pi[0] = fmin(fmax(gm0 * rs, kDblMin), kOneMinus);
                    pi[1] = fmin(fmax(gm1 * rs, kDblMin), kOneMinus);
                }
                if (c.flags & kDumpPi) {
                    c.pi_out[((long)r * G + g) * 2] = pi[0];
                    c.pi_out[((long)r * G + g) * 2 + 1] = pi[1];
                }
                pe1 = pi[1];
                if (ACC) {
                    const double s1 = pi[1] * kacc;
                    const bool in1 = s1 > 1e-3 && s1 < 1.0 - 1e-3;
                    const double p1c = fmin(fmax(s1, 1e-3), 1.0 - 1e-3);
                    const double l = flog(p1c * frcp(1.0 - p1c)) + lpn;
                    const double el = exp(l);
                    const double pn = el * frcp(1.0 + el);
                    const bool in2 = pn > 1e-3 && pn < 1.0 - 1e-3;
                    pe1 = fmin(fmax(pn, 1e-3), 1.0 - 1e-3);
                    dpe1_dl = in2 ? pn * (1.0 - pn) : 0.0;
                    dpe1_dpi1 = in1 ? dpe1_dl * frcp(p1c * (1.0 - p1c)) * kacc : 0.0;
                }
            }
            double x0 = 1.0;
            if (q0lik) {
                const double gm = c.gam[(long)r * G + g];
                x0 = c.x0_in ? gm
                             : (double)fminf(fmaxf((float)(gm * frcp(c.gsum[r])), 1.17549435e-38f), 0.99999994f);
                if (c.x0_out) c.x0_out[(long)r * G + g] = x0;
            }
            double e[B], ge[B], P0[B], P1[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const double tb = c.time[b];
                P1[b] = exp(mu1 * tb);
                P0[b] = MIX ? exp(u * tb) : 0.0;
                e[b] = MIX ? (ACC ? (1.0 - pe1) * P0[b] + pe1 * P1[b] : pi[0] * P0[b] + pi[1] * P1[b])
                           : (q0lik ? P1[b] * x0 : P1[b]);
                ge[b] = 0.0;
            }
            double nll = 0.0;
            const double* sm = c.smask + r * B;
#pragma unroll 1
            for (int lik = 0; lik < 2; ++lik) {
                if (lik == 1 && !use_bc) break;
                const float* xp = (lik ? c.Xbc : c.X) + (long)r * B * G + g;
                float n = 0.f;
#pragma unroll
                for (int b = 0; b < B; ++b) n += xp[(long)b * G];
                if (rgm && n > (float)c.mask_thres)
                    nll += dirmult_nll<B>(xp, (long)G, (lik ? c.sf_bc : c.sf) + r * B, sm,
                                          lik ? c.a0_bc[g] : c.a0[g], e, ge);
            }
            double dmu = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
            for (int b = 0; b < B; ++b) {
                dmu += ge[b] * c.time[b] * P1[b];
                g0 += ge[b] * P0[b];
                g1 += ge[b] * P1[b];
            }
            double gmu = pe1 * dmu;
            if (q0lik) {
                gmu = negc ? 0.0 : dmu * x0;
                // - log p(q_0) + log q(q_0): (ia - 1/G) log x per guide; normalisers in k_param
                const double dconc = ia - (c.prior_ia ? c.prior_ia[g] : (double)(1.0f / (float)c.G_tot));
                const double lx = flog(x0);
                nll += dconc * lx;
                c.gq[(long)r * G + g] = g1 + dconc * frcp(x0);
                acc[kPQ0] += lx;
            }
            if (MIX) {
                double gpi[2] = {g0, g1};
                if (ACC) {
                    gpi[0] = 0.0;
                    gpi[1] = (g1 - g0) * dpe1_dpi1;
                    acc[kPGnoise] += (g1 - g0) * dpe1_dl;
                }
                const double lpi[2] = {flog(pi[0]), flog(pi[1])};
                const double rpi[2] = {frcp(pi[0]), frcp(pi[1])};
                if (rgm) {
                    // control_allele_count ~ Multinomial(pi * exp(mu * t_ctrl)) (survival_model.py:326-346)
                    for (int cc = 0; cc < c.C; ++cc) {
                        const double tc = c.ctrl_time[cc];
                        const double gr[2] = {exp(u * tc), exp(mu1 * tc)};
                        const double wv[2] = {pi[0] * gr[0], pi[1] * gr[1]};
                        const double rW = frcp(wv[0] + wv[1]);
                        double n_in = 0.0, cnt[2];
                        bool inside[2];
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            const double pr = wv[a] * rW;
                            inside[a] = pr > kProbEps && pr < 1.0 - kProbEps;
                            cnt[a] = (double)c.allele[(((long)r * c.C + cc) * G + g) * 2 + a];
                            nll -= cnt[a] * flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                            if (inside[a]) n_in += cnt[a];
                        }
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            gpi[a] += ((inside[a] ? -cnt[a] * frcp(wv[a]) : 0.0) + n_in * rW) * gr[a];
                        gmu += ((inside[1] ? -cnt[1] : 0.0) + n_in * wv[1] * rW) * tc;
                    }
                    acc[kPNrg] += 1.0;
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (rgm) {
                        acc[kPLp + a] += lpi[a];
                        gpi[a] -= (cp[a] - 1.0) * rpi[a];
                    }
                    acc[kPLq + a] += lpi[a];
                    gpi[a] += (cq[a] - 1.0) * rpi[a];
                }
                const double proj = pi[0] * gpi[0] + pi[1] * gpi[1];
                const double total = cq[0] + cq[1];
                double path0 = 0.0, path1 = 0.0;
#pragma unroll 1
                for (int a = 0; a < 2; ++a) {
                    const bool skip = a ? cl[1] : cl[0];
                    if (skip) continue;
                    const double v = dirichlet_grad_one(a ? pi[1] : pi[0], a ? cq[1] : cq[0], total) *
                                     ((a ? gpi[1] : gpi[0]) - proj);
                    path0 = a ? path0 : v;
                    path1 = a ? v : path1;
                }
                acc[kPPath] += path0;
                acc[kPPath + 1] += path1;
                // ---- Dirichlet(q0) site: +log q(x) of the guide's draw, -log p(obs) of the model
                {
                    const double gm = c.gam[(long)r * G + g];
                    // float32 semantics of torch's draw: normalise, clamp to [FLT_MIN, 1 - 2^-24]
                    const double x = c.x0_in ? gm
                                             : (double)fminf(fmaxf((float)(gm * frcp(c.gsum[r])), 1.17549435e-38f),
                                                             0.99999994f);
                    if (c.x0_out) c.x0_out[(long)r * G + g] = x;
                    const double lx = flog(x), lobs = c.log_obs0[(long)r * G + g];
                    const double tot0 = c.gsum[c.R];
                    nll += (q0 - 1.0) * (lx - lobs);
                    const double gout = (q0 - 1.0) * frcp(x);
                    const double S = tot0 - (double)c.G_tot;  // sum_g x_g * gout_g
                    acc[kPQ0] += lx - lobs + dirichlet_grad_one(x, q0, tot0) * (gout - S);
                }
            }
            acc[kPGmu] += gmu;
            loss += nll;
        }
    }

    double* red = lds;  // [nw][kNumPart][64]
    if (nw > 1) {
#pragma unroll
        for (int q = 0; q < kNumPart; ++q) red[((long)w * kNumPart + q) * 64 + lane] = acc[q];
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int q = 0; q < kNumPart; ++q) {
                double s = acc[q];
                for (int ww = 1; ww < nw; ++ww) s += red[((long)ww * kNumPart + q) * 64 + lane];
                acc[q] = s;
            }
        }
    }
    if (w == 0 && valid) {
#pragma unroll
        for (int q = 0; q < kNumPart; ++q)
            if (MIX || q < 2) c.part[(long)q * G + g] = acc[q];
    }
    double* scratch = lds + (long)nw * kNumPart * 64;
    const double tot = block_sum(loss, scratch);
    if (threadIdx.x == 0) {
        loss_add(c, ctr.slot, tot);
        if (blockIdx.x == 0) publish_ctr(c, ctr);
    }
}

#endif  // BEAN_AB_KERNELS
// ------------------------------------------------------------------- k_allele
// (allele_slot_tables: above k_param, whose allele blocks share it)
// One thread per slot that holds an allele (DevArgs::live_slots, built once by bean_hip_prepare: the
// slots in (a1, g) order whose mask is set or whose edit list is not empty).  The other slots' tables are
// zero from the start and stay zero.  At BASELINE config 3 that is 193k of 350k slots, and because the
// empty ones are scattered (a wave of 64 consecutive guides at slot a1 nearly always held a few alleles)
// the launch ran every wave for half the lanes.
__global__ __launch_bounds__(256) void k_allele(DevArgs c) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
#if defined(BEAN_STAMP) && BEAN_STAMP == 5  // (records behind k_param's: edit blocks, guide blocks)
    const long rec0 = ((long)c.T * c.lpt + kParamBlock - 1) / kParamBlock + ((long)c.G * kAMax + kParamBlock - 1) / kParamBlock;
    BEAN_STAMP_RT(rec0 + blockIdx.x, 0);
    if (idx < c.n_live_slots) {
        const int s5 = c.live_slots[idx];
        allele_slot_tables(c, s5 / c.G, s5 % c.G);
    }
    __syncthreads();
    BEAN_STAMP_RT(rec0 + blockIdx.x, 7);
    return;
#endif
    if (idx >= c.n_live_slots) return;
    const int s = c.live_slots[idx];  // a1 * G + g
    allele_slot_tables(c, s / c.G, s % c.G);
}

#if defined(BEAN_AB_KERNELS) && BEAN_AMAX <= 8  // the block form: an A/B reference of the default allele count only
// ------------------------------------------------------------- k_guide_tiling
// MultiMixtureNormal per (rep, guide): A-component Dirichlet draw, mixture over
// the guide's alleles, both DirMult terms, Multinomial on control allele counts,
// implicit gradient.  Static loops over kAMax components, predicated by a < A.
// dynamic LDS = kTNumPart * 64 doubles + 16.
// SURV: tiling survival screens (growth instead of sorting bins), a compile-time switch so that the
// sorting instantiation carries none of its state.
template <int B, bool ACC, bool SURV>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BEAN_GUIDE_WAVES_PER_EU)))
void k_guide_tiling(DevArgs c) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int g = blockIdx.x * 64 + lane;
    const bool valid = g < c.G;
    const StepCtr ctr = *c.ctrB;
    const int G = c.G, A = c.A, A1 = c.A - 1;

    double path[kAMax], L[kAMax], gmu_s[kAMax - 1], gsig_s[kAMax - 1];
#pragma unroll
    for (int a = 0; a < kAMax; ++a) {
        path[a] = 0.0;
        L[a] = 0.0;
        if (a < kAMax - 1) {
            gmu_s[a] = 0.0;
            gsig_s[a] = 0.0;
        }
    }
    double gnoise = 0.0, nrg = 0.0, loss = 0.0;

    if (valid) {
        const bool use_bc = (c.flags & kUseBc) != 0;
        double cq[kAMax], total = 0.0;
        {
            double alpha[kAMax], S = 0.0;
#pragma unroll
            for (int a = 0; a < kAMax; ++a) {
                const bool am = a < A && c.amask[(long)g * A + a] != 0;
                alpha[a] = a < A ? (am ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
                S += alpha[a];
            }
            const double rs = frcp(S) * c.pi_a0[g];
#pragma unroll
            for (int a = 0; a < kAMax; ++a) {
                cq[a] = alpha[a] * rs;
                // the survival tiling guide clamps its concentration (survival_model.py:813-821)
                if (SURV && a < A && cq[a] < 1e-5) cq[a] = 1e-5;
                total += cq[a];
            }
        }
        double kacc = 0.0, lpn = 0.0;
        if (ACC) {
            kacc = c.kacc[g];
            lpn = c.lpn[g];
        }
        // unedited allele: the sorting bins' P0[b], or the guide's baseline growth exp(u_g t_b)
        const double u = SURV ? c.u_g[g] : 0.0;
        double P0g[B];
#pragma unroll
        for (int b = 0; b < B; ++b) P0g[b] = SURV ? exp(u * c.time[b]) : c.P0[b];
        for (int r = w; r < c.R; r += nw) {
            const bool rgm = c.rg[(long)r * G + g] != 0;
            // both pi sites, the Multinomial and the count likelihoods are masked by
            // repguide_mask in tiling (model.py:659,682,731; guide 941): nothing to do
            if (!rgm) {
                if (c.flags & kDumpPi)  // exported draws must stay a valid simplex point
                    for (int a = 0; a < A; ++a) c.pi_out[((long)r * G + g) * A + a] = 1.0 / A;
                continue;
            }
            double pi[kAMax], pe[kAMax], dpe_dpi[kAMax], dpe_dl[kAMax];
            if (c.pi_in) {
#pragma unroll
                for (int a = 0; a < kAMax; ++a) pi[a] = a < A ? c.pi_in[((long)r * G + g) * A + a] : 0.0;
            } else {
                Rng rng(c.seed, kSitePi, (unsigned long long)r * c.G_tot + guide_stream_id(c, g), ctr.step * 256ull);
                double sum = 0.0;
#pragma unroll
                for (int a = 0; a < kAMax; a += 2) {
                    pi[a] = 0.0;
                    pi[a + 1] = 0.0;
                    if (a < A) {  // components are drawn two at a time (one rejection loop per pair)
                        const GammaPair gp = sample_gamma_pair(cq[a], a + 1 < A ? cq[a + 1] : 1.0, rng);
                        rng.k = gp.k;
                        pi[a] = fmax(gp.g0, kDblMin);
                        sum += pi[a];
                        if (a + 1 < A) {
                            pi[a + 1] = fmax(gp.g1, kDblMin);
                            sum += pi[a + 1];
                        }
                    }
                }
                const double rs = frcp(sum);
#pragma unroll
                for (int a = 0; a < kAMax; ++a)
                    if (a < A) pi[a] = fmin(fmax(pi[a] * rs, kDblMin), kOneMinus);
            }
            if (c.flags & kDumpPi) {
#pragma unroll
                for (int a = 0; a < kAMax; ++a)
                    if (a < A) c.pi_out[((long)r * G + g) * A + a] = pi[a];
            }
            double pe0 = pi[0];
#pragma unroll
            for (int a = 1; a < kAMax; ++a) {
                pe[a] = pi[a];
                dpe_dpi[a] = 1.0;
                dpe_dl[a] = 0.0;
            }
            if (ACC) {
                double sum = 0.0;
#pragma unroll
                for (int a = 1; a < kAMax; ++a) {
                    if (a < A) {
                        const double s1 = pi[a] * kacc;
                        const bool in1 = s1 > 1e-3 && s1 < 1.0 - 1e-3;
                        const double p1c = fmin(fmax(s1, 1e-3), 1.0 - 1e-3);
                        const double l = flog(p1c * frcp(1.0 - p1c)) + lpn;
                        const double el = exp(l);
                        const double pn = el * frcp(1.0 + el);
                        const bool in2 = pn > 1e-3 && pn < 1.0 - 1e-3;
                        pe[a] = fmin(fmax(pn, 1e-3), 1.0 - 1e-3);
                        dpe_dl[a] = in2 ? pn * (1.0 - pn) : 0.0;
                        dpe_dpi[a] = in1 ? dpe_dl[a] * frcp(p1c * (1.0 - p1c)) * kacc : 0.0;
                        sum += pe[a];
                    }
                }
                pe0 = 1.0 - sum;
            }
            double e[B], ge[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                double v = pe0 * P0g[b];
#pragma unroll
                for (int a = 1; a < kAMax; ++a)
                    if (a < A) v += pe[a] * c.tabP[((long)b * A1 + (a - 1)) * G + g];
                e[b] = v;
                ge[b] = 0.0;
            }
            double nll = 0.0;
            const double* sm = c.smask + r * B;
#pragma unroll 1
            for (int lik = 0; lik < 2; ++lik) {
                if (lik == 1 && !use_bc) break;
                const float* xp = (lik ? c.Xbc : c.X) + (long)r * B * G + g;
                float n = 0.f;
#pragma unroll
                for (int b = 0; b < B; ++b) n += xp[(long)b * G];
                if (n > (float)c.mask_thres)
                    nll += dirmult_nll<B>(xp, (long)G, (lik ? c.sf_bc : c.sf) + r * B, sm,
                                          lik ? c.a0_bc[g] : c.a0[g], e, ge);
            }
            // back through the mixture
            double s0 = 0.0, gpi[kAMax];
#pragma unroll
            for (int b = 0; b < B; ++b) s0 += ge[b] * P0g[b];
            gpi[0] = ACC ? 0.0 : s0;
#pragma unroll
            for (int a = 1; a < kAMax; ++a) {
                gpi[a] = 0.0;
                if (a < A) {
                    double sa = 0.0, dm = 0.0, ds = 0.0;
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const long o = ((long)b * A1 + (a - 1)) * G + g;
                        sa += ge[b] * c.tabP[o];
                        dm += ge[b] * c.tabPmu[o];
                        ds += ge[b] * c.tabPy[o];
                    }
                    gmu_s[a - 1] += pe[a] * dm;
                    gsig_s[a - 1] += pe[a] * ds;
                    if (ACC) {
                        gpi[a] = (sa - s0) * dpe_dpi[a];
                        gnoise += (sa - s0) * dpe_dl[a];
                    } else {
                        gpi[a] = sa;
                    }
                }
            }
            // Multinomial on control allele counts + Dirichlet log-prob pieces
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < kAMax; ++a)
                if (a < A) s += pi[a];
            const double ls = s == 1.0 ? 0.0 : flog(s);
            const double rsum = s == 1.0 ? 1.0 : frcp(s);
            double proj = 0.0;
#pragma unroll
            for (int a = 0; a < kAMax; ++a) {
                if (a < A) {
                    const double lpi = flog(pi[a]), rpi = frcp(pi[a]);
                    if (!SURV) {
                        const double pr = pi[a] * rsum;
                        const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
                        const double lg = inside ? lpi - ls : flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                        double cnt = 0.0;
                        for (int cc = 0; cc < c.C; ++cc)
                            cnt += (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                        nll -= cnt * lg;
                        if (inside) gpi[a] -= cnt * rpi;
                    }
                    L[a] += lpi;
                    gpi[a] += (cq[a] - 1.0) * rpi;  // + d log q / d pi
                }
            }
            if (SURV) {
                // control_allele_count ~ Multinomial(pi * exp(mu * t_ctrl)), mu = [u, u + mu_a]
                // (survival_model.py:535-548): gradients to pi and, through the growth, to mu_a
                for (int cc = 0; cc < c.C; ++cc) {
                    const double tc = c.ctrl_time[cc];
                    double wv[kAMax], gr[kAMax], W = 0.0;
#pragma unroll
                    for (int a = 0; a < kAMax; ++a) {
                        gr[a] = 0.0;
                        wv[a] = 0.0;
                        if (a < A) {
                            const double m = a == 0 ? u : u + c.mu_a[(long)(a - 1) * G + g];
                            gr[a] = exp(m * tc);
                            wv[a] = pi[a] * gr[a];
                            W += wv[a];
                        }
                    }
                    const double rW = frcp(W);
                    double n_in = 0.0, cnt[kAMax];
                    bool inside[kAMax];
#pragma unroll
                    for (int a = 0; a < kAMax; ++a) {
                        cnt[a] = 0.0;
                        inside[a] = false;
                        if (a < A) {
                            const double pr = wv[a] * rW;
                            inside[a] = pr > kProbEps && pr < 1.0 - kProbEps;
                            cnt[a] = (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                            nll -= cnt[a] * flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                            if (inside[a]) n_in += cnt[a];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < kAMax; ++a)
                        if (a < A) {
                            gpi[a] += ((inside[a] ? -cnt[a] * frcp(wv[a]) : 0.0) + n_in * rW) * gr[a];
                            if (a >= 1) gmu_s[a - 1] += ((inside[a] ? -cnt[a] : 0.0) + n_in * wv[a] * rW) * tc;
                        }
                }
            }
            nrg += 1.0;
            loss += nll;
            // - d log p / d pi needs the model's floored concentration c_p: recompute from alpha
            {
                double alpha[kAMax], S = 0.0;
#pragma unroll
                for (int a = 0; a < kAMax; ++a) {
                    const bool am = a < A && c.amask[(long)g * A + a] != 0;
                    alpha[a] = a < A ? (am ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
                    S += alpha[a];
                }
                const double rSe = frcp(S + kEps) * c.pi_a0[g];
#pragma unroll
                for (int a = 0; a < kAMax; ++a)
                    if (a < A) {
                        const double v = (alpha[a] + kEps / A) * rSe;
                        const double cp = v < kEps ? kEps : v;
                        gpi[a] -= (cp - 1.0) * frcp(pi[a]);
                    }
            }
            proj = 0.0;
#pragma unroll
            for (int a = 0; a < kAMax; ++a)
                if (a < A) proj += pi[a] * gpi[a];
#pragma unroll
            for (int a = 0; a < kAMax; ++a)
                if (a < A) path[a] += dirichlet_grad_one(pi[a], cq[a], total) * (gpi[a] - proj);
        }
    }

    // ---- reduce over the block's waves in a fixed order (one LDS image)
    double* red = lds;  // [kTNumPart][64]
    for (int ww = 0; ww < nw; ++ww) {
        if (w == ww) {
            auto put = [&](int q, double v) {
                if (ww == 0) red[q * 64 + lane] = v;
                else red[q * 64 + lane] += v;
            };
            put(kTGnoise, gnoise);
            put(kTNrg, nrg);
#pragma unroll
            for (int a = 0; a < kAMax; ++a) {
                put(kTPath + a, path[a]);
                put(kTL + a, L[a]);
                if (a < kAMax - 1) {
                    put(kTGmu + a, gmu_s[a]);
                    put(kTGsig + a, gsig_s[a]);
                }
            }
        }
        __syncthreads();
    }
    if (valid) {
        // rows are spread over the waves; per-allele rows use the (A-1, G) slot layout
        for (int q = w; q < kTNumPart; q += nw) {
            const double v = red[q * 64 + lane];
            if (q >= kTGmu && q < kTGmu + (kAMax - 1)) {
                if (q - kTGmu < A1) c.part[(long)kTGmu * G + (long)(q - kTGmu) * G + g] = v;
            } else if (q >= kTGsig) {
                if (q - kTGsig < A1) c.part[(long)kTGsig * G + (long)(q - kTGsig) * G + g] = v;
            } else {
                c.part[(long)q * G + g] = v;
            }
        }
    }
    double* scratch = lds + kTNumPart * 64;
    const double tot = block_sum(loss, scratch);
    if (threadIdx.x == 0) {
        loss_add(c, ctr.slot, tot);
        if (blockIdx.x == 0) publish_ctr(c, ctr);
    }
}

#endif  // BEAN_AMAX <= 8

// ------------------------------------------------------- k_guide_tiling_wave
// Wave form of k_guide_tiling (same arithmetic): one single-wave workgroup per (64-guide tile,
// replicate), no per-thread arrays indexed at run time and no accumulators across replicates, so
// nothing lives in scratch.  Per-allele state is two register arrays (pi, d loss / d pi) walked by
// fully unrolled loops; everything indexed by the bin b is a thread-private LDS column walked by
// rolled loops: e[b], d nll / d e[b], the digamma differences of the current likelihood, the counts.
// The per-replicate rows go to trow[(q, r, g)]; k_sum_trow adds the replicates into `part`.
// dynamic LDS: (3 B [+ 3 kAMax if ACC]) * 64 doubles + 2 B * 64 floats.
template <bool ACC, bool SURV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4)))
void k_guide_tiling_wave(DevArgs c) {
    extern __shared__ double tls[];
    const int lane = threadIdx.x;
    // 1-D grid, XCD-aware decode (as k_guide_wave2): the R waves of a tile have equal blockIdx % 8, i.e.
    // share an L2, so the tile's allele tables (3 B (A - 1) doubles per guide: 42 MB at config 3) and
    // per-guide values come from HBM once, not once per replicate (measured 432 -> see profiles/)
    const int wg = blockIdx.x;
    const int kk = wg >> 3;
    const int r = kk % c.R;
    const int tile = (kk / c.R) * 8 + (wg & 7);
    if (tile * 64 >= c.G) return;
    const int g = tile * 64 + lane;
    const bool valid = g < c.G;
    const StepCtr ctr = *c.ctrB;
    const int G = c.G, A = c.A, A1 = c.A - 1, B = c.B;
    double* es = tls + lane;                 // e[b]              at es[b * 64]
    double* gs = tls + B * 64 + lane;        // d nll / d e[b]    at gs[b * 64]
    double* ds = tls + 2 * B * 64 + lane;    // digamma diffs     at ds[b * 64]
    double* ps = tls + 3 * B * 64 + lane;    // ACC: pe, d pe / d pi, d pe / d l at ps[(k * kAMax + a) * 64]
    float* xs = (float*)(tls + (3 * B + (ACC ? 3 * kAMax : 0)) * 64) + lane;  // xs[(lik * B + b) * 64]
    double loss = 0.0;

    if (valid) {
        const long RG = (long)c.R * G;
        double* row = c.trow + (long)r * G + g;  // row q of this replicate at row[q * RG]
        const bool rgm = c.rg[(long)r * G + g] != 0;
        const bool use_bc = (c.flags & kUseBc) != 0;
        if (!rgm) {
            // both pi sites, the Multinomial and the count likelihoods are masked by repguide_mask in
            // tiling (model.py:659,682,731; guide 941): the replicate contributes nothing
            for (int q = 0; q < kTNumPart; ++q) row[q * RG] = 0.0;
            if (c.flags & kDumpPi)
                for (int a = 0; a < A; ++a) c.pi_out[((long)r * G + g) * A + a] = 1.0 / A;
        } else {
            // counts of both likelihoods: one batch of loads, then LDS
            {
                float xv[2][kBMax];
#pragma unroll
                for (int b = 0; b < kBMax; ++b) {
                    const long xo = ((long)r * B + (b < B ? b : B - 1)) * G + g;
                    xv[0][b] = c.X[xo];
                    xv[1][b] = use_bc ? c.Xbc[xo] : 0.f;
                }
#pragma unroll
                for (int b = 0; b < kBMax; ++b) {
                    const int bb = b < B ? b : B - 1;
                    xs[(0 * B + bb) * 64] = xv[0][b];
                    xs[(1 * B + bb) * 64] = xv[1][b];
                }
                for (int b = kBMax; b < B; ++b) {  // more conditions than the register batch holds (B <= kBCap)
                    const long xo = ((long)r * B + b) * G + g;
                    xs[(0 * B + b) * 64] = c.X[xo];
                    xs[(1 * B + b) * 64] = use_bc ? c.Xbc[xo] : 0.f;
                }
            }
            // ---- concentrations of the guide's Dirichlet
            double alpha[kAMax], Ssum = 0.0;
#pragma unroll
            for (int a = 0; a < kAMax; ++a) {
                const bool am = a < A && c.amask[(long)g * A + a] != 0;
                alpha[a] = a < A ? (am ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
                Ssum += alpha[a];
            }
            const double pa0 = c.pi_a0[g];
            const double rsq = frcp(Ssum) * pa0;
            double cq[kAMax], total = 0.0;
#pragma unroll
            for (int a = 0; a < kAMax; ++a) {
                cq[a] = alpha[a] * rsq;
                if (SURV && a < A && cq[a] < 1e-5) cq[a] = 1e-5;  // guide-side clamp (survival_model.py:813-821)
                total += cq[a];
            }
            // ---- draw
            double pi[kAMax];
            if (c.pi_in) {
#pragma unroll
                for (int a = 0; a < kAMax; ++a) pi[a] = a < A ? c.pi_in[((long)r * G + g) * A + a] : 0.0;
            } else {
                Rng rng(c.seed, kSitePi, (unsigned long long)r * c.G_tot + guide_stream_id(c, g), ctr.step * 256ull);
                double sum = 0.0;
#pragma unroll
                for (int a = 0; a < kAMax; a += 2) {
                    pi[a] = 0.0;
                    pi[a + 1] = 0.0;
                    if (a < A) {  // components are drawn two at a time (one rejection loop per pair)
                        const GammaPair gp = sample_gamma_pair(cq[a], a + 1 < A ? cq[a + 1] : 1.0, rng);
                        rng.k = gp.k;
                        pi[a] = fmax(gp.g0, kDblMin);
                        sum += pi[a];
                        if (a + 1 < A) {
                            pi[a + 1] = fmax(gp.g1, kDblMin);
                            sum += pi[a + 1];
                        }
                    }
                }
                const double rs = frcp(sum);
#pragma unroll
                for (int a = 0; a < kAMax; ++a)
                    if (a < A) pi[a] = fmin(fmax(pi[a] * rs, kDblMin), kOneMinus);
            }
            if (c.flags & kDumpPi) {
#pragma unroll
                for (int a = 0; a < kAMax; ++a)
                    if (a < A) c.pi_out[((long)r * G + g) * A + a] = pi[a];
            }
            // ---- accessibility transform (utils.py:106-178); its per-allele pieces live in LDS
            double pe0 = pi[0];
            if (ACC) {
                const double kacc = c.kacc[g];
                const double lpn = c.lpn[g];
                double sum = 0.0;
#pragma unroll
                for (int a = 1; a < kAMax; ++a) {
                    if (a < A) {
                        const double s1 = pi[a] * kacc;
                        const bool in1 = s1 > 1e-3 && s1 < 1.0 - 1e-3;
                        const double p1c = fmin(fmax(s1, 1e-3), 1.0 - 1e-3);
                        const double l = flog(p1c * frcp(1.0 - p1c)) + lpn;
                        const double el = exp(l);
                        const double pn = el * frcp(1.0 + el);
                        const bool in2 = pn > 1e-3 && pn < 1.0 - 1e-3;
                        const double pea = fmin(fmax(pn, 1e-3), 1.0 - 1e-3);
                        const double dl = in2 ? pn * (1.0 - pn) : 0.0;
                        ps[(0 * kAMax + a) * 64] = pea;
                        ps[(1 * kAMax + a) * 64] = in1 ? dl * frcp(p1c * (1.0 - p1c)) * kacc : 0.0;
                        ps[(2 * kAMax + a) * 64] = dl;
                        sum += pea;
                    }
                }
                pe0 = 1.0 - sum;
            }
            const double u = SURV ? c.u_g[g] : 0.0;
            // ---- e[b] = sum_a pe_a P_a[b]
#pragma unroll 1
            for (int b = 0; b < B; ++b) {
                double v = pe0 * (SURV ? exp(u * uniform_ld(c.time, b)) : uniform_ld(c.P0, b));
#pragma unroll
                for (int a = 1; a < kAMax; ++a)
                    if (a < A)
                        v += (ACC ? ps[(0 * kAMax + a) * 64] : pi[a]) * c.tabP[((long)b * A1 + (a - 1)) * G + g];
                es[b * 64] = v;
                gs[b * 64] = 0.0;
            }
            // ---- both Dirichlet-Multinomial terms, d nll / d e[b] accumulated in gs
            double nll = 0.0;
            const double* sm = c.smask + r * B;
            const double epsB = kEps / (double)B;
#pragma unroll 1
            for (int lik = 0; lik < 2; ++lik) {
                if (lik == 1 && !use_bc) break;
                const float* xp = xs + lik * B * 64;
                const double* sf = (lik ? c.sf_bc : c.sf) + r * B;
                double nn = 0.0, S = 0.0;
#pragma unroll 1
                for (int b = 0; b < B; ++b) {
                    nn += (double)xp[b * 64];
                    S += es[b * 64] * uniform_ld(sf, b);
                }
                if (!(nn > (double)c.mask_thres)) continue;
                const double a0 = lik ? c.a0_bc[g] : c.a0[g];
                const double inv = frcp(S + kEps);
                double A0 = 0.0, lsum = 0.0, Ua = 0.0, Va = 0.0;
                bool anyfl = false;
#pragma unroll 1
                for (int b = 0; b < B; ++b) {
                    const double araw = (es[b * 64] * uniform_ld(sf, b) + epsB) * inv * a0 * uniform_ld(sm, b);
                    const bool floored = araw < kEps;
                    anyfl = anyfl || floored;
                    const double al = floored ? kEps : araw;
                    A0 += al;
                    const DD db = lgamma_digamma_diff_inl(al, (double)xp[b * 64]);
                    lsum += db.d;
                    ds[b * 64] = db.dp;
                    Ua += floored ? 0.0 : araw;
                    Va += floored ? 0.0 : db.dp * araw;
                }
                // total term: data unless a bin sits on its floor (DevArgs::tot_const)
                DD d0;
                d0.d = 0.0;
                d0.dp = 0.0;
                if (!c.tot_const) {
                    d0 = lgamma_digamma_diff(A0, nn);
                } else if (__any(anyfl)) {
                    const DD dt = lgamma_digamma_diff(A0, nn), dc = lgamma_digamma_diff(a0, nn);
                    if (anyfl) {
                        d0.d = dt.d - dc.d;
                        d0.dp = dt.dp;
                    }
                }
                nll += d0.d - lsum;
                const double W = (d0.dp * Ua - Va) * inv;
#pragma unroll 1
                for (int b = 0; b < B; ++b) {
                    const double sfb = uniform_ld(sf, b), smb = uniform_ld(sm, b);
                    const double araw = (es[b * 64] * sfb + epsB) * inv * a0 * smb;
                    const double ga = araw < kEps ? 0.0 : d0.dp - ds[b * 64];
                    gs[b * 64] += (ga * a0 * smb * inv - W) * sfb;
                }
            }
            // ---- back through the mixture: d loss / d pi_a and the per-allele-slot rows
            double s0 = 0.0;
#pragma unroll 1
            for (int b = 0; b < B; ++b)
                s0 += gs[b * 64] * (SURV ? exp(u * uniform_ld(c.time, b)) : uniform_ld(c.P0, b));
            double gpi[kAMax], gm[kAMax - 1], gnoise = 0.0;
            gpi[0] = ACC ? 0.0 : s0;
#pragma unroll
            for (int a = 1; a < kAMax; ++a) {
                gpi[a] = 0.0;
                gm[a - 1] = 0.0;
                if (a < A) {
                    double sa = 0.0, dm = 0.0, dsg = 0.0;
#pragma unroll 1
                    for (int b = 0; b < B; ++b) {
                        const long o = ((long)b * A1 + (a - 1)) * G + g;
                        const double ge = gs[b * 64];
                        sa += ge * c.tabP[o];
                        dm += ge * c.tabPmu[o];
                        if (!SURV) dsg += ge * c.tabPy[o];
                    }
                    const double pea = ACC ? ps[(0 * kAMax + a) * 64] : pi[a];
                    gm[a - 1] = pea * dm;
                    row[(long)(kTGsig + a - 1) * RG] = pea * dsg;
                    if (ACC) {
                        gpi[a] = (sa - s0) * ps[(1 * kAMax + a) * 64];
                        gnoise += (sa - s0) * ps[(2 * kAMax + a) * 64];
                    } else {
                        gpi[a] = sa;
                    }
                }
            }
            // ---- Multinomial on control allele counts + Dirichlet log-prob pieces
            if (!SURV) {
                double s = 0.0;
#pragma unroll
                for (int a = 0; a < kAMax; ++a)
                    if (a < A) s += pi[a];
                const double ls = s == 1.0 ? 0.0 : flog(s);
                const double rsum = s == 1.0 ? 1.0 : frcp(s);
#pragma unroll
                for (int a = 0; a < kAMax; ++a) {
                    if (a < A) {
                        const double pr = pi[a] * rsum;
                        const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
                        const double lg = inside ? flog(pi[a]) - ls : flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                        double cnt = 0.0;
                        for (int cc = 0; cc < c.C; ++cc)
                            cnt += (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                        nll -= cnt * lg;
                        if (inside) gpi[a] -= cnt * frcp(pi[a]);
                    }
                }
            } else {
                // control_allele_count ~ Multinomial(pi * exp(mu * t_ctrl)), mu = [u, u + mu_a]
                // (survival_model.py:535-548): gradients to pi and, through the growth, to mu_a
                for (int cc = 0; cc < c.C; ++cc) {
                    const double tc = c.ctrl_time[cc];
                    double W = 0.0;
#pragma unroll
                    for (int a = 0; a < kAMax; ++a)
                        if (a < A) W += pi[a] * exp((a == 0 ? u : u + c.mu_a[(long)(a - 1) * G + g]) * tc);
                    const double rW = frcp(W);
                    double n_in = 0.0;
#pragma unroll
                    for (int a = 0; a < kAMax; ++a)
                        if (a < A) {
                            const double wv = pi[a] * exp((a == 0 ? u : u + c.mu_a[(long)(a - 1) * G + g]) * tc);
                            const double pr = wv * rW;
                            const double cnt = (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                            nll -= cnt * flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                            if (pr > kProbEps && pr < 1.0 - kProbEps) n_in += cnt;
                        }
#pragma unroll
                    for (int a = 0; a < kAMax; ++a)
                        if (a < A) {
                            const double gr = exp((a == 0 ? u : u + c.mu_a[(long)(a - 1) * G + g]) * tc);
                            const double wv = pi[a] * gr, pr = wv * rW;
                            const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
                            const double cnt = (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                            gpi[a] += ((inside ? -cnt * frcp(wv) : 0.0) + n_in * rW) * gr;
                            if (a >= 1) gm[a - 1] += ((inside ? -cnt : 0.0) + n_in * wv * rW) * tc;
                        }
                }
            }
            const double dgS_t = c.dgq_t[(long)kAMax * G + g];  // digamma(sum c_q), tabulated by k_param
            // model-side floored concentration c_p (model.py:640-651) for - d log p / d pi
            const double rSe = frcp(Ssum + kEps) * pa0;
            double proj = 0.0;
#pragma unroll
            for (int a = 0; a < kAMax; ++a) {
                if (a < A) {
                    const double rpi = frcp(pi[a]);
                    row[(long)(kTL + a) * RG] = flog(pi[a]);
                    gpi[a] += (cq[a] - 1.0) * rpi;  // + d log q / d pi
                    const double v = (alpha[a] + kEps / A) * rSe;
                    gpi[a] -= ((v < kEps ? kEps : v) - 1.0) * rpi;
                    proj += pi[a] * gpi[a];
                    if (a >= 1) row[(long)(kTGmu + a - 1) * RG] = gm[a - 1];
                }
            }
#pragma unroll
            for (int a = 0; a < kAMax; ++a)
                if (a < A)
                    row[(long)(kTPath + a) * RG] =
                        dirichlet_grad_one_pre(pi[a], cq[a], total, c.dgq_t[(long)a * G + g], dgS_t) * (gpi[a] - proj);
            row[(long)kTGnoise * RG] = gnoise;
            row[(long)kTNrg * RG] = 1.0;
            loss = nll;
        }
    }
    const double tot = wave_sum(loss);
    if (lane == 0) {
        loss_add(c, ctr.slot, tot);
        if (wg == 0) publish_ctr(c, ctr);
    }
}

// part[q, g] = sum_r trow[q, r, g] for the rows of the alleles that exist (fixed order)
__global__ __launch_bounds__(256) void k_sum_trow(DevArgs c) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y;
    if (g >= c.G) return;
    const int A = c.A;
    bool used = q == kTGnoise || q == kTNrg;
    if (q >= kTPath && q < kTPath + kAMax) used = q - kTPath < A;
    if (q >= kTL && q < kTL + kAMax) used = q - kTL < A;
    if (q >= kTGmu && q < kTGmu + kAMax - 1) used = q - kTGmu < A - 1;
    if (q >= kTGsig) used = q - kTGsig < A - 1;
    if (!used) return;
    double s = 0.0;
    for (int r = 0; r < c.R; ++r) s += c.trow[((long)q * c.R + r) * c.G + g];
    c.part[(long)q * c.G + g] = s;
}

// ------------------------------------------------------------------ one-offs
// Data-only constants: P0[b] and the log-factorial terms of the three observed
// sites (they are part of the reported loss and carry no gradient).
__global__ __launch_bounds__(256) void k_prepare(DevArgs c) {
    __shared__ double scratch[16];
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n_rg = (long)c.R * c.G;
    double v = 0.0;
    if (idx < n_rg) {
        const int r = (int)(idx / c.G), g = (int)(idx % c.G);
        const bool rgm = c.rg[idx] != 0;
        double n = 0.0, lf = 0.0, nb = 0.0, lfb = 0.0;
        for (int b = 0; b < c.B; ++b) {
            const double x = (double)c.X[((long)r * c.B + b) * c.G + g];
            n += x;
            lf += lgamma(1.0 + x);
            if (c.flags & kUseBc) {
                const double y = (double)c.Xbc[((long)r * c.B + b) * c.G + g];
                nb += y;
                lfb += lgamma(1.0 + y);
            }
        }
        if (rgm && n > (double)c.mask_thres) v -= lgamma(1.0 + n) - lf;
        if ((c.flags & kUseBc) && rgm && nb > (double)c.mask_thres) v -= lgamma(1.0 + nb) - lfb;
        if (c.tot_const) {
            if (rgm && n > (double)c.mask_thres) v += lgamma_digamma_diff(c.a0[g], n).d;
            if ((c.flags & kUseBc) && rgm && nb > (double)c.mask_thres) v += lgamma_digamma_diff(c.a0_bc[g], nb).d;
        }
        if (c.wrow && r == 0) {
            // wave forms: per-guide count of unmasked replicates (the kPNrg row is data)
            double cnt = 0.0;
            for (int rr = 0; rr < c.R; ++rr) cnt += c.rg[(long)rr * c.G + g] != 0 ? 1.0 : 0.0;
            c.part[(long)kPNrg * c.G + g] = cnt;
        }
        if (c.nobs) {
            c.nobs[idx] = (rgm && n > (double)c.mask_thres) ? n : -1.0;
            c.nobs[n_rg + idx] = ((c.flags & kUseBc) && rgm && nb > (double)c.mask_thres) ? nb : -1.0;
        }
        if ((c.family == kMixture || c.family == kMultiMixture) && rgm) {
            for (int cc = 0; cc < c.C; ++cc) {
                double tot = 0.0, l = 0.0;
                for (int a = 0; a < c.A; ++a) {
                    const double y = (double)c.allele[(((long)r * c.C + cc) * c.G + g) * c.A + a];
                    tot += y;
                    l += lgamma(1.0 + y);
                }
                v -= lgamma(1.0 + tot) - l;
            }
        }
    }
    const double tot = block_sum(v, scratch);
    if (threadIdx.x == 0) fixed_add(c.const_acc, tot);
    if (blockIdx.x == 0 && (int)threadIdx.x < c.B && !c.survival) {
        const double zh = c.z_hi[threadIdx.x], zl = c.z_lo[threadIdx.x];
        const double ch = isinf(zh) ? 1.0 : norm_cdf(zh);
        const double cl = isinf(zl) ? 0.0 : norm_cdf(zl);
        c.P0[threadIdx.x] = ch - cl;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && c.ue_z && !c.survival) {
        int n = 0;
        for (int k = 0; k < 2 * c.B; ++k) {
            const double z = k < c.B ? c.z_hi[k] : c.z_lo[k - c.B];
            int idx = -1;
            if (!isinf(z)) {
                for (int q = 0; q < n; ++q)
                    if (c.ue_z[q] == z) idx = q;
                if (idx < 0) {
                    idx = n;
                    c.ue_z[n++] = z;
                }
            }
            c.ue_idx[k] = idx;
        }
        c.ue_idx[2 * c.B] = n;
    }
}

// most targets spanned by one 64-guide tile (guides are target-sorted; tile k = local guides
// [64 k - sh, 64 k - sh + 64), sh = g_off % 64); *out must start at 0
// The accessibility factor of `scale_pi_by_accessibility` (utils.py:106-131), exp(b) * acc^a with exp(b) taken in
// float32 as the reference does: it depends on the data only, and a float64 pow is some 200 instructions - every
// +Acc kernel used to evaluate it per (replicate, guide) and step (the allele-parallel tiling kernel per allele).
__global__ __launch_bounds__(256) void k_acc_scale(const double* acc, int G, double* out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) out[g] = (double)expf(kAccBf) * pow(acc[g], kAccA);
}

__global__ __launch_bounds__(256) void k_tile_targets(const int* g2t, int G, int sh, int* out) {
    const int tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile * 64 - sh >= G) return;
    const int first = tile * 64 - sh > 0 ? tile * 64 - sh : 0;
    const int last = tile * 64 - sh + 63 < G ? tile * 64 - sh + 63 : G - 1;
    atomicMax(out, g2t[last] - g2t[first] + 1);
}

// longest target (guides), from the offsets themselves; *out must start at 0
__global__ __launch_bounds__(256) void k_max_target_len(const int* toff, int T, int* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) atomicMax(out, toff[t + 1] - toff[t]);
}

// DevArgs::tdesc: where k_guide_wave2 leaves the partial sums of each target (bean_guide_v2.hpp)
__global__ __launch_bounds__(256) void k_tdesc(DevArgs c, int2* out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= c.T) return;
    const int g0 = c.toff[t], g1 = c.toff[t + 1];
    int2 d;
    d.x = 0;
    d.y = 0;
    if (g1 > g0) {
        const int tile_a = (c.g_sh + g0) >> 6, tile_b = (c.g_sh + g1 - 1) >> 6;
        const int first = tile_a * 64 - c.g_sh > 0 ? tile_a * 64 - c.g_sh : 0;
        d.x = tile_a * c.tile_targets + (t - c.g2t[first]);
        d.y = tile_b - tile_a + 1;
    }
    out[t] = d;
}

// loss_hist[i] = accumulated parts of step i + the data-only constant, for n slots from `first`
// (cur != 0: the one slot of the step that has just finished, read from the device step counter)
// One wave per slot: lane l reads accumulator line l (kLossSub = 64 lines), integer wave sums.
__global__ __launch_bounds__(256) void k_loss_finalize(DevArgs c, unsigned long long first, unsigned long long n,
                                                       int cur) {
    const int lane = threadIdx.x & 63;
    unsigned long long i = first + (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (cur) {
        // the n slots that end with the step which has just finished (device step counter): what a captured
        // graph of n steps closes with, whatever step it is replayed at
        const unsigned long long k = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (k >= n) return;
        i = c.ctrA->slot - k;
    } else if (i >= first + n) {
        return;
    }
    static_assert(kLossSub == 64, "one accumulator line per lane");
    long long* acc = c.loss_acc + ((long)i * kLossSub + lane) * kLossWords;
    const long long a0 = acc[0], a1 = acc[1], a2 = acc[2];
    // a finalized slot is left at zero: bean_hip_svi_resume starts a window without k_set_step
    acc[0] = 0;
    acc[1] = 0;
    acc[2] = 0;
    const long long hi = wave_sum_i64(a0), lo = wave_sum_i64(a1);
    long long bad = a2 != 0 ? 1 : 0;
    bad = wave_sum_i64(bad);
    if (lane == 0) {
        const double v = bad != 0 ? __builtin_nan("")
                                  : (double)hi * (1.0 / 1024.0) + (double)lo * (1.0 / 1099511627776.0);
        c.loss_hist[i] = v + fixed_value(c.const_acc, 1);
    }
}

// ---- sample covariates of the sorting NormalModel
// cov_sum[r] = sum_g d nll / d mu of replicate r's rows (fixed order: strided partials, block tree)
__global__ __launch_bounds__(1024) void k_cov_sum(DevArgs c) {
    __shared__ double scratch[16];
    const int r = blockIdx.x;
    double v = 0.0;
    if (c.tsum) {
        // the guide kernel's per-target-part sums of this replicate (slots no target uses hold zero)
        const long S = c.tsum_direct ? 2 * (long)c.T : (long)c.n_tiles * c.tile_targets;
        for (long i = threadIdx.x; i < S; i += blockDim.x) v += c.tsum[(long)r * S + i];
    } else {
        for (int g = threadIdx.x; g < c.G; g += blockDim.x) v += c.wrow[((long)kPGmu * c.R + r) * c.G + g];
    }
    const double tot = block_sum(v, scratch);
    if (threadIdx.x == 0) c.cov_sum[r] = tot;
}

// One block: prior / entropy terms, gradients and ClippedAdam of mu_cov_loc / mu_cov_scale (FINISH),
// the draw of the next step and the replicates' shifts (PREP).  Runs before the k_param launch with
// the same template arguments; reads the step counter the same way.
template <bool FINISH, bool ADAM, bool PREP>
__global__ __launch_bounds__(64) void k_cov_step(DevArgs c) {
    const StepCtr ctr = *c.ctrA;
    const unsigned long long s_prep = FINISH ? ctr.step + 1 : ctr.step;
    AdamCoef ak;
    ak.step_size = ctr.step_size;
    ak.clip = (float)c.clip;
    double loss = 0.0;
    for (int i = threadIdx.x; i < c.n_cov; i += blockDim.x) {
        float loc = c.p[5][i], su = c.p[6][i];
        if (FINISH) {
            const double m = c.cov_mu[i], eps = c.cov_eps[i], sc = exp((double)su);
            double G = 0.0;
            if (i == 0)  // `(data.rep_by_cov * mu_cov)[:, 0]`: only the first covariate shifts the means
                for (int r = 0; r < c.R; ++r) G += c.rbc[(long)r * c.n_cov] * c.cov_sum[r];
            // - log p (Normal(0, 1)) + log q (Normal(loc, scale))
            loss += 0.5 * m * m + kHalfLog2PiC + (-0.5 * eps * eps - (double)su - kHalfLog2PiC);
            const double Gm = G + m;
            emit_grad<ADAM>(c, 5, i, Gm, ak);
            emit_grad<ADAM>(c, 6, i, Gm * eps * sc - 1.0, ak);
            if (ADAM) {
                loc = c.p[5][i];
                su = c.p[6][i];
            }
        }
        if (PREP) {
            double eps;
            if (c.eps_noise_in) {
                eps = c.eps_noise_in[i];
            } else {
                eps = (double)normal2_at(c.seed, ((unsigned long long)kSiteCov << 48) + (unsigned long long)i, s_prep * 4ull).x;
            }
            const double m = (double)loc + eps * exp((double)su);
            c.cov_eps[i] = eps;
            c.cov_mu[i] = m;
            if (c.eps_noise_out) c.eps_noise_out[i] = eps;
            if (i == 0)
                for (int r = 0; r < c.R; ++r) c.cov_shift[r] = c.rbc[(long)r * c.n_cov] * m;
        }
    }
    if (FINISH) {
        // guide-sharded: mu_cov is replicated on every rank, one of them counts its prior / entropy terms
        const double tot = wave_sum(c.not_loss_owner ? 0.0 : loss);
        if (threadIdx.x == 0) loss_add(c, ctr.slot, tot);
    }
}

// grid = 1 + blocks over the accumulator words: also clears loss_hist and the loss accumulators of the
// n slots from `slot` (two memsets less per call)
__global__ __launch_bounds__(256) void k_set_step(DevArgs c, unsigned long long step, unsigned long long slot,
                                                  unsigned long long n) {
    const unsigned long long words = n * kLossSub * kLossWords;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < words;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        c.loss_acc[slot * kLossSub * kLossWords + i] = 0;
        if (i < n) c.loss_hist[slot + i] = 0.0;
    }
    // the arrival counters of the fused step kernel / the q0 totals return to zero by themselves when a launch
    // completes; an aborted capture or a failed launch would leave them non-zero for every later call
    if (c.tile_ctr)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.n_arrival_ctr; i += gridDim.x * blockDim.x) c.tile_ctr[i] = 0;
    if (c.q0_ctr && blockIdx.x == 0 && threadIdx.x == 1) *c.q0_ctr = 0;
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    StepCtr s;
    s.step = step;
    s.slot = slot;
    // ClippedAdam step size of the first update (t = step + 1): the fused step kernel's first launch
    // takes it from here, the two-launch path overwrites it (publish_ctr) with the same value
    s.step_size = adam_coef(c, step + 1).step_size;
    s.pad_ = 0.f;
    *c.ctrA = s;
    *c.ctrB = s;
}

// Stand-alone ClippedAdam over one parameter array (bean_hip_adam).
__global__ __launch_bounds__(256) void k_adam(float* p, const float* g, float* m, float* v, long n,
                                              DevArgs c, unsigned long long t) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AdamCoef k = adam_coef(c, t);
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, mm, vv, g[i], k);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
}

// Unit-test hook for the special functions (bean_hip_test_special).
__global__ __launch_bounds__(256) void k_test_special(int op, long n, const double* a, const double* x,
                                                      const double* b, double* o0, double* o1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (op == 0) {
        const DD d = lgamma_digamma_diff(a[i], x[i]);
        o0[i] = d.d;
        o1[i] = d.dp;
    } else if (op == 1) {
        double lg, dg;
        lgamma_digamma(a[i], lg, dg);
        o0[i] = lg;
        o1[i] = dg;
    } else if (op == 2) {
        o0[i] = dirichlet_grad_one(x[i], a[i], b[i]);
    } else if (op == 3) {
        o0[i] = norm_cdf(a[i]);
    } else if (op == 4) {
        unsigned long long seed;
        memcpy(&seed, &x[0], 8);
        Rng rng(seed, kSiteAux, (unsigned long long)i, 0ull);
        const GammaPair gp = sample_gamma_pair(a[i], b[i], rng);
        double g0 = fmax(gp.g0, kDblMin);
        double g1 = fmax(gp.g1, kDblMin);
        const double s = g0 + g1;
        o0[i] = fmin(fmax(g0 / s, kDblMin), kOneMinus);
        o1[i] = fmin(fmax(g1 / s, kDblMin), kOneMinus);
    } else if (op == 5) {
        // the two-chain form: this element as chain a, element i ^ 1 as chain b (must equal op 0 bit for bit)
        const long j = (i ^ 1) < n ? (i ^ 1) : i;
        const DD2 d = lgamma_digamma_diff2(a[i], x[i], a[j], x[j]);
        o0[i] = d.a.d;
        o1[i] = d.a.dp;
    } else if (op == 6 || op == 7) {
        // the Dirichlet-over-all-guides site's gammas as k_param's q0 blocks form them (float32 floor): from
        // the plain sampler (6) and from the one that leaves the rejection loop out where the boost factor
        // has already put the draw on the floor (7) - the same values
        unsigned long long seed;
        memcpy(&seed, &x[0], 8);
        Rng rng(seed, kSiteQ0, (unsigned long long)i, 0ull);
        const GammaPair gp = op == 6 ? sample_gamma_pair(a[i], b[i], rng) : sample_gamma_pair_floor32(a[i], b[i], rng);
        o0[i] = (double)fmaxf((float)gp.g0, 1.17549435e-38f);
        o1[i] = (double)fmaxf((float)gp.g1, 1.17549435e-38f);
    } else if (op == 8 || op == 9) {
        // the wide tiling kernel's allele gammas (double floor, DBL_MIN): plain sampler (8) and the one that
        // skips the rejection loop below the floor (9) - the same values
        unsigned long long seed;
        memcpy(&seed, &x[0], 8);
        Rng rng(seed, kSitePi, (unsigned long long)i, 0ull);
        const GammaPair gp = op == 8 ? sample_gamma_pair(a[i], b[i], rng) : sample_gamma_pair_floord(a[i], b[i], rng);
        o0[i] = fmax(gp.g0, kDblMin);
        o1[i] = fmax(gp.g1, kDblMin);
    }
}

}  // namespace bean

#include "bean_guide_v2.hpp"
#include "bean_async_v2.hpp"  // all the steps of a call in one launch, tile-asynchronous
#ifdef BEAN_AB_KERNELS  // opt-in steppers, both bit-identical to the default path and measured slower
#include "bean_step_v2.hpp"
#endif
#include "bean_survival_v2.hpp"
#include "bean_tiling_v2.hpp"
#ifdef BEAN_AB_KERNELS
#include "bean_tile_svi.hpp"
#endif

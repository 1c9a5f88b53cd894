// k_guide_wave2: the per-(replicate, guide) kernel of the variant sorting families, second form.
//
// Same arithmetic as k_guide_wave (bean_kernels.hpp); what changed is where state lives and what
// leaves the kernel:
//   * rows: everything k_param needs from a (replicate, guide) is five numbers - d/dmu_t, d/dy_t,
//     d/dnoise and, per Dirichlet component a, GA_a = [unclamped] (log pi_a + path_a) - [rg] log pi_a -
//     because the (c_p - 1) log pi / (c_q - 1) log pi terms of the two Dirichlet log-densities are
//     added to the loss here, where c_p / c_q are known.  v1 wrote eight to nine rows.
//   * (round 4) d/dmu_t and d/dy_t do not leave the wave per guide any more: the guides of a target are
//     consecutive lanes, so the wave adds them up per target (segmented scan, fixed shape) and stores one
//     pair per (replicate, TARGET PART) - DevArgs::tsum, target_part_sums below.  k_param adds R (or 2 R,
//     where a target straddles two tiles) numbers per target instead of R x its guides.  Tiles are aligned
//     to the GLOBAL guide index (a shard's first tile starts at g_off rounded down to 64), so a target is
//     cut at the same guides whatever the shard: parameters stay bitwise shard-independent.
//   * count totals n = sum_b x_b are re-summed from the staged counts (exact: integer-valued
//     float32), so the (2, R, G) `nobs` array is gone.
//   * the bin loop that contains the lgamma / digamma differences carries two accumulators (A0 and the
//     lgamma sum); the digamma differences are parked in a thread-private LDS column and the twelve
//     gradient sums of v1 become a light second loop with eight.  Nothing is spilled to scratch.
//   * wave-uniform per-bin constants (size factors, sample mask, P0) are staged in LDS with the first
//     batch of loads: the bin loops contain no scalar-memory loads (LDS and scalar loads share one
//     counter, so mixing them forces full drains).
//   * 1-D grid with an XCD-aware decode: the R waves of a tile have equal blockIdx % 8, i.e. share an
//     L2, so the tile's per-guide values and table columns are fetched from HBM once, not R times.
//
// Reference semantics as k_guide_wave: bean/model/model.py:378-547 (model), 785-858 (guide),
// bean/model/utils.py:10-31, 106-178.
#pragma once

namespace bean {

// per-replicate rows of the v2 wave form, (kW2Rows, R, G) doubles in DevArgs::wrow
enum W2Row { kW2Gmu = 0, kW2Gy = 1, kW2Gnoise = 2, kW2GA0 = 3, kW2GA1 = 4, kW2Rows = 5 };
constexpr int kW2Misc = 6;  // per-guide values staged in LDS: a0, a0_bc, allele counts (2), c_p (2)

// bytes of dynamic LDS of one wave
__host__ __device__ inline size_t guide_wave2_lds(int B, int ntm) {
    return ((size_t)3 * B * ntm + 4 * B + (size_t)B * 64 + (size_t)kW2Misc * 64) * sizeof(double) +
           (size_t)2 * B * 64 * sizeof(float);
}

// alpha_b before its floor, from the mixture weights: evaluated with explicit roundings so that the
// two bin loops (which both need "is alpha_b on its floor?") agree bit for bit
__device__ __forceinline__ double alpha_raw(double w0, double p0, double w1, double p1, double sfb, double epsB,
                                            double km) {
    const double e = fma(w0, p0, w1 * p1);
    return fma(e, sfb, epsB) * km;
}

// Diagnostic builds only (wrong results): -DBEAN_GW_DIAG=n removes one piece of the kernel so that A/B
// timings give that piece's cost in place: 1 sampler, 2 implicit-gradient calls, 4 second bin loop.
#ifndef BEAN_GW_DIAG
#define BEAN_GW_DIAG 0
#endif

// A row of this (replicate, guide) for the launch that follows (k_param), or - fused step kernel, STEP -
// for the wave of the SAME launch that finishes the tile: then an agent-scope store (global_store ... sc1).
// STEP is a mode: 0 = two launches per step (plain accesses); 1 = k_step_wave2, one launch per step (rows, sums and
// loss parts leave the wave through agent-scope stores); 2 = k_svi_async, one launch for MANY steps
// (bean_async_v2.hpp): as 1, and everything a finishing wave of an earlier step of the SAME launch has written -
// Phi tables, alpha_pi, the tabulated digammas, the noise draw - is read with agent-scope loads (sc1: they bypass
// this CU's L1, which no other CU's store ever refreshes).
template <int STEP>
__device__ __forceinline__ void w2_row_store(double* p, double v) {
    if (STEP) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
#define W2_ROW_STORE(ptr, val) w2_row_store<STEP>((ptr), (val))
// value another wave of this launch may have written at an earlier step (STEP == 2), else a plain load
template <int STEP, typename T>
__device__ __forceinline__ T w2_ld(const T* p) {
    if (STEP == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
// ... through a pointer the compiler knows to be global (a flat load in flight makes every later wait a full one)
template <int STEP>
__device__ __forceinline__ double w2_ld_g(const double __attribute__((address_space(1))) * p) {
    if (STEP == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// The arithmetic of one (replicate, guide): draw, accessibility transform, both Dirichlet-Multinomial terms
// with their gradients, Multinomial on the control allele counts, the pi site's densities and implicit
// gradient; stores the per-guide rows of the pair (d/dnoise, GA_0, GA_1), hands d/dmu_t and d/dy_t back
// in o_mu / o_y and returns the pair's part of the loss.  ONE body for every kernel
// that runs it (k_guide_wave2, k_step_wave2: a wave = 64 guides of one replicate, LS = 64; k_svi_tile:
// a workgroup = all replicates of a tile, LS = its thread count), so that they produce the same bits.
//   tp            this guide's column of the staged Phi tables: tp[(which * B + b) * ntm]
//   c_sf, c_sm, c_p0   the replicate's per-bin constants (size factors of both likelihoods, sample mask, P0)
//   xl            the pair's counts, xl[(lik * B + b) * LS] (float; reused for parked digamma halves)
//   dps           thread-private column of B doubles, dps[b * LS]
//   m_*           per-guide values staged by the caller (a0, a0_bcmatch, control allele counts) and two
//                 thread-private slots for the model-side concentrations
// The pi site's draw of one (replicate, guide) (model.py:439-450 / guide 812-826): needs the guide's alpha_pi and
// pi_a0 only - none of the counts or tables - so k_guide_wave2 runs it while those loads are still in flight.
// Leaves the model-side concentrations in the thread-private slots m_cp0 / m_cp1 for guide_pair_math.
template <int FAM>
__device__ __forceinline__ void guide_pair_draw(const DevArgs& c, const StepCtr& ctr, int r, int g, float api0,
                                                float api1, double pa0, const uint4* philox_first, double* m_cp0,
                                                double* m_cp1, double& pi0, double& pi1) {
    if (FAM != kMixture) return;
    const double al0 = (double)expf(api0), al1 = (double)expf(api1);
    const double rs = frcp(al0 + al1) * pa0;
    const double cp0 = al0 * rs, cp1 = al1 * rs;
    *m_cp0 = cp0;  // needed again after the likelihoods
    *m_cp1 = cp1;
    const double cq0 = cp0 < 1e-5 ? 1e-5 : cp0, cq1 = cp1 < 1e-5 ? 1e-5 : cp1;
#if BEAN_GW_DIAG == 1
    pi0 = 0.3 + 1e-3 * cq0;
    pi1 = 0.7 - 1e-3 * cq1;
#else
    // (draws handed in - DevArgs::pi_in, tests - are read by guide_pair_math: a load on either side of this
    // branch would make the compiler drain the loads in flight before the sampler may reuse its register)
    if (!c.pi_in) {
        Rng rng(c.seed, kSitePi, (unsigned long long)r * c.G_tot + (c.g_off + g), ctr.step * 256ull);
        const GammaPair gp = sample_gamma_pair_inl(cq0, cq1, rng, philox_first);
        const double gm0 = fmax(gp.g0, kDblMin), gm1 = fmax(gp.g1, kDblMin);
        const double rs2 = frcp(gm0 + gm1);
        pi0 = fmin(fmax(gm0 * rs2, kDblMin), kOneMinus);
        pi1 = fmin(fmax(gm1 * rs2, kDblMin), kOneMinus);
    }
#endif
}

// (pi0, pi1: the pair's draw, guide_pair_draw)
template <int FAM, bool ACC, int STEP, int LS>
__device__ __forceinline__ double guide_pair_math(const DevArgs& c, const StepCtr& ctr, int r, int g, bool rgm,
                                                  double pi0, double pi1,
                                                  const double* tp, int ntm, const double* c_sf, const double* c_sm,
                                                  const double* c_p0, float* xl, double* dps, const double* m_a0,
                                                  const double* m_a0bc, const double* m_cnt0, const double* m_cnt1,
                                                  double* m_cp0, double* m_cp1, double& o_mu, double& o_y) {
    constexpr bool MIX = FAM == kMixture;
    const int G = c.G, B = c.B, R = c.R;
    const bool use_bc = (c.flags & kUseBc) != 0;
    const long rgi = (long)r * G + g;
    const long RG = (long)R * G;
    double pe1 = 1.0;
    double dpe1_dpi1 = 0.0, dpe1_dl = 0.0;
    if (!MIX) {
        pi0 = 0.0;
        pi1 = 1.0;
    }
    if (MIX) {
#if BEAN_GW_DIAG != 1
        if (c.pi_in) {
            pi0 = c.pi_in[rgi * 2];
            pi1 = c.pi_in[rgi * 2 + 1];
        }
#endif
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
            const double l = flog(p1c * frcp(1.0 - p1c)) + w2_ld<STEP>(c.lpn + g);
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
    const double epsB = kEps / (double)B;
    double a_mu = 0.0, a_y = 0.0, g0 = 0.0, g1 = 0.0, nll = 0.0;
    // pass 1 of both likelihoods: n = sum x_b (data) and S = sum_b e_b sf_b
    double S_x = 0.0, S_bc = 0.0, n_x = 0.0, n_bc = 0.0;
#pragma unroll 1
    for (int b = 0; b < B; ++b) {
        const double e = fma(w0, MIX ? c_p0[b] : 0.0, w1 * tp[b * ntm]);
        S_x += e * c_sf[b];
        S_bc += e * c_sf[B + b];
        n_x += (double)xl[b * LS];
        n_bc += (double)xl[(B + b) * LS];
    }
    // ---- loop 1 of BOTH likelihoods in one pass over the bins: the lgamma / digamma differences of
    // X[b] and X_bcmatch[b] are independent dependency chains, evaluated side by side
    // (lgamma_digamma_diff2).  The digamma differences are parked in LDS: those of X in dps[b], those
    // of X_bcmatch as two 32-bit halves in the count slots of bin b, which are dead once read.
    // The site is masked by (sum_b x > mask_thres) & repguide_mask (model.py:526-547): lane by lane.
    const bool on_x = rgm && n_x > (double)c.mask_thres;
    const bool on_bc = use_bc && rgm && n_bc > (double)c.mask_thres;
    const double inv_x = frcp(S_x + kEps), inv_bc = frcp(S_bc + kEps);
    const double ai_x = *m_a0 * inv_x, ai_bc = *m_a0bc * inv_bc;
    double A0_x = 0.0, A0_bc = 0.0, lsum_x = 0.0, lsum_bc = 0.0;
    bool fl_x = false, fl_bc = false;
    if (use_bc) {
        unsigned int* xu = reinterpret_cast<unsigned int*>(xl);
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const double p0 = MIX ? c_p0[b] : 0.0, p1 = tp[b * ntm], smb = c_sm[b];
            const double x0 = (double)xl[b * LS], x1 = (double)xl[(B + b) * LS];
            const double ar0 = alpha_raw(w0, p0, w1, p1, c_sf[b], epsB, ai_x * smb);
            const double ar1 = alpha_raw(w0, p0, w1, p1, c_sf[B + b], epsB, ai_bc * smb);
            fl_x = fl_x || ar0 < kEps;
            fl_bc = fl_bc || ar1 < kEps;
            const double al0 = ar0 < kEps ? kEps : ar0, al1 = ar1 < kEps ? kEps : ar1;
            A0_x += al0;
            A0_bc += al1;
            const DD2 dd = lgamma_digamma_diff2(al0, x0, al1, x1);
            lsum_x += dd.a.d;
            lsum_bc += dd.b.d;
            dps[b * LS] = dd.a.dp;
            const unsigned long long bits = __builtin_bit_cast(unsigned long long, dd.b.dp);
            xu[b * LS] = (unsigned int)bits;
            xu[(B + b) * LS] = (unsigned int)(bits >> 32);
        }
    } else {
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const double x0 = (double)xl[b * LS];
            const double ar0 = alpha_raw(w0, MIX ? c_p0[b] : 0.0, w1, tp[b * ntm], c_sf[b], epsB, ai_x * c_sm[b]);
            fl_x = fl_x || ar0 < kEps;
            const double al0 = ar0 < kEps ? kEps : ar0;
            A0_x += al0;
            const DD db = lgamma_digamma_diff(al0, x0);
            lsum_x += db.d;
            dps[b * LS] = db.dp;
        }
    }
#pragma unroll 1
    for (int lik = 0; lik < 2; ++lik) {
        if (lik == 1 && !use_bc) break;
        if (!(lik ? on_bc : on_x)) continue;
        const double* sf = c_sf + lik * B;
        const double nn = lik ? n_bc : n_x;
        const double a0 = lik ? *m_a0bc : *m_a0;
        const double inv = lik ? inv_bc : inv_x;
        const double ai = lik ? ai_bc : ai_x;
        const double A0 = lik ? A0_bc : A0_x, lsum = lik ? lsum_bc : lsum_x;
        const bool floored = lik ? fl_bc : fl_x;
        // total term lgamma(A0 + n) - lgamma(A0): data unless a bin sits on its floor
        // (DevArgs::tot_const); a lane's arithmetic does not depend on its wave's other lanes
        DD d0;
        d0.d = 0.0;
        d0.dp = 0.0;
        if (!c.tot_const) {
            d0 = lgamma_digamma_diff(A0, nn);
        } else if (__any(floored)) {
            const DD dt = lgamma_digamma_diff(A0, nn), dc = lgamma_digamma_diff(a0, nn);
            if (floored) {
                d0.d = dt.d - dc.d;
                d0.dp = dt.dp;
            }
        }
        nll += d0.d - lsum;
        // loop 2: with ga_b = d0.dp - dpsi_b (0 where alpha_b sits on its floor) and
        // k_b = a0 m_b inv sf_b:  S_Q = sum ga_b k_b Q_b,  t_Q = sum sf_b Q_b,
        // Wa = sum ga_b alpha_b;  d nll / d(weight of Q) = S_Q - Wa inv t_Q
        double Wa = 0.0;
        double S_mu = 0.0, S_y = 0.0, S_0 = 0.0, S_1 = 0.0;
        double t_mu = 0.0, t_y = 0.0, t_0 = 0.0, t_1 = 0.0;
#pragma unroll 1
        for (int b = 0; b < (BEAN_GW_DIAG == 4 ? 1 : B); ++b) {
            const double p0 = MIX ? c_p0[b] : 0.0;
            const double p1 = tp[b * ntm], pmu = tp[(B + b) * ntm], py = tp[(2 * B + b) * ntm];
            const double sfb = sf[b];
            const double km = ai * c_sm[b];
            const double araw = alpha_raw(w0, p0, w1, p1, sfb, epsB, km);
            double dpb = dps[b * LS];
            if (lik) {
                const unsigned int* xu = reinterpret_cast<const unsigned int*>(xl);
                dpb = __builtin_bit_cast(double, ((unsigned long long)xu[(B + b) * LS] << 32) |
                                                     (unsigned long long)xu[b * LS]);
            }
            const double ga = araw < kEps ? 0.0 : d0.dp - dpb;
            Wa += ga * araw;
            const double cb = ga * km * sfb;
            S_mu += cb * pmu;
            t_mu += sfb * pmu;
            S_y += cb * py;
            t_y += sfb * py;
            S_1 += cb * p1;
            t_1 += sfb * p1;
            if (MIX) {
                S_0 += cb * p0;
                t_0 += sfb * p0;
            }
        }
        const double W = Wa * inv;
        a_mu += w1 * (S_mu - W * t_mu);
        a_y += w1 * (S_y - W * t_y);
        g0 += S_0 - W * t_0;
        g1 += S_1 - W * t_1;
    }
    double* row = c.wrow + rgi;  // row q of this replicate at row[q * RG]
    o_mu = a_mu;  // d nll / d mu_t, d nll / d y_t of this pair: summed per target by the caller
    o_y = a_y;
    if (MIX) {
        const double cp0 = *m_cp0, cp1 = *m_cp1;
        const bool cl0 = cp0 < 1e-5, cl1 = cp1 < 1e-5;
        const double cq0 = cl0 ? 1e-5 : cp0, cq1 = cl1 ? 1e-5 : cp1;
        // d loss / d pi through the likelihood
        double gpi0 = g0, gpi1 = g1;
        if (ACC) {
            gpi0 = 0.0;
            gpi1 = (g1 - g0) * dpe1_dpi1;
            W2_ROW_STORE(row + kW2Gnoise * RG, (g1 - g0) * dpe1_dl);
        }
        // digamma of the concentrations, tabulated by k_param (DevArgs::dgq): issued here, used
        // by the implicit-gradient calls below
        const double dgS = w2_ld<STEP>(c.dgq + 3 * (long)G + g), dg0 = w2_ld<STEP>(c.dgq + 4 * (long)G + g),
                     dg1 = w2_ld<STEP>(c.dgq + 5 * (long)G + g);
        const double lpi0 = flog(pi0), lpi1 = flog(pi1);
        const double rpi0 = frcp(pi0), rpi1 = frcp(pi1);
        if (rgm) {
            // Multinomial(probs = pi) on control allele counts (model.py:470-474):
            // torch renormalises the probabilities and clamps them to [eps, 1 - eps]
            const double s = pi0 + pi1;
            const double ls = s == 1.0 ? 0.0 : flog(s);
            const double rs = s == 1.0 ? 1.0 : frcp(s);
            const double cnt0 = *m_cnt0, cnt1 = *m_cnt1;
            const double pr0 = pi0 * rs, pr1 = pi1 * rs;
            const bool in0 = pr0 > kProbEps && pr0 < 1.0 - kProbEps;
            const bool in1 = pr1 > kProbEps && pr1 < 1.0 - kProbEps;
            const double lg0 = in0 ? lpi0 - ls : flog(fmin(fmax(pr0, kProbEps), 1.0 - kProbEps));
            const double lg1 = in1 ? lpi1 - ls : flog(fmin(fmax(pr1, kProbEps), 1.0 - kProbEps));
            nll -= cnt0 * lg0;
            nll -= cnt1 * lg1;
            if (in0) gpi0 -= cnt0 * rpi0;
            if (in1) gpi1 -= cnt1 * rpi1;
            // - log p(pi): Dirichlet(c_p) under the repguide mask (model.py:454-463); its
            // normaliser is per guide (k_param)
            gpi0 -= (cp0 - 1.0) * rpi0;
            gpi1 -= (cp1 - 1.0) * rpi1;
            nll -= (cp0 - 1.0) * lpi0 + (cp1 - 1.0) * lpi1;
        }
        // + log q(pi): Dirichlet(c_q), unmasked in the guide (model.py:839-847)
        gpi0 += (cq0 - 1.0) * rpi0;
        gpi1 += (cq1 - 1.0) * rpi1;
        nll += (cq0 - 1.0) * lpi0 + (cq1 - 1.0) * lpi1;
        const double proj = pi0 * gpi0 + pi1 * gpi1;
        const double total = cq0 + cq1;
        double path0 = 0.0, path1 = 0.0;
        // torch's approximation switches formula on x <= 0.5 / x >= 0.5: a lane takes its SMALLER
        // component in the first pass and the larger one in the second, so that a pass runs one side's
        // formulas for the whole wave (in component order every pass ran both sides': 13 % of
        // the kernel).  Same calls, same arguments, same bits.
        const int first = pi0 <= pi1 ? 0 : 1;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            const int a = pass ^ first;
            if (a ? cl1 : cl0) continue;
#if BEAN_GW_DIAG == 2
            const double v = (a ? pi1 : pi0) * ((a ? gpi1 : gpi0) - proj);
#else
            const double v = dirichlet_grad_one_pre(a ? pi1 : pi0, a ? cq1 : cq0, total, a ? dg1 : dg0, dgS) *
                             ((a ? gpi1 : gpi0) - proj);
#endif
            path0 = a ? path0 : v;
            path1 = a ? v : path1;
        }
        // d loss / d c_a of this replicate apart from the per-guide normaliser terms:
        // (c_q unclamped) log pi_a + pathwise term, minus (masked) log pi_a of the model site
        W2_ROW_STORE(row + kW2GA0 * RG, (cl0 ? 0.0 : lpi0 + path0) - (rgm ? lpi0 : 0.0));
        W2_ROW_STORE(row + kW2GA1 * RG, (cl1 ? 0.0 : lpi1 + path1) - (rgm ? lpi1 : 0.0));
    }
    return nll;
}

// Per-target sums of a wave's d/dmu_t, d/dy_t (a8's backward inside the wave).  The guides of a target are
// consecutive lanes (guides are target-sorted): a segmented inclusive scan in Hillis-Steele form - after the
// step with offset d a lane holds the sum of the (up to) 2 d values that end at it - whose addition tree
// depends on the position INSIDE the segment only, not on the lane, so that a target part gives the same
// bits wherever it sits in a wave.  `pos` = lanes between this lane and the head of its segment;
// c.seg_steps = ceil(log2(longest target, capped at a tile)).  The lane at the end of each segment stores
// the pair: tsum[(q R + r) S + tile ntm + tcol], S = n_tiles ntm (STEP: agent-scope stores, as the rows).
// Lanes without a guide (the ends of a shard's first and last tile) carry tcol = -1 and zeros.
// tsum_direct (no target longer than a tile): the slot is 2 t, or 2 t + 1 for the wave's FIRST segment when its
// target began in the previous tile (`cont0`) - a target then has at most these two parts.
template <int STEP>
__device__ __forceinline__ void target_part_sums(const DevArgs& c, int lane, int tile, int r, int tcol, bool valid,
                                                 double a_mu, double a_y, int t0, bool cont0) {
    // head of a segment: lane 0, or another target than the lane below
    const int below = __shfl_up(tcol, 1, 64);
    const bool head = lane == 0 || below != tcol;
    const unsigned long long heads = __ballot(head);
    // position inside the segment: lane - (highest head at or below this lane)
    const unsigned long long le = heads & (~0ull >> (63 - lane));
    const int pos = lane - (63 - __clzll((long long)le));
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull) != 0;
    const int steps = c.seg_steps;
    for (int s = 0; s < steps; ++s) {
        const int d = 1 << s;
        const double ym = __shfl_up(a_mu, d, 64), yy = __shfl_up(a_y, d, 64);
        if (pos >= d) {
            a_mu += ym;
            a_y += yy;
        }
    }
    if (valid && tail) {
        const long S = c.tsum_direct ? 2 * (long)c.T : (long)c.n_tiles * c.tile_targets;
        const long slot = c.tsum_direct ? 2 * (long)(t0 + tcol) + ((tcol == 0 && cont0) ? 1 : 0)
                                        : (long)tile * c.tile_targets + tcol;
        double* o = c.tsum + (long)r * S + slot;
        w2_row_store<STEP>(o, a_mu);
        w2_row_store<STEP>(o + (long)c.R * S, a_y);
    }
}

// The work of one wave = 64 consecutive guides of one replicate (tile k holds the guides whose GLOBAL index
// g_off + g lies in [64 (k + g_off / 64), + 64): DevArgs::g_sh = g_off % 64 lanes of a shard's first tile
// are empty).  Returns false for the padded tiles of the XCD-aware grid; otherwise the wave's part of the
// loss in `tot` (valid in lane 0).
// guide_wave2_tile: the work on (tile, replicate r); `wg` = the wave's id in the 1-D grid of padded tiles x replicates
// (index of its loss part, stamps).
template <int FAM, bool ACC, int STEP>
__device__ __forceinline__ void guide_wave2_tile(const DevArgs& c, const StepCtr& ctr, const int tile, const int r,
                                                 const int wg, int& t0_o, int& nt_o, double& tot_o) {
    constexpr bool MIX = FAM == kMixture;
    extern __shared__ double tabs[];
    const int lane = threadIdx.x;
    const int G = c.G, T = c.T, B = c.B, R = c.R;
    (void)wg;
    const int g = tile * 64 + lane - c.g_sh;
    const bool valid = g >= 0 && g < G;
    double loss = 0.0;
#ifdef BEAN_STAMP
    const long wave_gid = wg;
#endif
    BEAN_STAMP_AT(0);
    BEAN_STAMP_CLK(0);

    const int g_first = tile * 64 - c.g_sh > 0 ? tile * 64 - c.g_sh : 0;
    const int g_last = (tile * 64 + 63 - c.g_sh < G ? tile * 64 + 63 - c.g_sh : G - 1);
    const int t0 = __builtin_amdgcn_readfirstlane(c.g2t[g_first]);
    const int nt = __builtin_amdgcn_readfirstlane(c.g2t[g_last]) - t0 + 1;
    // did the tile's first target begin in the previous tile?  (wave-uniform; used when the sums are stored)
    const int tof_first = c.tsum_direct ? uniform_ld_i(c.toff, t0) : 0;
    const int ntm = c.tile_targets;
    t0_o = t0;
    nt_o = nt;
    // LDS: [3][B][ntm] table columns | [4][B] sf, sf_bc, sample mask, P0 of this replicate |
    //      [B][64] digamma differences | [kW2Misc][64] per-guide values | [2][B][64] counts (float)
    double* cst = tabs + 3 * B * ntm;
    double* dps = cst + 4 * B + lane;                 // dps[b * 64]
    double* ms = cst + 4 * B + B * 64 + lane;         // ms[q * 64]
    float* xs = (float*)(cst + 4 * B + B * 64 + kW2Misc * 64);
    const bool use_bc = (c.flags & kUseBc) != 0;
    int tcol = 0;
    uint4 philox_first = make_uint4(0u, 0u, 0u, 0u);
    bool rgm = false;
    float api0 = 0.f, api1 = 0.f;
    double pa0 = 0.0, pi0 = 0.0, pi1 = 1.0;
    {
        // everything the wave reads from global memory, issued as one batch before the first wait
        const int gc = valid ? g : (g < 0 ? 0 : G - 1);
        const long rgc = (long)r * G + gc;
        // (first in the batch: loads return in issue order, and the draw below waits for these three only)
        if (MIX) {
            api0 = w2_ld<STEP>(c.p[4] + 2 * gc);
            api1 = w2_ld<STEP>(c.p[4] + 2 * gc + 1);
            pa0 = c.pi_a0[gc];
        }
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
        const int per = 64 >> lg2;
        const int j = lane & ((1 << lg2) - 1), sub = lane >> lg2;
        const int n_rows = 3 * B;
        constexpr int kTabLoads = 6;
        const long covoff = c.n_cov ? (long)r * B * T : 0;  // sample covariates: tables per replicate
        double tv[kTabLoads];
        int wbv[kTabLoads];
        // (the three table pointers as register values: left to itself the compiler turns the selection below into
        // a per-lane LOAD of the pointer from the kernel-argument segment, i.e. one more round trip before the
        // table loads can be issued)
        // (and as GLOBAL pointers: through the asm the compiler no longer knows the address space, and a flat load
        // in flight makes every later wait a full one)
        typedef const double __attribute__((address_space(1))) * GlobalTab;
        const double *tabP_ = c.tabP, *tabPmu_ = c.tabPmu, *tabPy_ = c.tabPy;
        asm volatile("" : "+s"(tabP_), "+s"(tabPmu_), "+s"(tabPy_));
        const GlobalTab tabP = (GlobalTab)tabP_, tabPmu = (GlobalTab)tabPmu_, tabPy = (GlobalTab)tabPy_;
#pragma unroll
        for (int q = 0; q < kTabLoads; ++q) {
            const int wb = q * per + sub;
            const bool ok = wb < n_rows && j < nt;
            const int wbc = ok ? wb : 0;
            const int which = (wbc >= B) + (wbc >= 2 * B), bb = wbc - which * B;
            const GlobalTab tab = which == 0 ? tabP : (which == 1 ? tabPmu : tabPy);
            tv[q] = w2_ld_g<STEP>(tab + (covoff + (long)bb * T + t0 + (ok ? j : 0)));
            wbv[q] = ok ? wb : -1;
        }
        // wave-uniform per-bin constants of this replicate: lane k * 8 + b loads constant k of bin b
        double cv = 0.0;
        {
            const int kq = lane >> 3, bq = lane & 7;  // kBMax <= 16: two rounds when B > 8
            const double* src = kq == 0 ? c.sf + r * B : (kq == 1 ? (use_bc ? c.sf_bc : c.sf) + r * B
                                                                  : (kq == 2 ? c.smask + r * B : c.P0));
            if (kq < 4 && bq < B && (MIX || kq != 3)) cv = src[bq];
        }
        tcol = c.g2t[gc] - t0;
        rgm = c.rg[rgc] != 0;
        const double a00 = c.a0[gc], a01 = use_bc ? c.a0_bc[gc] : 0.0;
        // the draw's first Philox block needs the step and the guide index only: computed here, while
        // the loads above are in flight (after a kernel boundary they take ~3.5 us and every wave of the
        // SIMD waits for them at the same time)
        if (MIX && !c.pi_in) philox_first = philox_block(c.seed, ((unsigned long long)kSitePi << 48) +
                                                                   ((unsigned long long)r * c.G_tot + (c.g_off + gc)),
                                                         ctr.step * 256ull);
        // control allele counts of the first control condition (a branch-free load: with no control condition
        // two words of pi_a0 are read in their place and dropped below)
        float alc0 = 0.f, alc1 = 0.f;
        if (MIX) {
            const float* al = c.C > 0 ? c.allele + ((long)r * c.C * G + gc) * 2 : (const float*)(c.pi_a0 + gc);
            alc0 = al[0];
            alc1 = al[1];
        }
        // ---- the draw, under the loads: it is a fifth of a wave's arithmetic and needs none of the counts or
        // tables; after a kernel boundary all waves of the chip issue their loads within 1.6 us and wait ~3.5 us
        // for them - with nothing to issue from, since every wave of a SIMD is at the same point
        asm volatile("" ::: "memory");  // (the loads above are issued here, not sunk below the draw to their uses)
        if (valid) guide_pair_draw<FAM>(c, ctr, r, g, api0, api1, pa0, &philox_first, ms + 4 * 64, ms + 5 * 64, pi0, pi1);
        // (opaque to the compiler, or it converts the two counts where they are loaded - a full wait before the draw)
        asm volatile("" : "+v"(alc0), "+v"(alc1));
        double cnt0 = c.C > 0 ? (double)alc0 : 0.0, cnt1 = c.C > 0 ? (double)alc1 : 0.0;
        if (MIX)
            for (int cc = 1; cc < c.C; ++cc) {
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
        for (int b = kBMax; b < B; ++b) {  // more conditions than the register batch holds (B <= kBCap)
            const long xo = ((long)r * B + b) * G + gc;
            xs[(0 * B + b) * 64 + lane] = c.X[xo];
            xs[(1 * B + b) * 64 + lane] = use_bc ? c.Xbc[xo] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < kTabLoads; ++q)
            if (wbv[q] >= 0) tabs[wbv[q] * ntm + j] = tv[q];
        for (int wb0 = kTabLoads * per; wb0 < n_rows; wb0 += per) {
            const int wb = wb0 + sub;
            if (wb < n_rows && j < nt) {
                const int which = (wb >= B) + (wb >= 2 * B), bb = wb - which * B;
                const double* tab = which == 0 ? c.tabP : (which == 1 ? c.tabPmu : c.tabPy);
                tabs[wb * ntm + j] = w2_ld<STEP>(tab + (covoff + (long)bb * T + t0 + j));
            }
        }
        {
            const int kq = lane >> 3, bq = lane & 7;
            if (kq < 4 && bq < B) cst[kq * B + bq] = cv;
            for (int b2 = 8 + bq; b2 < B; b2 += 8) {  // bins 8 .. B - 1
                if (kq < 4) {
                    const double* src = kq == 0 ? c.sf + r * B : (kq == 1 ? (use_bc ? c.sf_bc : c.sf) + r * B
                                                                          : (kq == 2 ? c.smask + r * B : c.P0));
                    cst[kq * B + b2] = (MIX || kq != 3) ? src[b2] : 0.0;
                }
            }
        }
        ms[0 * 64] = a00;
        ms[1 * 64] = a01;
        ms[2 * 64] = cnt0;
        ms[3 * 64] = cnt1;
    }
    __syncthreads();

    double a_mu = 0.0, a_y = 0.0;
    if (valid)
        loss = guide_pair_math<FAM, ACC, STEP, 64>(c, ctr, r, g, rgm, pi0, pi1, tabs + tcol, ntm, cst,
                                                   cst + 2 * B, cst + 3 * B, xs + lane, dps, ms, ms + 64, ms + 2 * 64,
                                                   ms + 3 * 64, ms + 4 * 64, ms + 5 * 64, a_mu, a_y);
    target_part_sums<STEP>(c, lane, tile, r, valid ? tcol : -1, valid, a_mu, a_y, t0, tof_first < g_first);
    tot_o = wave_sum(loss);
    BEAN_STAMP_AT(7);
    BEAN_STAMP_CLK(2);
}
#undef W2_ROW_STORE

// One wave of the 1-D grid of padded tiles x replicates: which (tile, replicate) it is - blocks b and b + 8 share an
// XCD (observed placement; a speed choice only), so the R waves of a tile are given ids that are equal modulo 8 -
// and its work.  Returns false for the padded tiles.
template <int FAM, bool ACC, int STEP>
__device__ __forceinline__ bool guide_wave2_body(const DevArgs& c, const StepCtr& ctr, int& tile_o, int& r_o,
                                                 int& t0_o, int& nt_o, double& tot_o) {
    const int wg = blockIdx.x;
    const int kk = wg >> 3;
    const int r = kk % c.R;
    const int tile = (kk / c.R) * 8 + (wg & 7);
    if (tile >= c.n_tiles) return false;
    tile_o = tile;
    r_o = r;
    guide_wave2_tile<FAM, ACC, STEP>(c, ctr, tile, r, wg, t0_o, nt_o, tot_o);
    return true;
}

template <int FAM, bool ACC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BEAN_WAVE_EU)))
void k_guide_wave2(DevArgs c) {
    const StepCtr ctr = *c.ctrB;
    int tile, r, t0, nt;
    double tot;
    if (!guide_wave2_body<FAM, ACC, 0>(c, ctr, tile, r, t0, nt, tot)) return;
    if (threadIdx.x == 0) {
        wave_loss_out(c, ctr.slot, blockIdx.x, tot);
        if (blockIdx.x == 0) publish_ctr(c, ctr);
    }
}

}  // namespace bean

// Tiling families with MORE alleles per guide than the register-resident kernels hold (kAMax = 8 / 16):
// the reference has no bound on n_max_alleles (bean/preprocessing/data_class.py:617-699,
// bean/model/model.py:550-751) - unfiltered allele tables carry hundreds per guide (230 in the
// reference's own tests/data/tiling_mini_screen.h5ad).  Here the ALLELES are the parallel axis:
//
//   k_guide_tiling_wide  one wave per (replicate, guide); lane l owns alleles l, l + 64, ... (up to
//                        kWideSlots per lane, i.e. 256 alleles per guide); sums over alleles are wave
//                        reductions; the B bins of a Dirichlet-Multinomial term are evaluated by lanes
//                        0 .. B-1 in parallel
//   param_guide_tiling_wide   the per-guide finish of k_param, one wave per guide
//
// Row layout (runtime A): 0 d/dnoise | 1 n unmasked replicates | 2 + a path_a | 2 + A + a log pi_a |
// 2 + 2A + s d/dmu of allele slot s | 2 + 2A + (A-1) + s d/dsigma of slot s; the rows of one (replicate,
// guide) are contiguous, trow is (R, G, Q) (trow_sum), and the allele tables are allele-contiguous (tab_off):
// round 4 - with the guide-contiguous layouts of the narrow kernels every load and store of this kernel
// touched 64 cache lines (5.8 % VALU-busy, 1.86 ms per launch at 5 000 guides x 232 slots).
// Same arithmetic as k_guide_tiling_wave; the random stream is keyed per PAIR of alleles (site, (replicate,
// guide) * 256 + allele of the lane's even slot: alleles l + 128 k and l + 128 k + 64 are drawn together), so it
// differs from the narrow kernels' - they never meet: the engine picks the wide path only when n_max_alleles
// exceeds what the narrow build holds.
#pragma once

namespace bean {

// alleles per lane: 4 in libbean_hip.so (256 alleles per guide), 8 in the builds that also hold more conditions
// (libbean_hip_a16.so / _a32.so: 512 alleles per guide - the engine loads them for tables beyond 256; the per-lane arrays
// double, and with them the spills: a table of that width is far outside what `bean filter` leaves, it has to run, not to
// run fast)
#if BEAN_AMAX > 8
constexpr int kWideSlots = 8;
#else
constexpr int kWideSlots = 4;
#endif
constexpr int kWideMaxA = 64 * kWideSlots;
// the draws of allele a of (replicate, guide) i are stream i * stride + a: 256 for every table this path ran before the
// wider build existed (the streams of those screens do not change, whichever build serves them), 1024 beyond
__host__ __device__ inline unsigned long long wide_stream_stride(int A) { return A <= 256 ? 256ull : 1024ull; }
__host__ __device__ inline int tq_path(int A) { (void)A; return 2; }
__host__ __device__ inline int tq_L(int A) { return 2 + A; }
__host__ __device__ inline int tq_gmu(int A) { return 2 + 2 * A; }
__host__ __device__ inline int tq_gsig(int A) { return 2 + 2 * A + (A - 1); }
__host__ __device__ inline int tq_num(int A) { return 2 + 2 * A + 2 * (A - 1); }

// sum over the wave, result in every lane: the DPP scan of wave_sum (fixed order; six row operations and a
// readlane instead of six trips through the LDS crossbar - this kernel takes ~30 such sums per wave)
__device__ __forceinline__ double wave_allsum(double v) { return wave_sum(v); }

template <bool ACC, bool SURV>
__global__ __launch_bounds__(64) void k_guide_tiling_wide(DevArgs c) {
    const int lane = threadIdx.x;
    const int G = c.G, A = c.A, A1 = c.A - 1, B = c.B, R = c.R;
    // the R waves of a guide read the same table columns: they get block ids that are equal modulo 8 (one
    // XCD, one L2) and adjacent in dispatch order, so the columns come from HBM once, not R times
    const int wg = blockIdx.x, kk = wg >> 3;
    const int r = kk % R, g = (kk / R) * 8 + (wg & 7);
    if (g >= G) return;
    const StepCtr ctr = *c.ctrB;
    // the rows of this (replicate, guide) are contiguous - row q at row[q] - so that lanes (= alleles) store
    // consecutive doubles: (R, G, Q)
    double* row = c.trow + ((long)r * G + g) * tq_num(A);
    constexpr long RG = 1;
    const bool rgm = c.rg[(long)r * G + g] != 0;
    const bool use_bc = (c.flags & kUseBc) != 0;
    double loss = 0.0;
    if (!rgm) {
        // both pi sites, the Multinomial and the count likelihoods are masked by repguide_mask in tiling
        // (model.py:659,682,731; guide 941): the replicate contributes nothing
        for (int q = lane; q < tq_num(A); q += 64) row[(long)q * RG] = 0.0;
        if (c.flags & kDumpPi)
            for (int a = lane; a < A; a += 64) c.pi_out[((long)r * G + g) * A + a] = 1.0 / A;
    } else {
        // ---- concentrations
        double alpha[kWideSlots], cq[kWideSlots], pi[kWideSlots];
        double asum = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            const bool am = a < A && c.amask[(long)g * A + a] != 0;
            alpha[j] = a < A ? (am ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
            asum += alpha[j];
        }
        const double Ssum = wave_allsum(asum);
        const double pa0 = c.pi_a0[g];
        const double rsq = frcp(Ssum) * pa0;
        double csum = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            cq[j] = alpha[j] * rsq;
            if (SURV && a < A && cq[j] < 1e-5) cq[j] = 1e-5;  // guide-side clamp (survival_model.py:813-821)
            csum += cq[j];
        }
        const double total = wave_allsum(csum);
        // ---- draw: two of the lane's alleles per rejection loop (slots j and j + 1 share one generator, keyed
        // by the first of the two alleles: one loop and one Philox counter per round instead of two of each)
        double gsum = 0.0;
        static_assert(kWideSlots % 2 == 0, "alleles are drawn in pairs of slots");
#pragma unroll
        for (int j = 0; j < kWideSlots; j += 2) {
            const int a = j * 64 + lane, a2 = a + 64;
            pi[j] = 0.0;
            pi[j + 1] = 0.0;
            if (c.pi_in) {
                if (a < A) pi[j] = c.pi_in[((long)r * G + g) * A + a];
                if (a2 < A) pi[j + 1] = c.pi_in[((long)r * G + g) * A + a2];
            } else if (a < A) {
                Rng rng(c.seed, kSitePi, ((unsigned long long)r * c.G_tot + guide_stream_id(c, g)) * wide_stream_stride(A) + a,
                        ctr.step * 256ull);
                // (a draw whose boost factor has already put it below DBL_MIN skips the rejection loop: with both
                // slots of the pair masked - concentrations of ~1e-6 - that is most waves; same values after the floor)
                const GammaPair gp = sample_gamma_pair_floord(cq[j], a2 < A ? cq[j + 1] : 1.0, rng);
                pi[j] = fmax(gp.g0, kDblMin);
                if (a2 < A) pi[j + 1] = fmax(gp.g1, kDblMin);
            }
            if (a < A) gsum += pi[j];
            if (a2 < A) gsum += pi[j + 1];
        }
        if (!c.pi_in) {
            const double rs = frcp(wave_allsum(gsum));
#pragma unroll
            for (int j = 0; j < kWideSlots; ++j)
                if (j * 64 + lane < A) pi[j] = fmin(fmax(pi[j] * rs, kDblMin), kOneMinus);
        }
        if (c.flags & kDumpPi) {
#pragma unroll
            for (int j = 0; j < kWideSlots; ++j)
                if (j * 64 + lane < A) c.pi_out[((long)r * G + g) * A + j * 64 + lane] = pi[j];
        }
        // ---- accessibility transform (utils.py:106-178)
        double pe[kWideSlots], dpe_dpi[kWideSlots], dpe_dl[kWideSlots];
        double pesum = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            pe[j] = pi[j];
            dpe_dpi[j] = 1.0;
            dpe_dl[j] = 0.0;
            if (ACC && a >= 1 && a < A) {
                const double kacc = c.kacc[g];
                const double s1 = pi[j] * kacc;
                const bool in1 = s1 > 1e-3 && s1 < 1.0 - 1e-3;
                const double p1c = fmin(fmax(s1, 1e-3), 1.0 - 1e-3);
                const double l = flog(p1c * frcp(1.0 - p1c)) + c.lpn[g];
                const double el = exp(l);
                const double pn = el * frcp(1.0 + el);
                const bool in2 = pn > 1e-3 && pn < 1.0 - 1e-3;
                pe[j] = fmin(fmax(pn, 1e-3), 1.0 - 1e-3);
                dpe_dl[j] = in2 ? pn * (1.0 - pn) : 0.0;
                dpe_dpi[j] = in1 ? dpe_dl[j] * frcp(p1c * (1.0 - p1c)) * kacc : 0.0;
            }
            if (a >= 1 && a < A) pesum += pe[j];
        }
        const double pe_edit = wave_allsum(pesum);
        // pi[0] of lane 0 is the unedited allele's draw
        const double pi_wt = __shfl(pi[0], 0, 64);
        const double pe0 = ACC ? 1.0 - pe_edit : pi_wt;
        const double u = SURV ? c.u_g[g] : 0.0;
        // ---- e[b] = pe0 P0[b] + sum_a pe_a P_a[b]: lane b keeps e[b]
        double e_mine = 0.0, p0_mine = 0.0;
        for (int b = 0; b < B; ++b) {
            double part = 0.0;
#pragma unroll
            for (int j = 0; j < kWideSlots; ++j) {
                const int a = j * 64 + lane;
                if (a >= 1 && a < A) part += pe[j] * c.tabP[tab_off(c, b, a - 1, g)];
            }
            const double p0b = SURV ? exp(u * c.time[b]) : c.P0[b];
            const double eb = wave_allsum(part) + pe0 * p0b;
            if (lane == b) {
                e_mine = eb;
                p0_mine = p0b;
            }
        }
        // ---- both Dirichlet-Multinomial terms: lane b evaluates bin b
        double nll_u = 0.0, nll_l = 0.0, ge_mine = 0.0;  // wave-uniform part / this lane's part of -log p
        const bool binlane = lane < B;
        const double smb = binlane ? c.smask[r * B + lane] : 0.0;
        const double epsB = kEps / (double)B;
        for (int lik = 0; lik < 2; ++lik) {
            if (lik == 1 && !use_bc) break;
            const float* X = lik ? c.Xbc : c.X;
            const double x = binlane ? (double)X[((long)r * B + lane) * G + g] : 0.0;
            const double sfb = binlane ? (lik ? c.sf_bc : c.sf)[r * B + lane] : 0.0;
            const double nn = wave_allsum(x);
            const double S = wave_allsum(e_mine * sfb);
            if (!(nn > (double)c.mask_thres)) continue;  // wave-uniform
            const double a0 = lik ? c.a0_bc[g] : c.a0[g];
            const double inv = frcp(S + kEps);
            const double araw = (e_mine * sfb + epsB) * inv * a0 * smb;
            const bool floored = araw < kEps;
            const double al = binlane ? (floored ? kEps : araw) : 0.0;
            const double A0 = wave_allsum(al);
            DD db;
            db.d = 0.0;
            db.dp = 0.0;
            if (binlane) db = lgamma_digamma_diff(al, x);
            const double lsum = wave_allsum(db.d);
            const double Ua = wave_allsum(binlane && !floored ? araw : 0.0);
            const double Va = wave_allsum(binlane && !floored ? db.dp * araw : 0.0);
            // total term: data unless a bin sits on its floor (DevArgs::tot_const)
            DD d0;
            d0.d = 0.0;
            d0.dp = 0.0;
            if (!c.tot_const) {
                d0 = lgamma_digamma_diff(A0, nn);
            } else if (__any(binlane && floored)) {
                const DD dt = lgamma_digamma_diff(A0, nn), dc = lgamma_digamma_diff(a0, nn);
                d0.d = dt.d - dc.d;
                d0.dp = dt.dp;
            }
            nll_u += d0.d - lsum;
            const double W = (d0.dp * Ua - Va) * inv;
            if (binlane) {
                const double ga = floored ? 0.0 : d0.dp - db.dp;
                ge_mine += (ga * a0 * smb * inv - W) * sfb;
            }
        }
        // ---- back through the mixture
        double s0 = 0.0;
        double sa[kWideSlots], dm[kWideSlots], dsg[kWideSlots];
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) sa[j] = dm[j] = dsg[j] = 0.0;
        for (int b = 0; b < B; ++b) {
            const double ge = __shfl(ge_mine, b, 64);
            s0 += ge * __shfl(p0_mine, b, 64);
#pragma unroll
            for (int j = 0; j < kWideSlots; ++j) {
                const int a = j * 64 + lane;
                if (a >= 1 && a < A) {
                    const long o = tab_off(c, b, a - 1, g);
                    sa[j] += ge * c.tabP[o];
                    dm[j] += ge * c.tabPmu[o];
                    if (!SURV) dsg[j] += ge * c.tabPy[o];
                }
            }
        }
        double gpi[kWideSlots], gm[kWideSlots];
        double gnoise = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            gpi[j] = 0.0;
            gm[j] = 0.0;
            if (a == 0) gpi[j] = ACC ? 0.0 : s0;
            if (a >= 1 && a < A) {
                gm[j] = pe[j] * dm[j];
                row[(long)(tq_gsig(A) + a - 1) * RG] = pe[j] * dsg[j];
                if (ACC) {
                    gpi[j] = (sa[j] - s0) * dpe_dpi[j];
                    gnoise += (sa[j] - s0) * dpe_dl[j];
                } else {
                    gpi[j] = sa[j];
                }
            }
        }
        // ---- Multinomial on control allele counts
        if (!SURV) {
            double psum = 0.0;
#pragma unroll
            for (int j = 0; j < kWideSlots; ++j)
                if (j * 64 + lane < A) psum += pi[j];
            const double s = wave_allsum(psum);
            const double ls = s == 1.0 ? 0.0 : flog(s);
            const double rsum = s == 1.0 ? 1.0 : frcp(s);
#pragma unroll
            for (int j = 0; j < kWideSlots; ++j) {
                const int a = j * 64 + lane;
                if (a < A) {
                    const double pr = pi[j] * rsum;
                    const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
                    const double lg = inside ? flog(pi[j]) - ls : flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                    double cnt = 0.0;
                    for (int cc = 0; cc < c.C; ++cc) cnt += (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                    nll_l -= cnt * lg;
                    if (inside) gpi[j] -= cnt * frcp(pi[j]);
                }
            }
        } else {
            // control_allele_count ~ Multinomial(pi * exp(mu * t_ctrl)), mu = [u, u + mu_a]
            // (survival_model.py:535-548)
            for (int cc = 0; cc < c.C; ++cc) {
                const double tc = c.ctrl_time[cc];
                double gr[kWideSlots], cnt[kWideSlots], wpart = 0.0;
#pragma unroll
                for (int j = 0; j < kWideSlots; ++j) {
                    const int a = j * 64 + lane;
                    gr[j] = 0.0;
                    cnt[j] = 0.0;
                    if (a < A) {
                        gr[j] = exp((a == 0 ? u : u + c.mu_a[slot_off(c, a - 1, g)]) * tc);
                        cnt[j] = (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                        wpart += pi[j] * gr[j];
                    }
                }
                const double rW = frcp(wave_allsum(wpart));
                double nin = 0.0;
#pragma unroll
                for (int j = 0; j < kWideSlots; ++j) {
                    const int a = j * 64 + lane;
                    if (a < A) {
                        const double pr = pi[j] * gr[j] * rW;
                        nll_l -= cnt[j] * flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                        if (pr > kProbEps && pr < 1.0 - kProbEps) nin += cnt[j];
                    }
                }
                const double n_in = wave_allsum(nin);
#pragma unroll
                for (int j = 0; j < kWideSlots; ++j) {
                    const int a = j * 64 + lane;
                    if (a < A) {
                        const double wv = pi[j] * gr[j], pr = wv * rW;
                        const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
                        gpi[j] += ((inside ? -cnt[j] * frcp(wv) : 0.0) + n_in * rW) * gr[j];
                        if (a >= 1) gm[j] += ((inside ? -cnt[j] : 0.0) + n_in * wv * rW) * tc;
                    }
                }
            }
        }
        // ---- Dirichlet log-density pieces, model-side floored concentration c_p (model.py:640-651)
        const double rSe = frcp(Ssum + kEps) * pa0;
        double pj = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            if (a < A) {
                const double rpi = frcp(pi[j]);
                row[(long)(tq_L(A) + a) * RG] = flog(pi[j]);
                gpi[j] += (cq[j] - 1.0) * rpi;  // + d log q / d pi
                const double v = (alpha[j] + kEps / A) * rSe;
                gpi[j] -= ((v < kEps ? kEps : v) - 1.0) * rpi;
                pj += pi[j] * gpi[j];
                if (a >= 1) row[(long)(tq_gmu(A) + a - 1) * RG] = gm[j];
            }
        }
        const double proj = wave_allsum(pj);
        // implicit-reparameterisation gradient: digamma(total) is the same for every allele of the guide - formed
        // once, not inside each of the four calls (same function, same bits as dirichlet_grad_one evaluates)
        const double dg_total = digamma(total);
        // ... and the masked alleles of a guide all have ONE concentration (alpha = eps): where a whole pass of 64
        // alleles is masked - every pass but the first, for a guide with fewer than 64 alleles - its digamma is the
        // one evaluation below instead of one per pass
        const double cq_masked = SURV ? fmax(kEps * rsq, 1e-5) : kEps * rsq;
        const double dg_masked = digamma(cq_masked);
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            const bool real = a < A && c.amask[(long)g * A + a] != 0;
            double dgj = dg_masked;
            if (__any(real)) dgj = digamma(cq[j]);  // (a masked lane of a mixed pass evaluates the same value)
            if (a < A)
                row[(long)(tq_path(A) + a) * RG] =
                    dirichlet_grad_one_pre(pi[j], cq[j], total, dgj, dg_total) * (gpi[j] - proj);
        }
        const double gn = wave_allsum(gnoise);
        if (lane == 0) {
            row[0] = gn;
            row[RG] = 1.0;
        }
        loss = nll_l + (lane == 0 ? nll_u : 0.0);  // the uniform part is counted once
    }
    const double tot = wave_sum(loss);
    if (lane == 0) {
        loss_add(c, ctr.slot, tot);
        if (blockIdx.x == 0) publish_ctr(c, ctr);
    }
    (void)A1;
}

// Guide part of k_param for the wide tiling path: one wave per guide, lane l owns alleles l, l + 64, ...
// Same algebra as param_guide_tiling (Dirichlet normalisers of the A-component pi site, chain to alpha_pi
// through the guide's and the model's concentration maps: model.py:938 and 646-651).
template <bool FINISH, bool ADAM, bool PREP>
__device__ __forceinline__ void param_guide_tiling_wide(const DevArgs& c, int guide_block,
                                                        unsigned long long s_prep, AdamCoef ak, double& loss_fin) {
    const int g = guide_block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool acc_on = (c.flags & kAcc) != 0;
    const bool fit_noise = acc_on && (c.flags & kFitNoise);
    const bool in = g < c.G;
    const int A = c.A;
    const bool lead = in && lane == 0;
    float nl = 0.f, ns_u = 0.f;
    if (lead && fit_noise) {
        nl = c.p[5][g];
        ns_u = c.p[6][g];
    }
    if (FINISH) {
        double alpha[kWideSlots], cq[kWideSlots], cp[kWideSlots];
        bool live[kWideSlots], am[kWideSlots], cqc[kWideSlots], cpc[kWideSlots];
        double asum = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            live[j] = in && a < A;
            am[j] = live[j] && c.amask[(long)g * A + a] != 0;
            alpha[j] = live[j] ? (am[j] ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
            asum += alpha[j];
        }
        const double S = wave_allsum(asum);
        const double pa0 = in ? c.pi_a0[g] : 1.0;
        const double rS = frcp(in ? S : 1.0), rSe = frcp((in ? S : 1.0) + kEps);
        const bool clampq = c.survival != 0;  // survival_model.py:813-821
        const double nrg = in ? trow_sum(c, 1, g) : 0.0;
        double sq_p = 0.0, sp_p = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const double cqr = alpha[j] * rS * pa0;
            cqc[j] = clampq && cqr < 1e-5;
            cq[j] = cqc[j] ? 1e-5 : cqr;
            const double cpr = (alpha[j] + kEps / A) * rSe * pa0;
            cpc[j] = cpr < kEps;
            cp[j] = cpc[j] ? kEps : cpr;
            if (live[j]) {
                sq_p += cq[j];
                sp_p += cp[j];
            }
        }
        const double sq = wave_allsum(sq_p), sp = wave_allsum(sp_p);
        double lgS_q = 0.0, dgS_q = 0.0, lgS_p = 0.0, dgS_p = 0.0;
        if (in) {
            lgamma_digamma(sq, lgS_q, dgS_q);
            lgamma_digamma(sp, lgS_p, dgS_p);
        }
        double gq[kWideSlots], gp[kWideSlots], dq_p = 0.0, dp_p = 0.0;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            gq[j] = 0.0;
            gp[j] = 0.0;
            if (live[j]) {
                double lg, dg;
                lgamma_digamma(cq[j], lg, dg);
                const double L = trow_sum(c, tq_L(A) + a, g);
                const double lq = -nrg * lg + (cq[j] - 1.0) * L;
                gq[j] = cqc[j] ? 0.0 : L + nrg * (dgS_q - dg) + trow_sum(c, tq_path(A) + a, g);
                lgamma_digamma(cp[j], lg, dg);
                const double lp = -nrg * lg + (cp[j] - 1.0) * L;
                gp[j] = cpc[j] ? 0.0 : -(L + nrg * (dgS_p - dg));
                loss_fin += lq - lp;
                dq_p += gq[j] * alpha[j];
                dp_p += gp[j] * (alpha[j] + kEps / A);
            }
        }
        if (lead) loss_fin += nrg * (lgS_q - lgS_p);
        const double dq = wave_allsum(dq_p) * rS * rS;
        const double dp = wave_allsum(dp_p) * rSe * rSe;
#pragma unroll
        for (int j = 0; j < kWideSlots; ++j) {
            const int a = j * 64 + lane;
            if (live[j]) {
                const double ga = pa0 * (gq[j] * rS - dq + gp[j] * rSe - dp);
                emit_grad<ADAM>(c, 4, (long)g * A + a, am[j] ? ga * alpha[j] : 0.0, ak);
            }
        }
        if (lead && acc_on) {
            const double lpn = c.lpn[g], eps = c.eps_noise[g];
            const double gl = trow_sum(c, 0, g);  // row 0: d/dnoise
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
}

}  // namespace bean

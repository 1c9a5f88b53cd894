// k_guide_survival_wave: the per-(replicate, guide) kernel of the survival variant families in the
// wave form of k_guide_wave2 (bean_guide_v2.hpp): one single-wave workgroup per (64-guide tile,
// replicate), rolled loops over the timepoints with thread-private LDS columns, per-replicate rows
// (summed over replicates by k_param), XCD-aware 1-D grid.  It replaces the block form
// k_guide_survival<B, ...> (R waves per block, per-thread arrays e[B], ge[B], P0[B], P1[B]: 216 VGPRs
// at six timepoints, two waves per SIMD), which stays selectable (BEAN_HIP_SURVIVAL=block) as the A/B
// reference.
//
// Reference semantics: bean/model/survival_model.py - NormalModel 15-130 (+ guide 629-648),
// ControlNormalModel 133-212, MixtureNormalModel 215-424 (+ guide 651-739): component "bin
// probabilities" are exp(mu_a t_b) with mu = [u_g, u_g + mu_t]; control_allele_count ~
// Multinomial(pi exp(mu t_ctrl)); the Dirichlet-over-all-guides site is handled per (rep, guide).
//
// Likelihood algebra (same as k_guide_wave2, with the total term first so that no per-timepoint
// value has to be kept): with alpha_b = max((e_b sf_b + eps/B) k m_b, eps), k = a0 / (S + eps):
//   pre-pass   S = sum e_b sf_b;  A0 = sum alpha_b;  d0 = lgamma/digamma difference of (A0, n)
//   main loop  ga_b = d0.dp - dpsi_b (0 on the floor);  S_Q += ga_b k m_b sf_b Q_b;  t_Q += sf_b Q_b;
//              Wa += ga_b alpha_b                         for Q in {P0, P1, t P1}
//   close      d nll / d(weight of Q) = S_Q - Wa inv t_Q
#pragma once

namespace bean {

// LDS per wave: the per-timepoint constants and the two growth columns only (6.3 KB at six timepoints).
// The counts and the per-guide constants used to be staged there too (11.4 KB: 14 single-wave workgroups
// per CU, fewer than the 16 its 128 VGPRs allow); the counts are read from global memory where they are
// used (coalesced rows, twice per step).  BASELINE config 5: 92.7 -> 85.8 us per step.  The kernel is
// VALU-bound (SQ_INSTS_VALU x 4 cycles / 1 024 SIMDs = 39 us) and its 4 689 waves still do not fit the
// 4 096 slots of four waves per SIMD; five waves per SIMD (96 VGPRs: 52 spilled) measured 88.2 us, six
// (80 VGPRs: 78 spilled) 96.3 us.  Two timepoints per iteration through lgamma_digamma_diff2 (the metric kernel's
// two side-by-side chains): 85.8 against 86.2-86.5 us, with 33 instead of 11 spilled registers - not adopted THEN;
// round 5, on the kernel without spills: adopted (the likelihood loop below).
// What the launch's 66 us are (scripts/stamps_surv_guide.py, every wave's start and end on the real-time clock):
// 4 096 waves are resident for the first 25 us; the four waves of a SIMD finish one after another (25, 37, 45,
// 53 us: the oldest wave issues first), and the 593 waves that are left start on the SIMDs that free up first and
// run until 66 us - 593 of the 1 024 SIMDs do five waves' work (5 x 13.2 us) and the others four.  Raising the late
// waves' priority (s_setprio) makes THEM finish early and the older waves of the same SIMDs late: 64.2 us.
// The launch is at its issue bound for whole-wave work items; the balance is a property of 4 689 / 1 024.
// (round 4) + kSurvPark thread-private columns in which values that are loaded with the first batch but needed
// only after the likelihoods wait (log of the observed t0 abundance, the gamma draw of the Dirichlet-over-guides
// site, pi_a0, alpha_pi, q0; with accessibility scaling the two derivatives of the transform): the kernel needed
// 128 VGPRs + 11 spilled (27 with accessibility) to carry them through the timepoint loops; 8.9 KB per wave
// still lets sixteen single-wave workgroups share a CU.
constexpr int kSurvPark = 7;
__host__ __device__ inline size_t guide_survival_wave_lds(int B) {
    return ((size_t)4 * B + (size_t)2 * B * 64 + (size_t)kSurvPark * 64) * sizeof(double);
}

#ifndef BEAN_SURV_EU
#define BEAN_SURV_EU 4
#endif
template <int FAM, bool ACC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BEAN_SURV_EU)))
void k_guide_survival_wave(DevArgs c) {
    constexpr bool MIX = FAM == kMixture;
    extern __shared__ double sls[];
    const int lane = threadIdx.x;
    const int G = c.G, B = c.B, R = c.R;
    const int wg = blockIdx.x;
    const int kk = wg >> 3;
    const int r = kk % R;
    const int tile = (kk / R) * 8 + (wg & 7);  // the R waves of a tile share an XCD (blockIdx % 8)
    if (tile * 64 >= G) return;
    const int g = tile * 64 + lane;
    const bool valid = g < G;
    const StepCtr ctr = *c.ctrB;
    double loss = 0.0;
    BEAN_STAMP_RT(wg, 5);  // (slots 5, 6: k_param's block roles use 0 ... 4 and 7 of the same records)

    // LDS: [4][B] sf, sf_bc, sample mask, time | [B][64] P0 | [B][64] P1
    double* cst = sls;
    double* p0s = cst + 4 * B + lane;              // exp(u t_b)          at p0s[b * 64]
    double* p1s = cst + 4 * B + B * 64 + lane;     // exp((u + mu_t) t_b) at p1s[b * 64]
    double* park = cst + 4 * B + 2 * B * 64 + lane;  // parked values at park[k * 64] (thread-private)
    enum { kPkLobs = 0, kPkGam = 1, kPkPa0 = 2, kPkApi = 3, kPkP7 = 4, kPkDpi = 5, kPkU = 6 };
    const bool use_bc = (c.flags & kUseBc) != 0;
    const bool q0lik = !MIX && c.surv_q0lik;
    bool rgm = false, negc = false;
    uint4 philox_first = make_uint4(0u, 0u, 0u, 0u);
    float api0 = 0.f, api1 = 0.f;
    double pa0 = 0.0, mu_t = 0.0, u = 0.0, gam = 0.0, a00 = 0.0, a01 = 0.0;
    double n_x = 0.0, n_bc = 0.0;
    double pi0 = 0.0, pi1 = 1.0;
    {
        const int gc = valid ? g : G - 1;
        const long rgc = (long)r * G + gc;
        // (first in the batch: loads return in issue order; the target index has a load depending on it, issued last
        // - and the draw below waits for the next three only)
        const int tix = c.g2t[gc];
        if (MIX) {
            api0 = c.p[4][2 * gc];
            api1 = c.p[4][2 * gc + 1];
            pa0 = c.pi_a0[gc];
        }
        // totals of the guide's counts over the timepoints (the loads in flight together; the likelihood
        // loop reads the counts again, one coalesced row per timepoint)
        float xv[2][kBMax];
#pragma unroll
        for (int b = 0; b < kBMax; ++b) {
            const long xo = ((long)r * B + (b < B ? b : B - 1)) * G + gc;
            xv[0][b] = c.X[xo];
            xv[1][b] = use_bc ? c.Xbc[xo] : 0.f;
        }
        double cv = 0.0;
        {
            const int kq = lane >> 3, bq = lane & 7;
            const double* src = kq == 0 ? c.sf + r * B : (kq == 1 ? (use_bc ? c.sf_bc : c.sf) + r * B
                                                                  : (kq == 2 ? c.smask + r * B : c.time));
            if (kq < 4 && bq < B) cv = src[bq];
        }
        rgm = c.rg[rgc] != 0;
        // survival NormalModel: mu of negative-control guides is forced to 0 (survival_model.py:59-60)
        negc = q0lik && c.negctrl && c.negctrl[gc] != 0;
        a00 = c.a0[gc];
        a01 = use_bc ? c.a0_bc[gc] : 0.0;
        // the pi draw's first Philox block, under the latency of the loads above (as in k_guide_wave2)
        if (MIX && !c.pi_in) philox_first = philox_block(c.seed, ((unsigned long long)kSitePi << 48) +
                                                                   ((unsigned long long)r * c.G_tot + (c.g_off + gc)),
                                                         ctr.step * 256ull);
        double lobs = 0.0;
        float p7 = 0.f;
        if (MIX) {
            u = c.u_g[gc];
            lobs = c.log_obs0[rgc];
        }
        if (MIX || q0lik) {
            p7 = c.p[7][gc];
            gam = c.gam[rgc];
        }
        // ---- the pi draw, under the loads (as in k_guide_wave2: it needs alpha_pi and pi_a0 only; draws handed in
        // - DevArgs::pi_in, tests - are read further down, a load on either side of this branch would drain the
        // loads in flight first)
        mu_t = c.mu_t[tix];
        asm volatile("" ::: "memory");
        if (MIX && valid && !c.pi_in) {
            const double al0 = (double)expf(api0), al1 = (double)expf(api1);
            const double rs = frcp(al0 + al1) * pa0;
            const double cp0 = al0 * rs, cp1 = al1 * rs;
            const double cq0 = cp0 < 1e-5 ? 1e-5 : cp0, cq1 = cp1 < 1e-5 ? 1e-5 : cp1;
            Rng rng(c.seed, kSitePi, (unsigned long long)r * c.G_tot + (c.g_off + g), ctr.step * 256ull);
            const GammaPair gp = sample_gamma_pair_inl(cq0, cq1, rng, &philox_first);
            const double gm0 = fmax(gp.g0, kDblMin), gm1 = fmax(gp.g1, kDblMin);
            const double rs2 = frcp(gm0 + gm1);
            pi0 = fmin(fmax(gm0 * rs2, kDblMin), kOneMinus);
            pi1 = fmin(fmax(gm1 * rs2, kDblMin), kOneMinus);
        }
        // (the counts as sixteen values in flight: left alone the compiler adds each one up where it is loaded,
        // one full wait per timepoint)
#pragma unroll
        for (int b = 0; b < kBMax; ++b) asm volatile("" : "+v"(xv[0][b]), "+v"(xv[1][b]));
#pragma unroll
        for (int b = 0; b < kBMax; ++b) {
            if (b < B) {
                n_x += (double)xv[0][b];
                n_bc += (double)xv[1][b];
            }
        }
        for (int b = kBMax; b < B; ++b) {  // more timepoints than the register batch holds (B <= kBCap)
            const long xo = ((long)r * B + b) * G + gc;
            n_x += (double)c.X[xo];
            n_bc += use_bc ? (double)c.Xbc[xo] : 0.0;
        }
        {
            const int kq = lane >> 3, bq = lane & 7;
            if (kq < 4 && bq < B) cst[kq * B + bq] = cv;
            for (int b2 = 8 + bq; b2 < B; b2 += 8) {  // timepoints 8 .. B - 1
                if (kq < 4) {
                    const double* src = kq == 0 ? c.sf + r * B : (kq == 1 ? (use_bc ? c.sf_bc : c.sf) + r * B
                                                                          : (kq == 2 ? c.smask + r * B : c.time));
                    cst[kq * B + b2] = src[b2];
                }
            }
        }
        // needed after the likelihoods only: parked (MixtureNormal: the gamma draw too - NormalModel uses it
        // as a weight of the likelihood right away)
        // (+Acc: the draw pi_0, pi_1 is not a weight of the likelihood and takes the first two columns; the log
        // abundance and the gamma draw are loaded where they are used)
        if (!(MIX && ACC)) park[kPkLobs * 64] = lobs;
        park[kPkP7 * 64] = (double)p7;
        if (MIX) {
            if (!ACC) park[kPkGam * 64] = gam;
            park[kPkPa0 * 64] = pa0;
            park[kPkApi * 64] = __builtin_bit_cast(double, ((unsigned long long)__builtin_bit_cast(unsigned int, api1) << 32) |
                                                               (unsigned long long)__builtin_bit_cast(unsigned int, api0));
        }
    }
    __syncthreads();

    if (valid) {
        const long rgi = (long)r * G + g;
        const long RG = (long)R * G;
        const double* c_sf = cst;
        const double* c_sm = cst + 2 * B;
        const double* c_tm = cst + 3 * B;
        if (negc) mu_t = 0.0;
        const double mu1 = u + mu_t;
        // growth of the two components over the timepoints: thread-private LDS columns
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            const double tb = c_tm[b];
            p1s[b * 64] = exp(mu1 * tb);
            if (MIX) p0s[b * 64] = exp(u * tb);
        }
        if (MIX) park[kPkU * 64] = u;
        // (a compiler-level fence after a park: without it the stored value is forwarded to its later load, i.e.
        // stays in its register, and nothing is gained)
        asm volatile("" ::: "memory");
        double pe1 = 1.0, dpe1_dpi1 = 0.0, dpe1_dl = 0.0;
        if (MIX) {
            if (c.pi_in) {
                pi0 = c.pi_in[rgi * 2];
                pi1 = c.pi_in[rgi * 2 + 1];
            }
            if (c.flags & kDumpPi) {
                c.pi_out[rgi * 2] = pi0;
                c.pi_out[rgi * 2 + 1] = pi1;
            }
            pe1 = pi1;
            if (ACC) {
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
                park[kPkDpi * 64] = dpe1_dpi1;  // (d pe1 / d l is pe1 (1 - pe1) inside the clamp, 0 on it: re-formed below)
                park[kPkLobs * 64] = pi0;
                park[kPkGam * 64] = pi1;
                asm volatile("" ::: "memory");
            }
        }
        // the draw of the Dirichlet-over-all-guides site, float32 semantics of torch's sampler:
        // normalise, clamp to [FLT_MIN, 1 - 2^-24] (MixtureNormal: formed after the likelihoods, where it is used)
        double x0 = 1.0;
        if (q0lik) {
            x0 = c.x0_in ? gam
                         : (double)fminf(fmaxf((float)(gam * frcp(c.gsum[r])), 1.17549435e-38f), 0.99999994f);
            if (c.x0_out) c.x0_out[rgi] = x0;
        }
        // weights of the two growth columns in e_b = w0 P0_b + w1 P1_b
        const double w0 = MIX ? (ACC ? 1.0 - pe1 : pi0) : 0.0;
        const double w1 = MIX ? (ACC ? pe1 : pi1) : (q0lik ? x0 : 1.0);
        const double epsB = kEps / (double)B;
        double g0 = 0.0, g1 = 0.0, dmu = 0.0, nll = 0.0;
#pragma unroll 1
        for (int lik = 0; lik < 2; ++lik) {
            if (lik == 1 && !use_bc) break;
            const double nn = lik ? n_bc : n_x;
            if (!(rgm && nn > (double)c.mask_thres)) continue;
            const float* xp = (lik ? c.Xbc : c.X) + (long)r * B * G + g;  // timepoint b at xp[b * G]
            const double* sf = c_sf + lik * B;
            // S = sum_b e_b sf_b of THIS likelihood (formed here, not for both ahead of the loop: two
            // registers fewer to carry through the first likelihood)
            double S = 0.0;
#pragma unroll 1
            for (int b = 0; b < B; ++b) S += fma(w0, MIX ? p0s[b * 64] : 0.0, w1 * p1s[b * 64]) * sf[b];
            const double a0 = lik ? a01 : a00;
            const double inv = frcp(S + kEps);
            const double ai = a0 * inv;
            double A0 = 0.0;
            bool anyfl = false;
#pragma unroll 1
            for (int b = 0; b < B; ++b) {
                const double araw = alpha_raw(w0, MIX ? p0s[b * 64] : 0.0, w1, p1s[b * 64], sf[b], epsB, ai * c_sm[b]);
                anyfl = anyfl || araw < kEps;
                A0 += araw < kEps ? kEps : araw;
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
            double lsum = 0.0, Wa = 0.0;
            double S_0 = 0.0, S_1 = 0.0, S_t = 0.0, t_0 = 0.0, t_1 = 0.0, t_t = 0.0;
            int b = 0;
#ifndef BEAN_SURV_SINGLE_BINS
            // two timepoints per pass: their lgamma / digamma differences side by side (lgamma_digamma_diff2, the same
            // operations per chain); every sum takes the two in order - same bits.  Round 3 measured this form at 128 VGPRs
            // + 33 spilled (no gain); after round 4's register diet, and with the growth columns read again behind the chains
            // instead of carried across them, it is 115 - 117 VGPRs, nothing spilled, and config 5 steps in 80.5 - 80.9 us
            // against 82.7 - 83.4 (kernel 59.5 against 62 - 65; -DBEAN_SURV_SINGLE_BINS: one per pass)
            auto acc_bin = [&](double p0, double p1, double sfb, double tb, double km, double araw, bool floored, const DD& db) {
                lsum += db.d;
                const double ga = floored ? 0.0 : d0.dp - db.dp;
                Wa += ga * araw;
                const double cb = ga * km * sfb;
                const double tp1 = tb * p1;
                S_1 += cb * p1;
                t_1 += sfb * p1;
                S_t += cb * tp1;
                t_t += sfb * tp1;
                if (MIX) {
                    S_0 += cb * p0;
                    t_0 += sfb * p0;
                }
            };
#pragma unroll 1
            for (; b + 1 < B; b += 2) {
                double arawa, arawb;
                {
                    const double p0a = MIX ? p0s[b * 64] : 0.0, p1a = p1s[b * 64];
                    const double p0b = MIX ? p0s[(b + 1) * 64] : 0.0, p1b = p1s[(b + 1) * 64];
                    arawa = alpha_raw(w0, p0a, w1, p1a, sf[b], epsB, ai * c_sm[b]);
                    arawb = alpha_raw(w0, p0b, w1, p1b, sf[b + 1], epsB, ai * c_sm[b + 1]);
                }
                const bool fla = arawa < kEps, flb = arawb < kEps;
                const DD2 dd = lgamma_digamma_diff2(fla ? kEps : arawa, (double)xp[(long)b * G], flb ? kEps : arawb,
                                                    (double)xp[(long)(b + 1) * G]);
                // (the columns are read AGAIN behind the two chains - LDS reads - instead of carried across them: the
                // barrier makes them new values to the compiler; carried, they were four spilled registers)
                asm volatile("" ::: "memory");
                acc_bin(MIX ? p0s[b * 64] : 0.0, p1s[b * 64], sf[b], c_tm[b], ai * c_sm[b], arawa, fla, dd.a);
                acc_bin(MIX ? p0s[(b + 1) * 64] : 0.0, p1s[(b + 1) * 64], sf[b + 1], c_tm[b + 1], ai * c_sm[b + 1], arawb, flb, dd.b);
            }
#endif
#pragma unroll 1
            for (; b < B; ++b) {
                const double p0 = MIX ? p0s[b * 64] : 0.0, p1 = p1s[b * 64];
                const double sfb = sf[b], tb = c_tm[b];
                const double km = ai * c_sm[b];
                const double araw = alpha_raw(w0, p0, w1, p1, sfb, epsB, km);
                const bool floored = araw < kEps;
                const DD db = lgamma_digamma_diff_inl(floored ? kEps : araw, (double)xp[(long)b * G]);
                lsum += db.d;
                const double ga = floored ? 0.0 : d0.dp - db.dp;
                Wa += ga * araw;
                const double cb = ga * km * sfb;
                const double tp1 = tb * p1;
                S_1 += cb * p1;
                t_1 += sfb * p1;
                S_t += cb * tp1;
                t_t += sfb * tp1;
                if (MIX) {
                    S_0 += cb * p0;
                    t_0 += sfb * p0;
                }
            }
            nll += d0.d - lsum;
            const double W = Wa * inv;
            g0 += S_0 - W * t_0;
            g1 += S_1 - W * t_1;
            dmu += S_t - W * t_t;
        }
        double* row = c.wrow + rgi;  // row q of this replicate at row[q * RG]
        // d nll / d mu_t: through the edited column, whose weight is w1
        double gmu = negc ? 0.0 : w1 * dmu;
        if (q0lik) {
            // survival NormalModel: - log p(q_0) + log q(q_0) = (ia - 1/G) log x per guide (normalisers
            // in k_param); d loss / d q_0 for the pathwise gradient of the G-dimensional Dirichlet
            const double ia = (double)expf((float)park[kPkP7 * 64]);
            const double dconc = ia - (c.prior_ia ? c.prior_ia[g] : (double)(1.0f / (float)c.G_tot));
            const double lx = flog(x0);
            nll += dconc * lx;
            c.gq[rgi] = g1 + dconc * frcp(x0);
            row[kPQ0 * RG] = lx;
        }
        if (MIX) {
            // the concentrations again (as before the draw: same expression, same bits), from the parked inputs
            const unsigned long long apb = __builtin_bit_cast(unsigned long long, park[kPkApi * 64]);
            const float api0r = __builtin_bit_cast(float, (unsigned int)apb);
            const float api1r = __builtin_bit_cast(float, (unsigned int)(apb >> 32));
            const double al0 = (double)expf(api0r), al1 = (double)expf(api1r);
            const double rs = frcp(al0 + al1) * park[kPkPa0 * 64];
            const double cp0 = al0 * rs, cp1 = al1 * rs;
            const bool cl0 = cp0 < 1e-5, cl1 = cp1 < 1e-5;
            const double cq0 = cl0 ? 1e-5 : cp0, cq1 = cl1 ? 1e-5 : cp1;
            double gpi0 = g0, gpi1 = g1;
            if (ACC) {
                pi0 = park[kPkLobs * 64];
                pi1 = park[kPkGam * 64];
                gpi0 = 0.0;
                gpi1 = (g1 - g0) * park[kPkDpi * 64];
                row[kW2Gnoise * RG] = (g1 - g0) * ((pe1 > 1e-3 && pe1 < 1.0 - 1e-3) ? pe1 * (1.0 - pe1) : 0.0);
            }
            // digamma of the concentrations, tabulated by k_param (DevArgs::dgq): issued here, used
            // by the implicit-gradient calls below
            const double dgS = c.dgq[3 * (long)G + g], dg0 = c.dgq[4 * (long)G + g], dg1 = c.dgq[5 * (long)G + g];
            const double lpi0 = flog(pi0), lpi1 = flog(pi1);
            const double rpi0 = frcp(pi0), rpi1 = frcp(pi1);
            if (rgm) {
                // control_allele_count ~ Multinomial(pi * exp(mu * t_ctrl)) (survival_model.py:326-346)
                const double ur = park[kPkU * 64], mu1r = ur + mu_t;
                for (int cc = 0; cc < c.C; ++cc) {
                    const double tc = c.ctrl_time[cc];
                    const double gr0 = exp(ur * tc), gr1 = exp(mu1r * tc);
                    const double wv0 = pi0 * gr0, wv1 = pi1 * gr1;
                    const double rW = frcp(wv0 + wv1);
                    const float* al = c.allele + (((long)r * c.C + cc) * G + g) * 2;
                    const double cnt0 = (double)al[0], cnt1 = (double)al[1];
                    const double pr0 = wv0 * rW, pr1 = wv1 * rW;
                    const bool in0 = pr0 > kProbEps && pr0 < 1.0 - kProbEps;
                    const bool in1 = pr1 > kProbEps && pr1 < 1.0 - kProbEps;
                    nll -= cnt0 * flog(fmin(fmax(pr0, kProbEps), 1.0 - kProbEps));
                    nll -= cnt1 * flog(fmin(fmax(pr1, kProbEps), 1.0 - kProbEps));
                    const double n_in = (in0 ? cnt0 : 0.0) + (in1 ? cnt1 : 0.0);
                    gpi0 += ((in0 ? -cnt0 * frcp(wv0) : 0.0) + n_in * rW) * gr0;
                    gpi1 += ((in1 ? -cnt1 * frcp(wv1) : 0.0) + n_in * rW) * gr1;
                    gmu += ((in1 ? -cnt1 : 0.0) + n_in * wv1 * rW) * tc;
                }
                gpi0 -= (cp0 - 1.0) * rpi0;
                gpi1 -= (cp1 - 1.0) * rpi1;
                nll -= (cp0 - 1.0) * lpi0 + (cp1 - 1.0) * lpi1;
            }
            gpi0 += (cq0 - 1.0) * rpi0;
            gpi1 += (cq1 - 1.0) * rpi1;
            nll += (cq0 - 1.0) * lpi0 + (cq1 - 1.0) * lpi1;
            const double proj = pi0 * gpi0 + pi1 * gpi1;
            const double total = cq0 + cq1;
            double path0 = 0.0, path1 = 0.0;
            const int first = pi0 <= pi1 ? 0 : 1;  // smaller component first: see k_guide_wave2
#pragma unroll 1
            for (int pass = 0; pass < 2; ++pass) {
                const int a = pass ^ first;
                if (a ? cl1 : cl0) continue;
                const double v = dirichlet_grad_one_pre(a ? pi1 : pi0, a ? cq1 : cq0, total, a ? dg1 : dg0, dgS) *
                                 ((a ? gpi1 : gpi0) - proj);
                path0 = a ? path0 : v;
                path1 = a ? v : path1;
            }
            row[kW2GA0 * RG] = (cl0 ? 0.0 : lpi0 + path0) - (rgm ? lpi0 : 0.0);
            row[kW2GA1 * RG] = (cl1 ? 0.0 : lpi1 + path1) - (rgm ? lpi1 : 0.0);
            // ---- Dirichlet(q0) site: + log q(x) of the guide's draw, - log p(obs) of the model
            // (observed in the model, sampled in the guide: survival_model.py:306-311, 665-669)
            {
                const double gamr = ACC ? c.gam[rgi] : park[kPkGam * 64];
                x0 = c.x0_in ? gamr
                             : (double)fminf(fmaxf((float)(gamr * frcp(c.gsum[r])), 1.17549435e-38f), 0.99999994f);
                if (c.x0_out) c.x0_out[rgi] = x0;
                const double q0 = (double)expf((float)park[kPkP7 * 64]);
                const double lx = flog(x0);
                const double tot0 = c.gsum[R];
                const double lobs = ACC ? c.log_obs0[rgi] : park[kPkLobs * 64];
                nll += (q0 - 1.0) * (lx - lobs);
                const double gout = (q0 - 1.0) * frcp(x0);
                const double Sx = tot0 - (double)c.G_tot;  // sum_g x_g * gout_g
                row[kPQ0 * RG] = lx - lobs + dirichlet_grad_one(x0, q0, tot0) * (gout - Sx);
            }
        }
        row[kPGmu * RG] = gmu;
        loss = nll;
    }
    const double tot = wave_sum(loss);
    BEAN_STAMP_RT(wg, 6);
    if (lane == 0) {
        wave_loss_out(c, ctr.slot, wg, tot);
        if (wg == 0) publish_ctr(c, ctr);
    }
}

}  // namespace bean

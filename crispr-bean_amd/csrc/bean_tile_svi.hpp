// k_svi_tile (OPT-IN: BEAN_HIP_STEP=tile): the SVI loop of the variant sorting families (Normal,
// MixtureNormal(+Acc)) as ONE launch per call of bean_hip_svi_run - a workgroup owns a tile of targets for
// all the steps of the call.
//
// Why it was built: in these families every parameter is per target (mu_loc, mu_scale, sd_loc, sd_scale) or
// per guide (alpha_pi, the accessibility noise), and a target's likelihood sees only its own guides
// (bean/model/model.py:378-547, 785-858): apart from the scalar loss that is REPORTED, nothing couples two
// targets.  The reference steps the whole screen in lock step because Pyro traces one model; the two-launch
// path ({k_param, k_guide_wave2} per step) inherited that, and at the metric shape pays per step ~13.5 us of
// launch head / tail in the guide kernel plus a latency-bound k_param launch.  A tile of targets can instead
// run ahead on its own: its workgroup loops over the steps, nothing is launched per step, and the only
// inter-workgroup traffic is the loss (order-independent integer atomics).
//
// Why it is not the default - measured, same box, us per step (tile / two launches):
//      5k guides 32.0 / 29.0     25k 45.9 / 37.6     50k 72.4 / 57.4     62.5k 85.0 / 67.5     500k 654 / 362
// A lone tile-step is a 32 us chain (guide_pair_math alone is ~19 us for a wave without neighbours; then the
// finish phase with one or two of the workgroup's four waves active; table / count staging and four
// argument reloads per step), and with four workgroups per CU the SIMDs are short of runnable waves exactly
// as in the fused step kernel (bean_step_v2.hpp): every wave spends a third of its cycle in latency-bound
// phases that k_param, as a launch of its own, runs chip-wide in 10 us.  Workgroups of 128 threads (the
// finish phase then keeps both waves busy) measured the same (74 us at 50k).  What a launch boundary buys is
// width for the latency-bound part, once more.  Kept because it is bit-identical to the default path
// (tests/test_gpu_tile_svi.py) and pins guide_pair_math, the shared body of all three kernels.
//
// Tile = consecutive whole targets with at most Gb = 256 / R guides in total (host-built table, never
// straddling a target), workgroup = 256 threads, thread = r * ng + j: ALL replicates of the tile's ng
// guides (the layout of k_guide_tiling_rep: 250 of 256 lanes at 5 guides per target x 5 replicates).  A
// workgroup takes tiles blockIdx.x, blockIdx.x + gridDim.x, ... one after the other, each for all steps.
// Per step:
//   guide phase    per (replicate, guide): guide_pair_math (bean_guide_v2.hpp) - the very function
//                  k_guide_wave2 runs - on tables / counts staged in LDS every step (L2 hits), per-guide data and
//                  per-replicate constants staged ONCE per tile; rows go to global memory with agent-scope stores
//   finish phase   what k_param does for the tile's targets and guides (FINISH of this step, PREP of the
//                  next): four lanes per target (sums in k_param's 16-lane order, priors, ClippedAdam, draw),
//                  the Phi tables from the distinct bin edges, param_guide_mix per guide - the code of the
//                  one-launch step kernel (bean_step_v2.hpp), spread over the workgroup; targets and guides
//                  run in different waves at the same time.
// Every phase is an out-of-line function with its own register allocation (inlined into the step loop the
// compiler hoists each step-invariant value of guide_pair_math above both loops: 254 spilled VGPRs), reading
// DevArgs from a copy in global memory into SGPRs (bean_devargs_sgpr.hpp).
// Same arithmetic, same draws (Philox keyed by global indices and the step), same summation orders:
// parameters are bit-identical to the two-launch path; the loss history agrees to the 2^-40 granule of its
// fixed-point partial sums.
#pragma once

#include "bean_devargs_sgpr.hpp"

namespace bean {

constexpr int kTileThreads = 256;

// dynamic LDS of one workgroup (doubles first, then the float counts)
__host__ __device__ inline size_t svi_tile_lds(int B, int R, int ntm, int gbm) {
    const int Bd = B < 3 ? 3 : B;  // the finish phase parks three edge arrays in the digamma columns
    return ((size_t)3 * B * ntm + (size_t)R * 4 * B + (size_t)Bd * kTileThreads + (size_t)5 * gbm +
            (size_t)2 * kTileThreads + 16) * sizeof(double) +
           (size_t)kTileThreads * sizeof(int) + (size_t)2 * B * kTileThreads * sizeof(float);
}

// DevArgs for the out-of-line pieces: from the copy bean_hip_svi_run keeps in global memory, into SGPRs
// (bean_devargs_sgpr.hpp; passing the 700-byte struct by value would go through scratch instead)
#define BEAN_TILE_ARGS(cp) const DevArgs c = dev_args_in_sgprs(cp)

// FINISH of the step (and PREP of the next) for the tile's targets: returns the thread's loss terms
__device__ __noinline__ double tile_finish_targets(const DevArgs* cp, int t0, int nt, int ntm, unsigned long long s_prep,
                                                   float step_size, int prep) {
    BEAN_TILE_ARGS(cp);
    constexpr int NT = kTileThreads;
    extern __shared__ double sm[];
    t0 = __builtin_amdgcn_readfirstlane(t0);
    nt = __builtin_amdgcn_readfirstlane(nt);
    ntm = __builtin_amdgcn_readfirstlane(ntm);
    prep = __builtin_amdgcn_readfirstlane(prep);
    const int tid = threadIdx.x;
    const int G = c.G, R = c.R;
    AdamCoef ak;
    ak.step_size = step_size;
    ak.clip = (float)c.clip;
    double loss_fin = 0.0;
    double* hmu = sm;            // drawn mu / y of the tile's targets (the tables are dead until phase C)
    double* hy = sm + ntm;
    // ---- phases A + B: four lanes per target (lane q: unconstrained parameter q), NT / 4 targets per pass
    {
        const int q = tid & 3;
        float* const P = q == 0 ? c.p[0] : (q == 1 ? c.p[1] : (q == 2 ? c.p[2] : c.p[3]));
        float* const M = q == 0 ? c.m[0] : (q == 1 ? c.m[1] : (q == 2 ? c.m[2] : c.m[3]));
        float* const V = q == 0 ? c.v[0] : (q == 1 ? c.v[1] : (q == 2 ? c.v[2] : c.v[3]));
        for (int base = 0; base < nt; base += NT / 4) {
            const int tl = base + (tid >> 2);
            const bool act = tl < nt;
            if (!__any(act)) continue;  // whole waves without a target go straight to the guides below
            const int tc = t0 + (act ? tl : nt - 1);
            const int tg0 = c.toff[tc], tng = c.toff[tc + 1] - tg0, n = tng * R;
            float pj = P[tc], mj = M[tc], vj = V[tc];
            const float p1 = c.p[1][tc], p3 = c.p[3][tc];
            const double eps1 = c.eps_mu[tc], eps2 = c.eps_sd[tc], mu = c.mu_t[tc], y = c.y_t[tc];
            // the (guide, replicate) rows of the target in k_param's order: its 16 lanes take entries
            // lg, lg + 16, ... and combine by an xor tree (8, 4, 2, 1); lane q here plays lanes q + 4 k
            double am[4] = {0.0, 0.0, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.0, 0.0};
            const float rng = 1.0f / (float)tng;  // i / tng below: exact for i < 2^20
            for (int i0 = 0; i0 < n; i0 += 32) {
                double xm[2][4], xy[2][4];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + 16 * mm + q + 4 * k;
                        xm[mm][k] = 0.0;
                        xy[mm][k] = 0.0;
                        if (act && i < n) {
                            const int rr = (int)(((float)i + 0.5f) * rng), gg = tg0 + (i - rr * tng);
                            xm[mm][k] = row_ld<true>(c.wrow + ((long)kPGmu * R + rr) * G + gg);
                            xy[mm][k] = row_ld<true>(c.wrow + ((long)kPGy * R + rr) * G + gg);
                        }
                    }
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + 16 * mm + q + 4 * k;
                        if (act && i < n) {  // (an absent entry must not add +0.0 to a -0.0 sum)
                            am[k] += xm[mm][k];
                            ay[k] += xy[mm][k];
                        }
                    }
            }
            double gmu = (am[0] + am[2]) + (am[1] + am[3]);
            double gy = (ay[0] + ay[2]) + (ay[1] + ay[3]);
            gmu += __shfl_xor(gmu, 2, 64);
            gy += __shfl_xor(gy, 2, 64);
            gmu += __shfl_xor(gmu, 1, 64);
            gy += __shfl_xor(gy, 1, 64);
            // FINISH of this step
            double dlogp_mu, dlogp_dy, lt;
            tgt_prior_terms(c, tc, tgt_sd_prior(c, tc), mu, y, eps1, eps2, p1, p3, dlogp_mu, dlogp_dy, lt);
            if (act && q == 0) loss_fin += lt;
            const double Gd = q < 2 ? gmu - dlogp_mu : gy - dlogp_dy;
            const double grad = tgt_grad(q, Gd, q < 2 ? eps1 : eps2, exp((double)pj));
            adam_update(pj, mj, vj, (float)grad, ak);
            if (act) {
                P[tc] = pj;
                M[tc] = mj;
                V[tc] = vj;
            }
            if (prep) {
                // PREP of the next step: the draw (Philox keyed by the global target index and the step)
                const float2 nrm = normal2_at(c.seed, ((unsigned long long)kSiteTarget << 48) + (unsigned long long)(c.t_off + tc),
                                              s_prep * 4ull);
                const double en = q < 2 ? (double)nrm.x : (double)nrm.y;
                const float p_scale = __shfl_xor(pj, 1, 64);  // even lanes: the updated log scale of their pair
                const double val = tgt_draw(pj, en, p_scale);
                if (act && (q & 1) == 0) {
                    (q == 0 ? c.eps_mu : c.eps_sd)[tc] = en;
                    (q == 0 ? c.mu_t : c.y_t)[tc] = val;
                    (q == 0 ? hmu : hy)[tl] = val;
                }
            }
        }
    }
    return loss_fin;
}

// ... for the tile's guides, in the LAST threads of the workgroup (other waves than the targets wherever
// both fit): alpha_pi (and the accessibility noise site), lgamma tables of the next step
__device__ __noinline__ double tile_finish_guides(const DevArgs* cp, int g0, int ng, unsigned long long s_prep,
                                                  float step_size, int prep) {
    BEAN_TILE_ARGS(cp);
    constexpr int NT = kTileThreads;
    constexpr bool MIX = true;
    g0 = __builtin_amdgcn_readfirstlane(g0);
    ng = __builtin_amdgcn_readfirstlane(ng);
    prep = __builtin_amdgcn_readfirstlane(prep);
    const int tid = threadIdx.x;
    AdamCoef ak;
    ak.step_size = step_size;
    ak.clip = (float)c.clip;
    double loss_fin = 0.0;
    // ---- the tile's guides, in the LAST threads of the workgroup (other waves than the targets
    // wherever both fit): alpha_pi (and the accessibility noise site), lgamma tables of the next step
    if (MIX) {
        for (int jb = 0; jb < ng; jb += NT) {
            const int jg = jb + (NT - 1 - tid);
            double lg = 0.0;  // (param_guide_mix assigns its loss terms)
            if (jg < ng) {
                if (prep) param_guide_mix<true, true, true, true>(c, g0 + jg, ak, s_prep, lg);
                else param_guide_mix<true, true, false, true>(c, g0 + jg, ak, s_prep, lg);
            }
            loss_fin += lg;
        }
    }
    return loss_fin;
}

// Phase C: the Phi tables of the new draws (hmu / hy in LDS) of the tile's targets
__device__ __noinline__ void tile_phase_c(const DevArgs* cp, int t0, int nt, int ntm) {
    BEAN_TILE_ARGS(cp);
    constexpr int NT = kTileThreads;
    extern __shared__ double sm[];
    t0 = __builtin_amdgcn_readfirstlane(t0);
    nt = __builtin_amdgcn_readfirstlane(nt);
    ntm = __builtin_amdgcn_readfirstlane(ntm);
    const int tid = threadIdx.x;
    const int B = c.B, T = c.T, R = c.R;
    const bool prep = true;
    double* hmu = sm;
    double* hy = sm + ntm;
    double* dcol = sm + 3 * B * ntm + R * 4 * B;
    // ---- phase C: the Phi tables of the new draws.  One lane per DISTINCT finite bin edge of a
    // target (DevArgs::ue_z), Phi / phi / u phi through LDS, then one lane per (target, bin) forms the
    // three table entries - phi_edge's formulas on the same operands: the same bits.
    if (prep) {
#pragma clang fp contract(off)
        const int nue = c.ue_idx[2 * B];
        const int nu1 = nue > 0 ? nue : 1, per = NT / nu1;
        double* const ecdf = dcol;  // [per * nue] each
        double* const epdf = dcol + NT;
        double* const eupd = dcol + 2 * NT;
        for (int base = 0; base < nt; base += per) {
            const int grp = tid / nu1, ue = tid - grp * nu1;
            const int tl = base + grp;
            const bool live = grp < per && tl < nt && nue > 0;
            const int tq0 = live ? tl : 0;
            {
                const double mu = hmu[tq0], y = hy[tq0];
                const double sigma = c.family == kNormal ? exp(0.5 * y) : exp(y);
                const double inv = 1.0 / sigma;
                const double u = (c.ue_z[live ? ue : 0] - mu) * inv;
                const double pdf = norm_pdf(u);
                if (live) {
                    ecdf[tid] = norm_cdf(u);
                    epdf[tid] = pdf;
                    eupd[tid] = u * pdf;
                }
            }
            __syncthreads();
            const int cnt = (nt - base < per ? nt - base : per) * B;  // (target, bin) pairs of this pass
            for (int qq = tid; qq < cnt; qq += NT) {
                const int gq = qq / B, b = qq - gq * B, tq = base + gq;
                const double y = hy[tq];
                const double sigma = c.family == kNormal ? exp(0.5 * y) : exp(y);
                const double dsig_dy = c.family == kNormal ? 0.5 * sigma : sigma;
                const double inv = 1.0 / sigma;
                const int ih = c.ue_idx[b], il = c.ue_idx[B + b];
                const double ch = ih < 0 ? 1.0 : ecdf[gq * nu1 + ih], cl = il < 0 ? 0.0 : ecdf[gq * nu1 + il];
                const double fh = ih < 0 ? 0.0 : epdf[gq * nu1 + ih], fl = il < 0 ? 0.0 : epdf[gq * nu1 + il];
                const double uh = ih < 0 ? 0.0 : eupd[gq * nu1 + ih], ul = il < 0 ? 0.0 : eupd[gq * nu1 + il];
                const long o = (long)b * T + t0 + tq;
                __hip_atomic_store(c.tabP + o, ch - cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c.tabPmu + o, -(fh - fl) * inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c.tabPy + o, -(uh - ul) * inv * dsig_dy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
    }
}

// The guide phase of one (tile, step), out of line: compiled on its own this is k_guide_wave2's code (no spills);
// inlined into the step loop the compiler hoists every step-invariant value of guide_pair_math (eps / B,
// (double)mask_thres, the LDS offsets of the bin slots, row pointers, ...) above both loops and spills them.
template <int FAM, bool ACC>
__device__ __noinline__ void tile_guide_step(const DevArgs* cp, int g0, int t0, int nt, int ntm, int gbm,
                                             unsigned long long step, int par) {
    BEAN_TILE_ARGS(cp);
    constexpr bool MIX = FAM == kMixture;
    constexpr int NT = kTileThreads;
    extern __shared__ double sm[];
    g0 = rfl_i(g0);
    t0 = rfl_i(t0);
    nt = rfl_i(nt);
    ntm = rfl_i(ntm);
    gbm = rfl_i(gbm);
    par = rfl_i(par);
    step = rfl_u64(step);
    const int B = c.B, R = c.R;
    const int Bd = B < 3 ? 3 : B;
    double* const tabs = sm;
    double* const cst = tabs + 3 * B * ntm;
    double* const dcol = cst + R * 4 * B;
    double* const mper = dcol + Bd * NT;
    double* const mcnt = mper + 5 * gbm;
    double* const misc = mcnt + 2 * NT;
    int* const linfo = (int*)(misc + 16);
    float* const xs = (float*)(linfo + NT);
    double* const mstep = misc + par * 8;
    const bool use_bc = (c.flags & kUseBc) != 0;
    StepCtr ctr;
    ctr.step = step;
    ctr.slot = 0;
    ctr.step_size = 0.f;
    ctr.pad_ = 0.f;
    const int tid = threadIdx.x, G = c.G, T = c.T;
    // the tile's table columns (written by the previous finish phase, or by k_param's PREP before
    // the first step of the call): 3 B rows of nt entries
    for (int q = tid; q < 3 * B * nt; q += NT) {
        const int wb = q / nt, jj = q - wb * nt;
        const int which = (wb >= B) + (wb >= 2 * B), bb = wb - which * B;
        const double* tab = which == 0 ? c.tabP : (which == 1 ? c.tabPmu : c.tabPy);
        tabs[wb * ntm + jj] = row_ld<true>(tab + (long)bb * T + t0 + jj);
    }
    __syncthreads();  // (also: linfo and the per-tile data of the first step)
    const int info = linfo[tid];
    const int r = info & 0xff, j = (info >> 8) & 0xff, tcol = (info >> 16) & 0xff;
    const bool rgm = (info >> 24) & 1, pair_on = (info >> 25) & 1;
    const int g = g0 + j;
    // the pair's counts: staged every step (guide_pair_math parks digamma halves in the dead slots)
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
            xs[(0 * B + bb) * NT + tid] = xv[0][b];
            xs[(1 * B + bb) * NT + tid] = xv[1][b];
        }
    }
    float api0 = 0.f, api1 = 0.f;
    double pa0 = 0.0;
    if (MIX) {
        api0 = __hip_atomic_load(c.p[4] + 2 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        api1 = __hip_atomic_load(c.p[4] + 2 * g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pa0 = mper[4 * gbm + j];
    }
    const double* c_sf = cst + r * 4 * B;
    double loss = 0.0;
    if (pair_on) {
        // (this kernel keeps d/dmu_t, d/dy_t as per-guide rows: its finish phase adds them in the order k_param
        // used until round 4; k_guide_wave2 now sums per target inside its waves)
        double a_mu = 0.0, a_y = 0.0;
        double pi0 = 0.0, pi1 = 1.0;
        guide_pair_draw<FAM>(c, ctr, r, g, api0, api1, pa0, nullptr, mper + 2 * gbm + j, mper + 3 * gbm + j, pi0, pi1);
        loss = guide_pair_math<FAM, ACC, true, NT>(c, ctr, r, g, rgm, pi0, pi1, tabs + tcol, ntm,
                                                   c_sf, c_sf + 2 * B, c_sf + 3 * B, xs + tid, dcol + tid, mper + j,
                                                   mper + gbm + j, mcnt + tid, mcnt + NT + tid, mper + 2 * gbm + j,
                                                   mper + 3 * gbm + j, a_mu, a_y);
        double* row = c.wrow + (long)r * c.G + g;
        w2_row_store<true>(row + (long)kW2Gmu * c.R * c.G, a_mu);
        w2_row_store<true>(row + (long)kW2Gy * c.R * c.G, a_y);
    }
    const double wl = wave_sum(loss);
    if ((tid & 63) == 0) mstep[tid >> 6] = wl;
}

template <int FAM, bool ACC>
__global__ __launch_bounds__(kTileThreads) __attribute__((amdgpu_waves_per_eu(4)))
void k_svi_tile(DevArgs c, const DevArgs* cp, const int* __restrict__ tile_g0, const int* __restrict__ tile_t0, int n_tiles, int ntm, int gbm,
                unsigned long long step0, unsigned long long slot0, int n_steps, int prep_last,
                const float* __restrict__ step_sizes) {
    constexpr bool MIX = FAM == kMixture;
    constexpr int NT = kTileThreads;
    extern __shared__ double sm[];
    const int B = c.B, R = c.R;
    const int Bd = B < 3 ? 3 : B;
    // LDS: [3][B][ntm] tables | [R][4][B] sf, sf_bc, sample mask, P0 | [Bd][NT] digamma columns |
    //      [5][gbm] a0, a0_bc, c_p0, c_p1, pi_a0 per guide | [2][NT] control allele counts | [16] misc |
    //      [NT] packed (replicate, guide, table column, masks) per thread | [2][B][NT] counts
    double* const tabs = sm;
    double* const cst = tabs + 3 * B * ntm;
    double* const dcol = cst + R * 4 * B;
    double* const mper = dcol + Bd * NT;
    double* const mcnt = mper + 5 * gbm;
    double* const misc = mcnt + 2 * NT;
    int* const linfo = (int*)(misc + 16);
    float* const xs = (float*)(linfo + NT);
    const bool use_bc = (c.flags & kUseBc) != 0;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int g0 = tile_g0[tile], ng = tile_g0[tile + 1] - g0;
        const int t0 = tile_t0[tile], nt = tile_t0[tile + 1] - t0;
        __syncthreads();  // the previous tile's last reads of LDS are done
        // ---- once per tile: thread -> (replicate, guide); per-guide data, control allele counts, constants.
        // Nothing per thread stays in registers across the steps: the guide phase needs them all.
        {
            const int tid = threadIdx.x, G = c.G;
            int r = (int)(((float)tid + 0.5f) / (float)ng);
            const bool pair_on = r < R;
            if (!pair_on) r = R - 1;
            const int j = pair_on ? tid - r * ng : 0;
            const int g = g0 + j;
            double cnt0 = 0.0, cnt1 = 0.0;
            if (MIX)
                for (int cc = 0; cc < c.C; ++cc) {
                    const float* al = c.allele + (((long)r * c.C + cc) * G + g) * 2;
                    cnt0 += (double)al[0];
                    cnt1 += (double)al[1];
                }
            mcnt[tid] = cnt0;
            mcnt[NT + tid] = cnt1;
            const int rgm = c.rg[(long)r * G + g] != 0 ? 1 : 0;
            const int tcol = c.g2t[g] - t0;
            linfo[tid] = r | (j << 8) | (tcol << 16) | (rgm << 24) | ((pair_on ? 1 : 0) << 25);
            if (tid < ng) {
                mper[tid] = c.a0[g0 + tid];
                mper[gbm + tid] = use_bc ? c.a0_bc[g0 + tid] : 0.0;
                mper[4 * gbm + tid] = MIX ? c.pi_a0[g0 + tid] : 0.0;
            }
            for (int q = tid; q < R * 4 * B; q += NT) {
                const int rr = q / (4 * B), k = (q - rr * 4 * B) / B, b = q - rr * 4 * B - k * B;
                const double* src = k == 0 ? c.sf + rr * B : (k == 1 ? (use_bc ? c.sf_bc : c.sf) + rr * B
                                                                     : (k == 2 ? c.smask + rr * B : c.P0));
                cst[q] = (MIX || k != 3) ? src[b] : 0.0;
            }
        }

        for (int s = 0; s < n_steps; ++s) {
            StepCtr ctr;
            ctr.step = step0 + (unsigned long long)s;
            ctr.slot = slot0 + (unsigned long long)s;
            ctr.step_size = step_sizes[s];  // ClippedAdam step size of update t = step + 1 (k_step_sizes)
            ctr.pad_ = 0.f;
            double* const mstep = misc + (s & 1) * 8;  // this step's loss parts (double-buffered: thread 0 reads late)
            // ================================ guide phase ================================
            tile_guide_step<FAM, ACC>(cp, g0, t0, nt, ntm, gbm, ctr.step, s & 1);
            // every row store of this workgroup has completed before any of its threads reads one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            // ================================ finish phase ================================
            // (out of line, reading DevArgs from its copy in global memory: each piece gets a register
            // allocation of its own instead of competing with guide_pair_math's ~110 VGPRs)
            const bool prep = prep_last || s + 1 < n_steps;
            double loss_fin = tile_finish_targets(cp, t0, nt, ntm, ctr.step + 1, ctr.step_size, prep ? 1 : 0);
            if (MIX) loss_fin += tile_finish_guides(cp, g0, ng, ctr.step + 1, ctr.step_size, prep ? 1 : 0);
            __syncthreads();  // hmu / hy complete
            if (prep) tile_phase_c(cp, t0, nt, ntm);
            // ---- loss of this (tile, step): the waves' likelihood parts + the prior / entropy terms, integer atomics
            {
                const int tid = threadIdx.x;
                const double lf = wave_sum(loss_fin);
                if ((tid & 63) == 0) mstep[4 + (tid >> 6)] = lf;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // table / parameter stores before the next step reads them
                __syncthreads();
                if (tid == 0) {
                    double tot = 0.0;
                    for (int i = 0; i < NT / 64; ++i) tot += mstep[i];
                    for (int i = 0; i < NT / 64; ++i) tot += mstep[4 + i];
                    fixed_add(c.loss_acc + ((long)ctr.slot * kLossSub + (tile & (kLossSub - 1))) * kLossWords, tot);
                }
            }
        }
    }
}

// (k_put_args, k_step_sizes: bean_async_v2.hpp)

}  // namespace bean

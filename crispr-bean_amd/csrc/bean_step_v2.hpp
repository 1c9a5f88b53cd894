// One launch per SVI step for the variant sorting families (OPT-IN: BEAN_HIP_STEP=fused): k_step_wave2 =
// k_guide_wave2's wave work, then - in the wave that finishes a tile last - everything k_param does
// for that tile's targets and guides (FINISH of this step, PREP of the next).
//
// Why it was built: a kernel boundary costs 3-5 us on this part (eight XCDs, eight L2s) and k_param's
// ~6 us of in-wave time is memory round trips and one serial chain; at the metric shape the
// {k_param, guide} pair spends ~13 of its 59 us there.
//
// Why it is not the default: measured (same box, 50k guides x 5 replicates) 65.5 us per step against
// 59.6 us for the two launches (500k guides: 441 vs 400).  All tiles finish together, so the 782
// finishing waves run alone at the end of the launch, each a serial chain of ~25 k cycles (row sums ->
// priors / ClippedAdam / draw -> Phi tables -> the guides' alpha_pi and lgamma tables -> loss parts;
// scripts/stamps_tail.py), i.e. 15 us where k_param's 3 284 waves need 6.  What a launch of its own
// buys is width, and that is worth more than the boundary it costs.  Kept because it is bit-identical
// to the two-launch path (tests/test_gpu_step_fused.py), which pins the shared per-target / per-guide
// code and the arrival mechanism below - the building block a persistent, tile-asynchronous step would need.
//
// How a tile is finished without a fence: the R waves of a tile store their rows, per-target sums and their loss
// part with agent-scope stores (write-through, `sc1`), wait for those stores to complete
// (s_waitcnt vmcnt(0)) and then count themselves in with one relaxed agent-scope atomic; the wave that
// counts R - 1 predecessors reads the rows with agent-scope loads.  (An agent-scope FENCE per wave -
// __threadfence(): L2 write-back + invalidate - was measured at 46 -> 189 us per launch; the scoped
// accesses cost nothing measurable.)  No wave ever waits for another: there is no spin, nothing to
// deadlock, and the counters are back at zero when the launch ends.
//
// Targets are not aligned to the 64-guide tiles.  A target that straddles the boundary between tiles
// i and i + 1 is finished by whichever of the two tiles completes second (a second counter per
// boundary); the path is taken only when no target is longer than 64 guides, so a target touches at
// most two tiles.  The per-target and per-guide arithmetic is the same code k_param runs
// (tgt_prior_terms, tgt_grad, tgt_draw, phi_edge's formulas, param_guide_mix: roundings pinned), the
// (guide, replicate) sums reproduce k_param's 16-lane order: parameters are bit-identical.
#pragma once

namespace bean {

template <int FAM, bool ACC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BEAN_WAVE_EU)))
void k_step_wave2(DevArgs c, int flip) {
    constexpr bool MIX = FAM == kMixture;
    extern __shared__ double tabs[];
    // step counters ping-pong between launches: no wave of this launch reads what another one writes
    const StepCtr* in = flip ? c.ctrB : c.ctrA;
    StepCtr* out = flip ? c.ctrA : c.ctrB;
    const StepCtr ctr = *in;
    int tile, r, t0, nt;
    double tot;
    if (!guide_wave2_body<FAM, ACC, true>(c, ctr, tile, r, t0, nt, tot)) return;
    const int lane = threadIdx.x, wg = blockIdx.x, R = c.R, G = c.G, B = c.B;
    // where the tile's first and last targets begin / end (data; issued before the arrival, used after it)
    const int t1 = t0 + nt - 1;
    const int tof0 = c.toff[t0], tof1 = c.toff[t1 + 1];
    if (lane == 0) {
        // this wave's part of the loss as the three integer words of fixed_add (see wave_loss_out)
        long long w0 = 0, w1 = 0, w2 = 1;
        if (fabs(tot) < kLossPartMax) {
            const double hi = rint(tot * 1024.0);
            w0 = (long long)hi;
            w1 = (long long)rint((tot - hi * (1.0 / 1024.0)) * 1099511627776.0);
            w2 = 0;
        }
        long long* o = c.lpart + 3 * (long)wg;
        __hip_atomic_store(o, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wg == 0) {
            // for the next launch: its step, its loss slot, the ClippedAdam step size of ITS update
            StepCtr nxt;
            nxt.step = ctr.step + 1;
            nxt.slot = ctr.slot + 1;
            nxt.step_size = adam_coef(c, ctr.step + 2).step_size;
            nxt.pad_ = 0.f;
            *out = nxt;
        }
    }
    // ---- arrival: every row / loss-part store of this wave has completed before it is counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) {
        const int old = __hip_atomic_fetch_add(c.tile_ctr + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == R - 1) {
            last = 1;
            __hip_atomic_store(c.tile_ctr + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;

    // =========================== the tile is complete: finish it ===========================
#if defined(BEAN_STAMP) && BEAN_STAMP == 3
    const long wave_gid = tile;
#endif
    BEAN_STAMP_TL(0);
    const unsigned long long s_prep = ctr.step + 1;
    AdamCoef ak;
    ak.step_size = ctr.step_size;  // of update t = step + 1 (k_set_step / the previous launch)
    ak.clip = (float)c.clip;
    const int g_first = tile * 64 - c.g_sh > 0 ? tile * 64 - c.g_sh : 0;  // (tiles follow the global guide index)
    const int g_last = tile * 64 + 63 - c.g_sh < G ? tile * 64 + 63 - c.g_sh : G - 1;
    const bool left_str = tof0 < g_first;
    const bool right_str = tof1 > g_last + 1;
    int own_left = left_str ? 0 : 1, own_right = right_str ? 0 : 1;
    if (lane == 0) {
        // both boundary counters in flight before either result is looked at
        int* const bl = c.bnd_ctr + (left_str ? tile - 1 : tile);
        int* const br = c.bnd_ctr + tile;
        int ol = 0, orr = 0;
        if (left_str) ol = __hip_atomic_fetch_add(bl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (right_str) orr = __hip_atomic_fetch_add(br, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left_str && ol == 1) {
            own_left = 1;
            __hip_atomic_store(bl, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (right_str && orr == 1) {
            own_right = 1;
            __hip_atomic_store(br, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    own_left = __builtin_amdgcn_readfirstlane(own_left);
    own_right = __builtin_amdgcn_readfirstlane(own_right);
    const int ta = own_left ? t0 : t0 + 1, tb = own_right ? t1 : t1 - 1;  // this wave's targets, ta > tb: none
    double loss_fin = 0.0;
    double* hmu = tabs;        // drawn mu / y of the targets, hmu[t - ta] (<= 64 targets per tile)
    double* hy = tabs + 64;
    __syncthreads();  // single-wave workgroup: the body's LDS is free from here
    BEAN_STAMP_TL(1);

    // ---- phases A + B: four lanes per target (lane j: unconstrained parameter j), 16 targets per pass
    {
        const int j = lane & 3;
        float* const P = j == 0 ? c.p[0] : (j == 1 ? c.p[1] : (j == 2 ? c.p[2] : c.p[3]));
        float* const M = j == 0 ? c.m[0] : (j == 1 ? c.m[1] : (j == 2 ? c.m[2] : c.m[3]));
        float* const V = j == 0 ? c.v[0] : (j == 1 ? c.v[1] : (j == 2 ? c.v[2] : c.v[3]));
        for (int base = ta; base <= tb; base += 16) {
            const int t = base + (lane >> 2);
            const bool act = t <= tb;
            const int tc = act ? t : tb;
            // everything that does not depend on the sums first: one round trip
            int2 dsc;
            dsc.x = 0;
            dsc.y = 2;
            if (!c.tsum_direct) dsc = c.tdesc[tc];
            const int n = dsc.y * R, ntm = c.tile_targets;
            const long S = c.tsum_direct ? 2 * (long)c.T : (long)c.n_tiles * ntm;
            float pj = P[tc], mj = M[tc], vj = V[tc];
            const float p1 = c.p[1][tc], p3 = c.p[3][tc];
            const double eps1 = c.eps_mu[tc], eps2 = c.eps_sd[tc], mu = c.mu_t[tc], y = c.y_t[tc];
            // the (part, replicate) sums of the target in k_param's order: its 16 lanes take entries
            // lg, lg + 16, ... and combine by an xor tree (8, 4, 2, 1); lane j here plays lanes j + 4 k
            double am[4] = {0.0, 0.0, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.0, 0.0};
            const float rR = 1.0f / (float)R;  // i / R below: exact for i < 2^20
            for (int i0 = 0; i0 < n; i0 += 32) {
                // 16 loads in flight per lane (an atomic load is waited for where it is used, and the
                // compiler keeps atomic loads in program order: load first, add afterwards)
                double xm[2][4], xy[2][4];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + 16 * mm + j + 4 * k;
                        xm[mm][k] = 0.0;
                        xy[mm][k] = 0.0;
                        if (act && i < n) {
                            const int part = (int)(((float)i + 0.5f) * rR), rr = i - part * R;
                            const long o = (long)rr * S + (c.tsum_direct ? 2 * (long)tc + part
                                                                         : (part == 0 ? dsc.x : (dsc.x / ntm + part) * ntm));
                            xm[mm][k] = row_ld<true>(c.tsum + o);
                            xy[mm][k] = row_ld<true>(c.tsum + (long)R * S + o);
                        }
                    }
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = i0 + 16 * mm + j + 4 * k;
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
            if (act && j == 0) loss_fin += lt;
            const double Gd = j < 2 ? gmu - dlogp_mu : gy - dlogp_dy;
            const double grad = tgt_grad(j, Gd, j < 2 ? eps1 : eps2, exp((double)pj));
            adam_update(pj, mj, vj, (float)grad, ak);
            if (act) {
                P[t] = pj;
                M[t] = mj;
                V[t] = vj;
            }
            // PREP of the next step: the draw (Philox keyed by the global target index and the step)
            const float2 nrm = normal2_at(c.seed, ((unsigned long long)kSiteTarget << 48) + (unsigned long long)(c.t_off + tc),
                                          s_prep * 4ull);
            const double en = j < 2 ? (double)nrm.x : (double)nrm.y;
            const float p_scale = __shfl_xor(pj, 1, 64);  // even lanes: the updated log scale of their pair
            const double val = tgt_draw(pj, en, p_scale);
            if (act && (j & 1) == 0) {
                (j == 0 ? c.eps_mu : c.eps_sd)[t] = en;
                (j == 0 ? c.mu_t : c.y_t)[t] = val;
                (j == 0 ? hmu : hy)[t - ta] = val;
            }
        }
    }
    __syncthreads();
    BEAN_STAMP_TL(2);
    // ---- phase C: the Phi tables of the new draws.  One lane per DISTINCT finite bin edge of a target
    // (DevArgs::ue_z: 4 for the standard bins, so 16 targets per pass instead of 6), Phi / phi / u phi
    // through LDS, then one lane per (target, bin) forms the three table entries - phi_edge's formulas
    // on the same operands: the same bits.
    {
#pragma clang fp contract(off)
        const int nue = c.ue_idx[2 * B];
        const int nu1 = nue > 0 ? nue : 1, per = 64 / nu1;
        double* const ecdf = tabs + 128;  // [per * nue] each
        double* const epdf = ecdf + 64;
        double* const eupd = epdf + 64;
        for (int base = ta; base <= tb; base += per) {
            const int grp = lane / nu1, ue = lane - grp * nu1;
            const int t = base + grp;
            const bool live = grp < per && t <= tb && nue > 0;
            const int tl = live ? t - ta : 0;
            {
                const double mu = hmu[tl], y = hy[tl];
                const double sigma = c.family == kNormal ? exp(0.5 * y) : exp(y);
                const double inv = 1.0 / sigma;
                const double u = (c.ue_z[live ? ue : 0] - mu) * inv;
                const double pdf = norm_pdf(u);
                if (live) {
                    ecdf[lane] = norm_cdf(u);
                    epdf[lane] = pdf;
                    eupd[lane] = u * pdf;
                }
            }
            __syncthreads();
            const int cnt = (tb - base + 1 < per ? tb - base + 1 : per) * B;  // (target, bin) pairs of this pass
            for (int q = lane; q < cnt; q += 64) {
                const int gq = q / B, b = q - gq * B, tq = base + gq;
                const double y = hy[tq - ta];
                const double sigma = c.family == kNormal ? exp(0.5 * y) : exp(y);
                const double dsig_dy = c.family == kNormal ? 0.5 * sigma : sigma;
                const double inv = 1.0 / sigma;
                const int ih = c.ue_idx[b], il = c.ue_idx[B + b];
                const double ch = ih < 0 ? 1.0 : ecdf[gq * nu1 + ih], cl = il < 0 ? 0.0 : ecdf[gq * nu1 + il];
                const double fh = ih < 0 ? 0.0 : epdf[gq * nu1 + ih], fl = il < 0 ? 0.0 : epdf[gq * nu1 + il];
                const double uh = ih < 0 ? 0.0 : eupd[gq * nu1 + ih], ul = il < 0 ? 0.0 : eupd[gq * nu1 + il];
                const long o = (long)b * c.T + tq;
                c.tabP[o] = ch - cl;
                c.tabPmu[o] = -(fh - fl) * inv;
                c.tabPy[o] = -(uh - ul) * inv * dsig_dy;
            }
            __syncthreads();
        }
    }
    BEAN_STAMP_TL(3);
    // ---- the tile's guides: alpha_pi (and the accessibility noise site), tables for the next launch
    if (MIX) {
        const int g = tile * 64 + lane - c.g_sh;
        double lg = 0.0;  // (param_guide_mix assigns its loss terms)
        if (g >= 0 && g < G) param_guide_mix<true, true, true, true>(c, g, ak, s_prep, lg);
        loss_fin += lg;
    }
    BEAN_STAMP_TL(4);
    // ---- loss: the R waves' parts of this tile + this wave's prior / entropy terms, integer atomics
    const double lsum = wave_sum(loss_fin);
    if (lane == 0) {
        // the words of the tile's R waves: eight waves' loads in flight, then integer adds
        long long a = 0, b = 0, d = 0;
        for (int r0 = 0; r0 < R; r0 += 8) {
            long long w[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                w[u][0] = w[u][1] = w[u][2] = 0;
                if (r0 + u < R) {
                    const long long* o = c.lpart + 3 * (long)(((tile >> 3) * R + r0 + u) * 8 + (tile & 7));
                    w[u][0] = __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w[u][1] = __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    w[u][2] = __hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a += w[u][0];
                b += w[u][1];
                d += w[u][2];
            }
        }
        if (fabs(lsum) < kLossPartMax) {
            const double hi = rint(lsum * 1024.0);
            a += (long long)hi;
            b += (long long)rint((lsum - hi * (1.0 / 1024.0)) * 1099511627776.0);
        } else {
            d += 1;
        }
        long long* acc = c.loss_acc + ((long)ctr.slot * kLossSub + (tile & (kLossSub - 1))) * kLossWords;
        atomicAdd((unsigned long long*)acc, (unsigned long long)a);
        atomicAdd((unsigned long long*)acc + 1, (unsigned long long)b);
        if (d) atomicAdd((unsigned long long*)acc + 2, (unsigned long long)d);
    }
    BEAN_STAMP_TL(5);
}

}  // namespace bean

// k_guide_tiling_rep: the per-(replicate, guide) kernel of the tiling families (MultiMixtureNormal,
// up to kAMax alleles per guide), third form: THE REPLICATES OF A GUIDE SHARE A WORKGROUP.
//
// Same per-lane arithmetic as k_guide_tiling_wave (bean_kernels.hpp; reference semantics:
// bean/model/model.py:550-751 model, 878-962 guide; survival_model.py:427-626, 738-833); what changed is
// the thread map and, with it, what leaves the kernel:
//   * thread = r * Gb + j: a workgroup of NT = 64 W threads holds Gb = NT / R consecutive guides times all
//     R replicates (W = 4: 51 guides x 5 replicates = 255 of 256 lanes at the shape of BASELINE config 3;
//     the host picks the W in {1, 2, 4} that fills most lanes).  Everything k_param needs
//     from this kernel is a sum over the replicates of a guide, so each row is reduced IN THE WORKGROUP
//     (through a dead LDS column, fixed order r = 0, 1, ... - the order k_sum_trow used, same bits) and the
//     thread of replicate 0 stores one value to part[(q, g)].  The wave form wrote 32 rows per (replicate,
//     guide) (79 MB per launch at config 3, the largest item of its traffic) for a second launch,
//     k_sum_trow, to read back and add: both are gone.
//   * the R threads of a guide read the same allele-table entries, per-guide values and parameters: one
//     request per workgroup instead of one per replicate's wave (the wave form relied on an XCD-aware
//     block order to get those re-reads out of L2 at least).
//   * register diet (the wave form: 128 VGPRs + 10 spilled, 64 B of scratch per lane): the guide's
//     concentrations (alpha_a, c_q a: 32 VGPRs) are needed by the sampler at the start and by the
//     Dirichlet terms at the end, and are RECOMPUTED there from alpha_pi instead of held across the
//     likelihoods; a row is reduced and stored where it is formed instead of collected in an array.
//   * (forming the allele tables - k_allele, 17 us - at the head of this kernel was measured too: the step
//     takes the same 183 us, the kernel 160 instead of 148 + 17, and its traffic then includes the 42 MB
//     table round trip; k_allele stays a launch of its own.)
//   * why not simply one wave = 12 guides x 5 replicates: 60 of 64 lanes, i.e. 4 167 waves for 50k guides
//     against the chip's 4 096 wave slots (256 CUs x 16): a second round of 71 waves, measured + 50 %
//     (205 us against 137).  The launch must stay inside one round at this size, so lanes cannot idle.
// All threads run the whole body (threads past the last guide / past R * Gb on clamped indices, threads
// whose (replicate, guide) is masked by repguide_mask on their real data): the row reductions have
// workgroup barriers.
#pragma once

// Diagnostic builds only (wrong results): -DBEAN_TL_DIAG=n removes one piece of k_guide_tiling_rep so that A/B timings
// (scripts/micro/tiling_sorted.py with BEAN_HIP_LIB / BEAN_HIP_LIB_A16) give that piece's cost in place: 1 the rejection
// loops of the draw, 2 the implicit-gradient calls, 3 the lgamma / digamma differences.
#ifndef BEAN_TL_DIAG
#define BEAN_TL_DIAG 0
#endif
namespace bean {

// Dynamic LDS of the register-resident tiling guide kernels, nt threads per workgroup: three columns of B
// doubles per thread; k_guide_tiling_wave (the per-replicate wave form) also parks the accessibility pieces of
// every allele (3 kAMax columns) and stages the counts.  k_guide_tiling_rep parks NOTHING for accessibility
// (round 5): with those columns a 256-thread workgroup asked for 79 872 B at config 3, two workgroups per CU,
// 512 places for 981 workgroups - the launch ran as two rounds, 156.8 us against 110.4 without accessibility.
__host__ __device__ inline size_t guide_tiling_lds(int B, bool acc, size_t nt, bool wave_form) {
    return (size_t)(3 * B + (acc && wave_form ? 3 * kAMax : 0)) * nt * sizeof(double) +
           (wave_form ? (size_t)2 * B * nt * sizeof(float) : 0);
}

// The accessibility transform of one edited allele's share pi_a (utils.py:106-178; kacc = exp(b) acc^a, formed once
// by k_acc_scale; lpn the guide's logit-noise draw): scaled share s1, clamp, logit + noise, sigmoid, clamp.
//   pea = the share the mixture sees; dpi = d pea / d pi_a; dl = d pea / d noise.
// Cheap enough (one log, one exp, three reciprocals) to be formed where it is used, forward AND backward, instead of
// parked across the likelihoods: same expressions, same values.
struct AccShare {
    double pea, dpi, dl;
};
__device__ __forceinline__ AccShare acc_share(double pia, double kacc, double lpn) {
    const double s1 = pia * kacc;
    const bool in1 = s1 > 1e-3 && s1 < 1.0 - 1e-3;
    const double p1c = fmin(fmax(s1, 1e-3), 1.0 - 1e-3);
    const double l = flog(p1c * frcp(1.0 - p1c)) + lpn;
    const double el = exp(l);
    const double pn = el * frcp(1.0 + el);
    const bool in2 = pn > 1e-3 && pn < 1.0 - 1e-3;
    AccShare o;
    o.pea = fmin(fmax(pn, 1e-3), 1.0 - 1e-3);
    o.dl = in2 ? pn * (1.0 - pn) : 0.0;
    o.dpi = in1 ? o.dl * frcp(p1c * (1.0 - p1c)) * kacc : 0.0;
    return o;
}


constexpr int kTilingRepMaxR = 64;  // more replicates than that: the wave form + k_sum_trow
// waves per workgroup of k_guide_tiling_rep: of 1, 2, 4 the one that keeps most lanes busy - the share of
// lanes holding a (replicate, guide) pair (R = 5: 60 / 64, 125 / 128, 255 / 256) times the share of the CU's 16
// wave slots its LDS lets the kernel use (160 KB per CU; with many conditions or the accessibility columns a
// 256-thread workgroup no longer fits four times); the smaller on a tie.  Measured at config 3 (B = 5): W = 2
// 188.6 us per step, W = 4 183.1.
// A screen (or one rank's shard of it) small enough to give every SIMD of the device at most one single-wave workgroup
// takes W = 1: such a launch is as long as one wave's dependency chain, and a wave alone in its workgroup waits for no
// other wave at the row barriers (BASELINE config 3 cut in eight, 6 250 guides: 85.5 us per step at W = 1, 87.2 at 2, 91.5
// at 4; at 12 500 guides - 1 042 single-wave workgroups for 1 024 SIMDs - 92.6 / 94.5 / 93.5: the rule below again).
__host__ inline int tiling_rep_waves(int R, int B, bool acc, long G = 0, long n_simd = 0) {
    if (G > 0 && n_simd > 0 && R <= 64) {
        const long gw1 = 64 / R;  // guides per single-wave workgroup
        if ((G + gw1 - 1) / gw1 <= n_simd) return 1;
    }
    int best = 1;
    double best_score = 0.0;
    for (int w = 1; w <= 4; w *= 2) {
        const double eff = (double)((64 * w / R) * R) / (double)(64 * w);
        const size_t lds = guide_tiling_lds(B, acc, (size_t)64 * w, false);
        // (a request of EXACTLY a quarter / an eighth ... of the CU's 160 KB does not reliably fit that many times -
        // round 3's 40 960 B workgroups ran as one or as two rounds from run to run - so 2 KB per CU are left out)
        const size_t wgs = lds ? (size_t)158 * 1024 / lds : 16;
        const double waves = (double)(wgs * w < 16 ? wgs * w : 16) / 16.0;
        if (eff * waves > best_score + 1e-12) {
            best_score = eff * waves;
            best = w;
        }
    }
    return best;
}

// (Which guides a workgroup takes - round 5, measured: workgroup b takes guides [b Gw, (b + 1) Gw), and with the guides
// ordered by allele count, heaviest first, that is already the balanced assignment: the XCD's workgroups go round its
// 32 CUs, so the four workgroups a CU holds, b, b + 256, b + 512, b + 768, come from the four quarters of the order.
// Mapped so that a CU's four come from ONE quarter the launch takes 124 us instead of 115; a mapping that is balanced
// under either dispatch order 117.  And the launch is as long as its average wave, not its heaviest: screens whose
// guides all have 1 / ~2.5 / ~4 / 7 edited alleles take 98 / 104 / 113 / 131 us - a masked allele slot costs three
// quarters of a filled one, and neither the floor sampler nor an all-floor shortcut of the implicit gradient on the slots
// above a wave's alleles takes anything off it: 7 % of a masked allele's draws stay above the floor.)
// The rows k_guide_tiling_rep hands to k_param are sums over the replicates of a guide.  They are collected in batches:
// a thread parks its value of row n of the batch in column n of a buffer of B dead LDS columns (tiling_rows_flush's
// caller), and once a buffer is full - or the kernel ends - ONE barrier closes the batch and the thread of (replicate r,
// guide j) adds up row r (and r + R, ...) of guide j over the replicates, r = 0 first (the order k_sum_trow used: same
// bits), and stores it.  Round 3's form closed every row with a barrier of its own and let the R - 1 other replicates'
// threads idle while replicate 0's added: 32 barriers and 32 five-term sums per thread of replicate 0 at config 3, now 7
// barriers and <= 7 sums per thread (~750 of a wave's 11 600 instructions).  Two buffers (the e[] columns and the
// digamma-difference columns, both dead once the likelihoods are done) alternate, so a batch is written while the
// previous one may still be read: the barrier of batch n + 1 lies between the reads of batch n and the writes of n + 2.
// qpack: the batch's row numbers, eight bits each.  Measured at config 3: 146.9 -> 145.0 us per step (kernel 116.0 -> 114.7);
// as a function out of line - one call per batch - 148.1 (kernel 118.3).  Far less than 750 of 11 600 instructions would
// give an issue-bound launch: by its counters the kernel keeps the vector pipe busy for 65 % of its cycles
// (SQ_ACTIVE_INST_VALU x 4 / 1 024 SIMDs against GRBM_GUI_ACTIVE / 8; profiles/r05_counters_tiling.txt).
// Which slice of Gw consecutive guides workgroup b of a grid of n takes.  mode 0: b.  mode 1 (n a multiple of 32): XCD
// b & 7 takes, of each QUARTER of the guide order, one contiguous run of n / 32 slices - its j-th workgroup (j = b >> 3)
// the slice j % (n / 32) of quarter j / (n / 32) - so that workgroups with neighbouring guides share an L2 (a workgroup's
// 51 guides x 8 bytes do not end on a cache line) while a CU's four workgroups (j, j + 32, j + 64, j + 96 for n = 1 024)
// still come from the four quarters of the allele-count order.
__device__ __forceinline__ long tiling_rep_slice(int b, int n, int mode) {
    if (mode == 0) return b;
    const int x = b & 7, j = b >> 3, run = n >> 5;
    const int quarter = j / run, i = j - quarter * run;
    return (long)quarter * (n >> 2) + (long)x * run + i;
}

// Diagnostic builds (-DBEAN_STAMP=6): every wave of k_guide_tiling_rep leaves eight stamps of the 100 MHz real-time clock
// (scripts/stamps_tiling_rep.py): 0 start, 1 first loads + concentrations, 2 draw, 3 forward mix, 4 likelihoods,
// 5 backward loop + rows, 6 Multinomial + Dirichlet terms + implicit gradients, 7 end.
#if defined(BEAN_STAMP) && BEAN_STAMP == 6
#define BEAN_STAMP_TR(slot)                                                                      \
    do {                                                                                         \
        unsigned long long u_;                                                                   \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u_)::"memory");          \
        if ((lane & 63) == 0) c.dbg[((long)blockIdx.x * (NT >> 6) + (lane >> 6)) * 8 + (slot)] = u_; \
    } while (0)
#else
#define BEAN_STAMP_TR(slot) do {} while (0)
#endif

// s_setprio with a run-time level (the instruction takes an immediate)
__device__ __forceinline__ void wave_prio(int level) {
    if (level <= 0) __builtin_amdgcn_s_setprio(0);
    else if (level == 1) __builtin_amdgcn_s_setprio(1);
    else if (level == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

constexpr int kTilingRowBatchMax = 8;
__device__ __forceinline__ void tiling_rows_flush(const double* buf, int NT, int n_pend, unsigned long long qpack, int R,
                                               int Gw, int r, int j, bool ok, double* part, long G, int g) {
    __syncthreads();
    if (ok) {
        for (int k = r; k < n_pend; k += R) {
            const double* col = buf + (long)k * NT;
            double s = col[j];
            for (int rr = 1; rr < R; ++rr) s += col[rr * Gw + j];
            part[(long)((qpack >> (8 * k)) & 255ull) * G + g] = s;
        }
    }
}

template <bool ACC, bool SURV>
// (the 32-allele build: two waves per SIMD and 251 VGPRs instead of four and 128 + ~300 spilled - 274 -> 200 us at
// 20 000 guides with 24 slots, 464 -> 445 at 50 000 with 32; three waves per SIMD are slower than either.  The
// 16-allele build stays at four: without its 60 spills it is 3 % faster at 20 000 guides and 32 % slower at 50 000.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BEAN_AMAX > 16 ? 2 : 4)))
void k_guide_tiling_rep(DevArgs c, int Gw_mode) {
    const int Gw = Gw_mode & 255;
    const long wg_slice = tiling_rep_slice((int)blockIdx.x, (int)gridDim.x, (Gw_mode >> 8) & 255);
    // Issue priority by progress (bit 16 of the argument; BEAN_HIP_TILING_PRIO=0 switches it off): the draw runs at
    // priority 3, forward mix + likelihoods at 2, the backward loop at 1, the Dirichlet terms and implicit gradients at 0.
    // Among equals the arbiter takes the OLDEST wave, and the workgroups are dispatched heaviest first: without this the
    // oldest quarter of the grid has finished at 72 - 78 us of a 103 us launch while the youngest - the lightest
    // guides - waits through its draw (36 us against 12) and runs its last phases alone; with it the quarters end at
    // 78 - 82 / 83 - 86 / 89 - 94 / 96 - 100 us (scripts/stamps_tiling_rep.py, profiles/r05_timeline_tiling_rep.txt).
    // Config 3: 143.5 -> 142.2 us per step.  (Measured too: 2,1,1,0: the same; 1,1,0,0 and the reverse order 0,1,2,3:
    // nothing; one or two levels more for the youngest quarter / half of the grid on top: nothing.)
    const bool prio_on = ((Gw_mode >> 16) & 1) != 0;
    auto phase_prio = [&](int ph) {
        if (prio_on) wave_prio(3 - ph);
    };
    phase_prio(0);
    if (wg_slice * Gw >= c.G) return;  // (a padded grid's workgroups without guides; workgroup 0 always has guides)
    extern __shared__ double tls[];
    const int lane = threadIdx.x;  // thread of the workgroup: columns in LDS have NT entries
    const int NT = blockDim.x;
    const int G = c.G, A = c.A, A1 = c.A - 1, B = c.B, R = c.R;
    // thread -> (replicate, guide of the workgroup); threads >= R * Gw are spare
    int r = (int)(((float)lane + 0.5f) / (float)Gw);
    const bool spare = r >= R;
    if (spare) r = R - 1;
    const int j = lane - r * Gw;
    const long g_raw = wg_slice * Gw + j;
    const bool valid = !spare && g_raw < G;
    const int g = g_raw < G ? (int)g_raw : G - 1;
    const StepCtr ctr = *c.ctrB;
    double* es = tls + lane;                 // e[b]              at es[b * NT]
    double* gs = tls + B * NT + lane;        // d nll / d e[b]    at gs[b * NT]
    double* ds = tls + 2 * B * NT + lane;    // digamma diffs     at ds[b * NT]

    BEAN_STAMP_TR(0);
    const bool rgm = c.rg[(long)r * G + g] != 0;
    // both pi sites, the Multinomial and the count likelihoods are masked by repguide_mask in tiling
    // (model.py:659,682,731; guide 941): a masked (replicate, guide) contributes nothing
    const bool on = valid && rgm;
    const bool use_bc = (c.flags & kUseBc) != 0;
    // row q of this guide = sum over its replicates, r = 0 first: collected in batches of up to B rows (tiling_rows_flush)
    const int row_batch = B < kTilingRowBatchMax ? B : kTilingRowBatchMax;
    int n_pend = 0, n_batches = 0;
    unsigned long long qpack = 0;
    auto rows_flush = [&]() {
        if (n_pend == 0) return;
        tiling_rows_flush(tls + ((n_batches & 1) ? 2 * B * NT : 0), NT, n_pend, qpack, R, Gw, r, j, valid, c.part, G, g);
        n_pend = 0;
        qpack = 0;
        ++n_batches;
    };
    auto row_out = [&](int q, double v) {
        double* buf = tls + ((n_batches & 1) ? 2 * B * NT : 0);
        buf[n_pend * NT + lane] = on ? v : 0.0;
        qpack |= (unsigned long long)q << (8 * n_pend);
        if (++n_pend == row_batch) rows_flush();
    };

    // totals of the guide's counts over the conditions, both likelihoods: one batch of loads.  (The
    // likelihood loop reads the counts again from global memory, one coalesced row per condition: staged
    // in LDS they made the workgroup's 40 KB a quarter of the CU's LDS EXACTLY, and whenever the CU could
    // not give all of it the grid ran as two rounds - steps of 181 or 280 us from one run to the next.)
    double n_x = 0.0, n_bc = 0.0;
    {
        float xv[2][kBMax];
        // (no branch between the loads - X_bcmatch is read through X's pointer where it is not used, and dropped - so that
        // all 2 kBMax are in flight together: with the branch the compiler waited for every pair before it asked for the
        // next, eight round trips in a row at the head of every wave)
        // (the default build; the 16-condition builds keep the loads under the branch - twice the registers in flight)
#if BEAN_BMAX <= 8
        const float* const xbc = use_bc ? c.Xbc : c.X;
#pragma unroll
        for (int b = 0; b < kBMax; ++b) {
            const long xo = ((long)r * B + (b < B ? b : B - 1)) * G + g;
            xv[0][b] = c.X[xo];
            xv[1][b] = xbc[xo];
        }
        if (!use_bc) {
#pragma unroll
            for (int b = 0; b < kBMax; ++b) xv[1][b] = 0.f;
        }
#else
#pragma unroll
        for (int b = 0; b < kBMax; ++b) {
            const long xo = ((long)r * B + (b < B ? b : B - 1)) * G + g;
            xv[0][b] = c.X[xo];
            xv[1][b] = use_bc ? c.Xbc[xo] : 0.f;
        }
#endif
#pragma unroll
        for (int b = 0; b < kBMax; ++b) {
            if (b < B) {
                n_x += (double)xv[0][b];
                n_bc += (double)xv[1][b];
            }
        }
        for (int b = kBMax; b < B; ++b) {  // more conditions than the register batch holds (B <= kBCap)
            const long xo = ((long)r * B + b) * G + g;
            n_x += (double)c.X[xo];
            n_bc += use_bc ? (double)c.Xbc[xo] : 0.0;
        }
    }
    BEAN_STAMP_TR(1);
    const double pa0 = c.pi_a0[g];
    // ---- draw: the concentrations of the guide's Dirichlet live only until the draw is done
    double pi[kAMax];
    // Aw: one more than the highest allele slot that any guide of this WAVE fills.  The tables of an empty slot are
    // zero (k_allele leaves them), so the two loops over (slot, condition) below stop at Aw instead of A - with
    // the guides ordered by allele count (parallel.order_by_alleles) most waves stop early: same sums (the terms
    // left out are exact zeros), fewer loads of zeros
    int Aw = 1;
    {
        double alpha[kAMax], Ssum = 0.0;
        // (mask bytes and parameters of all slots are asked for together, slots beyond A on the last one's address: a load
        // under `am ? ... :` was a round trip of its own behind the mask byte's - sixteen in a row)
        // (the default build; with 16 or 32 slots the values in flight would be spilled)
#if BEAN_AMAX <= 8
        uint8_t amb[kAMax];
        float apu[kAMax];
#pragma unroll
        for (int a = 0; a < kAMax; ++a) {
            const long o = (long)g * A + (a < A ? a : A - 1);
            amb[a] = c.amask[o];
            apu[a] = c.p[4][o];
        }
#pragma unroll
        for (int a = 0; a < kAMax; ++a) {
            const bool am = a < A && amb[a] != 0;
            if (__any(am)) Aw = a + 1;
            alpha[a] = a < A ? (am ? (double)expf(apu[a]) : kEps) : 0.0;
            Ssum += alpha[a];
        }
#else
#pragma unroll
        for (int a = 0; a < kAMax; ++a) {
            const bool am = a < A && c.amask[(long)g * A + a] != 0;
            if (__any(am)) Aw = a + 1;
            alpha[a] = a < A ? (am ? (double)expf(c.p[4][(long)g * A + a]) : kEps) : 0.0;
            Ssum += alpha[a];
        }
#endif
        const double rsq = frcp(Ssum) * pa0;
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
                    double cq0 = alpha[a] * rsq, cq1 = a + 1 < A ? alpha[a + 1] * rsq : 1.0;
                    if (SURV) {  // guide-side clamp (survival_model.py:813-821)
                        cq0 = cq0 < 1e-5 ? 1e-5 : cq0;
                        if (a + 1 < A) cq1 = cq1 < 1e-5 ? 1e-5 : cq1;
                    }
#if BEAN_TL_DIAG == 1
                    GammaPair gp;
                    gp.g0 = 0.3 + cq0;
                    gp.g1 = 0.2 + cq1;
                    gp.k = rng.k;
#else
                    const GammaPair gp = sample_gamma_pair(cq0, cq1, rng);
#endif
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
    }
    BEAN_STAMP_TR(2);
    phase_prio(1);
    if ((c.flags & kDumpPi) && valid) {
#pragma unroll
        for (int a = 0; a < kAMax; ++a)
            if (a < A) c.pi_out[((long)r * G + g) * A + a] = rgm ? pi[a] : 1.0 / A;
    }
    const double u = SURV ? c.u_g[g] : 0.0;
    // ---- accessibility transform (utils.py:106-178) and e[b] = sum_a pe_a P_a[b].  The transformed shares live in
    // registers for the mix only; the backward loop forms each allele's share again (acc_share) where it needs it
    {
        double pe[kAMax];
        pe[0] = pi[0];
#pragma unroll
        for (int a = 1; a < kAMax; ++a) pe[a] = pi[a];
        if (ACC) {
            const double kacc = c.kacc[g];
            const double lpn = c.lpn[g];
            double sum = 0.0;
#pragma unroll
            for (int a = 1; a < kAMax; ++a) {
                if (a < A) {
                    pe[a] = acc_share(pi[a], kacc, lpn).pea;
                    sum += pe[a];
                }
            }
            pe[0] = 1.0 - sum;
        }
#if !defined(BEAN_TL_PLAIN_FWD) && BEAN_AMAX <= 8
        {
            // the table entries of condition b + 1 are asked for before those of b are used - these are the kernel's cold
            // round trips to the allele tables, five in a row on every wave's chain: same sums.  An eighth of config 3
            // (6 250 guides, one wave per SIMD): 85.6 -> 80.1 us per step, the kernel 69.9 -> 63.6; the whole screen
            // 142.5 -> 140.8.  (The default build; 119 VGPRs.)
            double tp[kAMax], tn[kAMax];
#pragma unroll
            for (int a = 1; a < kAMax; ++a) tp[a] = a < Aw ? c.tabP[(long)(a - 1) * G + g] : 0.0;
#pragma unroll 1
            for (int b = 0; b < B; ++b) {
#pragma unroll
                for (int a = 1; a < kAMax; ++a)
                    tn[a] = (a < Aw && b + 1 < B) ? c.tabP[((long)(b + 1) * A1 + (a - 1)) * G + g] : 0.0;
                double v = pe[0] * (SURV ? exp(u * uniform_ld(c.time, b)) : uniform_ld(c.P0, b));
#pragma unroll
                for (int a = 1; a < kAMax; ++a)
                    if (a < Aw) v += pe[a] * tp[a];
                es[b * NT] = v;
                gs[b * NT] = 0.0;
#pragma unroll
                for (int a = 1; a < kAMax; ++a) tp[a] = tn[a];
            }
        }
#else
#pragma unroll 1
        for (int b = 0; b < B; ++b) {
            double v = pe[0] * (SURV ? exp(u * uniform_ld(c.time, b)) : uniform_ld(c.P0, b));
#pragma unroll
            for (int a = 1; a < kAMax; ++a)
                if (a < Aw) v += pe[a] * c.tabP[((long)b * A1 + (a - 1)) * G + g];
            es[b * NT] = v;
            gs[b * NT] = 0.0;
        }
#endif
    }
    BEAN_STAMP_TR(3);
    // ---- both Dirichlet-Multinomial terms, d nll / d e[b] accumulated in gs
    double nll = 0.0;
    {
        const double* sm = c.smask + r * B;
        const double epsB = kEps / (double)B;
#pragma unroll 1
        for (int lik = 0; lik < 2; ++lik) {
            if (lik == 1 && !use_bc) break;
            const float* xp = (lik ? c.Xbc : c.X) + (long)r * B * G + g;  // condition b at xp[b * G]
            const double* sf = (lik ? c.sf_bc : c.sf) + r * B;
            const double nn = lik ? n_bc : n_x;
            double S = 0.0;
#pragma unroll 1
            for (int b = 0; b < B; ++b) S += es[b * NT] * sf[b];
            if (!(nn > (double)c.mask_thres)) continue;
            const double a0 = lik ? c.a0_bc[g] : c.a0[g];
            const double inv = frcp(S + kEps);
            double A0 = 0.0, lsum = 0.0, Ua = 0.0, Va = 0.0;
            bool anyfl = false;
            int b = 0;
#if BEAN_TL_DIAG != 3 && !defined(BEAN_TL_SINGLE_BINS) && BEAN_AMAX <= 8
            // two conditions per pass: their lgamma / digamma differences as two chains side by side (lgamma_digamma_diff2:
            // the same operations per chain, the same bits); the sums take the two in order.  Config 3: 142.9 -> 141.8 us per
            // step (117 instead of 108 VGPRs, nothing spilled; -DBEAN_TL_SINGLE_BINS: one per pass).  The default build only: the
            // 16-allele build, which spills already, loses by it (16 slots / 4 edited alleles, 20 000 guides: 211.9 against 205.4)
#pragma unroll 1
            for (; b + 1 < B; b += 2) {
                const double araw0 = (es[b * NT] * sf[b] + epsB) * inv * a0 * sm[b];
                const double araw1 = (es[(b + 1) * NT] * sf[b + 1] + epsB) * inv * a0 * sm[b + 1];
                const bool fl0 = araw0 < kEps, fl1 = araw1 < kEps;
                anyfl = anyfl || fl0 || fl1;
                const double al0 = fl0 ? kEps : araw0, al1 = fl1 ? kEps : araw1;
                const DD2 dd = lgamma_digamma_diff2(al0, (double)xp[(long)b * G], al1, (double)xp[(long)(b + 1) * G]);
                A0 += al0;
                lsum += dd.a.d;
                ds[b * NT] = dd.a.dp;
                Ua += fl0 ? 0.0 : araw0;
                Va += fl0 ? 0.0 : dd.a.dp * araw0;
                A0 += al1;
                lsum += dd.b.d;
                ds[(b + 1) * NT] = dd.b.dp;
                Ua += fl1 ? 0.0 : araw1;
                Va += fl1 ? 0.0 : dd.b.dp * araw1;
            }
#endif
#pragma unroll 1
            for (; b < B; ++b) {
                const double araw = (es[b * NT] * sf[b] + epsB) * inv * a0 * sm[b];
                const bool floored = araw < kEps;
                anyfl = anyfl || floored;
                const double al = floored ? kEps : araw;
                A0 += al;
#if BEAN_TL_DIAG == 3
                DD db;
                db.d = al * 0.5;
                db.dp = (double)xp[(long)b * G] * 1e-3;
#else
                const DD db = lgamma_digamma_diff_inl(al, (double)xp[(long)b * G]);
#endif
                lsum += db.d;
                ds[b * NT] = db.dp;
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
                const double sfb = sf[b], smb = sm[b];
                const double araw = (es[b * NT] * sfb + epsB) * inv * a0 * smb;
                const double ga = araw < kEps ? 0.0 : d0.dp - ds[b * NT];
                gs[b * NT] += (ga * a0 * smb * inv - W) * sfb;
            }
        }
    }
    BEAN_STAMP_TR(4);
    phase_prio(2);
    // ---- back through the mixture: d loss / d pi_a and the per-allele-slot rows
    // (survival: the guide's baseline growth draw is loaded again rather than carried across the likelihoods)
    double u_b = 0.0;
    if (SURV) {
        __asm__ volatile("" ::: "memory");
        u_b = c.u_g[g];
    }
    double s0 = 0.0;
#pragma unroll 1
    for (int b = 0; b < B; ++b)
        s0 += gs[b * NT] * (SURV ? exp(u_b * uniform_ld(c.time, b)) : uniform_ld(c.P0, b));
    // survival: control_allele_count ~ Multinomial(pi * exp(mu * t_ctrl)), mu = [u, u + mu_a]
    // (survival_model.py:535-548) adds to d / d pi_a AND to d / d mu_a.  Its per-control-timepoint normaliser
    // 1 / W and in-range count n_in are formed FIRST (they need every allele), so that the loop over alleles
    // below finishes d / d mu_a of an allele and hands the row over at once: no array of kAMax - 1 partial
    // results is carried across the control term (round 3: 14 - 22 spilled VGPRs in the survival builds).
    // kCtrlRegs control timepoints are kept in registers (ONE: what the reference's screens have - a second pair cost
    // the accessibility build four of its six spilled VGPRs); further ones
    // are re-formed per allele.
    constexpr int kCtrlRegs = 1;
    double ctl_rW[kCtrlRegs], ctl_nin[kCtrlRegs];
    auto ctrl_norm = [&](int cc, double& rW, double& n_in) {
        const double tc = c.ctrl_time[cc];
        double W = 0.0;
#pragma unroll
        for (int a = 0; a < kAMax; ++a)
            if (a < A) W += pi[a] * exp((a == 0 ? u_b : u_b + c.mu_a[(long)(a - 1) * G + g]) * tc);
        rW = frcp(W);
        n_in = 0.0;
#pragma unroll
        for (int a = 0; a < kAMax; ++a)
            if (a < A) {
                const double wv = pi[a] * exp((a == 0 ? u_b : u_b + c.mu_a[(long)(a - 1) * G + g]) * tc);
                const double pr = wv * rW;
                const double cnt = (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                nll -= cnt * flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                if (pr > kProbEps && pr < 1.0 - kProbEps) n_in += cnt;
            }
    };
    if (SURV) {
#pragma unroll
        for (int k = 0; k < kCtrlRegs; ++k) {
            ctl_rW[k] = 0.0;
            ctl_nin[k] = 0.0;
            if (k < c.C) ctrl_norm(k, ctl_rW[k], ctl_nin[k]);
        }
    }
    // one allele's share of the control term: added to its d / d pi_a (gp) and d / d mu_a (gm), control
    // timepoint by control timepoint (the order of the additions of the round-3 form)
    auto ctrl_allele = [&](int a, double& gp, double& gm) {
        for (int cc = 0; cc < c.C; ++cc) {
            double rW, n_in;
            if (cc < kCtrlRegs) {
                rW = ctl_rW[0];
                n_in = ctl_nin[0];
            } else {
                // re-formed; the loss terms ctrl_norm adds are counted once, by allele 0 (the first call)
                const double nll_keep = nll;
                ctrl_norm(cc, rW, n_in);
                if (a != 0) nll = nll_keep;
            }
            const double tc = c.ctrl_time[cc];
            const double gr = exp((a == 0 ? u_b : u_b + c.mu_a[(long)(a - 1) * G + g]) * tc);
            const double wv = pi[a] * gr, pr = wv * rW;
            const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
            const double cnt = (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
            gp += ((inside ? -cnt * frcp(wv) : 0.0) + n_in * rW) * gr;
            if (a >= 1) gm += ((inside ? -cnt : 0.0) + n_in * wv * rW) * tc;
        }
    };
    double gpi[kAMax], gnoise = 0.0;
    gpi[0] = ACC ? 0.0 : s0;
    // accessibility: the guide's factor and noise draw are loaded AGAIN here (the compiler barrier makes them new
    // values to it), so neither they nor the shares formed from them in the forward pass are carried across the
    // likelihoods in registers - which is what the parked LDS columns were for
    double kacc_b = 0.0, lpn_b = 0.0;
    if (ACC) {
        __asm__ volatile("" ::: "memory");
        kacc_b = c.kacc[g];
        lpn_b = c.lpn[g];
    }
    if (SURV) {
        double unused = 0.0;
        ctrl_allele(0, gpi[0], unused);
    }
#pragma unroll
    for (int a = 1; a < kAMax; ++a) {
        gpi[a] = 0.0;
        if (a < A) {
            double sa = 0.0, dm = 0.0, dsg = 0.0;
            if (a < Aw) {
#if !defined(BEAN_TL_PLAIN_BWD) && BEAN_AMAX <= 8
                // the table entries of condition b + 1 are asked for before those of b are used: same sums.  Config 3
                // 142.3 -> 141.6 us per step, cut in eight 86.0 -> 85.4 (an earlier state of the kernel: nothing)
                long o = (long)(a - 1) * G + g;
                const long ob = (long)A1 * G;
                double tp = c.tabP[o], tm = c.tabPmu[o], ty = SURV ? 0.0 : c.tabPy[o];
#pragma unroll 1
                for (int b = 0; b < B; ++b) {
                    double np = 0.0, nm = 0.0, ny = 0.0;
                    if (b + 1 < B) {
                        o += ob;
                        np = c.tabP[o];
                        nm = c.tabPmu[o];
                        if (!SURV) ny = c.tabPy[o];
                    }
                    const double ge = gs[b * NT];
                    sa += ge * tp;
                    dm += ge * tm;
                    if (!SURV) dsg += ge * ty;
                    tp = np;
                    tm = nm;
                    ty = ny;
                }
#else
#pragma unroll 1
                for (int b = 0; b < B; ++b) {
                    const long o = ((long)b * A1 + (a - 1)) * G + g;
                    const double ge = gs[b * NT];
                    sa += ge * c.tabP[o];
                    dm += ge * c.tabPmu[o];
                    if (!SURV) dsg += ge * c.tabPy[o];
                }
#endif
            }
            double pea = pi[a];
            if (ACC) {
                const AccShare sh = acc_share(pi[a], kacc_b, lpn_b);
                pea = sh.pea;
                gpi[a] = (sa - s0) * sh.dpi;
                gnoise += (sa - s0) * sh.dl;
            } else {
                gpi[a] = sa;
            }
            double gma = pea * dm;
            if (SURV) ctrl_allele(a, gpi[a], gma);
            row_out(kTGmu + a - 1, gma);
            row_out(kTGsig + a - 1, pea * dsg);
        }
    }
    BEAN_STAMP_TR(5);
    phase_prio(3);
    // ---- Multinomial on control allele counts (sorting; the survival form is the control term above)
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
                double cnt = 0.0;
                for (int cc = 0; cc < c.C; ++cc)
                    cnt += (double)c.allele[(((long)r * c.C + cc) * G + g) * A + a];
                // (a slot without a count anywhere in the wave - the empty slots - adds exact zeros)
                if (__any(cnt != 0.0)) {
                    const double pr = pi[a] * rsum;
                    const bool inside = pr > kProbEps && pr < 1.0 - kProbEps;
                    const double lg = inside ? flog(pi[a]) - ls : flog(fmin(fmax(pr, kProbEps), 1.0 - kProbEps));
                    nll -= cnt * lg;
                    if (inside) gpi[a] -= cnt * frcp(pi[a]);
                }
            }
        }
    }
    // ---- the two Dirichlet log-densities of the pi site; the concentrations are formed again from
    // alpha_pi (same expressions as before the draw, same values).  The barrier keeps the compiler
    // from carrying them across the likelihoods instead (which is what spilled).
    __asm__ volatile("" ::: "memory");
    {
        // (the guide index as a new value: otherwise the 64-bit addresses formed from it before the draw are kept
        // - and spilled - to be used again here)
        int ge = g;
        asm volatile("" : "+v"(ge));
        double alpha[kAMax], Ssum = 0.0;
#if BEAN_AMAX <= 8
        uint8_t amb[kAMax];
        float apu[kAMax];
#pragma unroll
        for (int a = 0; a < kAMax; ++a) {  // (all asked for together, as before the draw)
            const long o = (long)ge * A + (a < A ? a : A - 1);
            amb[a] = c.amask[o];
            apu[a] = c.p[4][o];
        }
#pragma unroll
        for (int a = 0; a < kAMax; ++a) {
            const bool am = a < A && amb[a] != 0;
            alpha[a] = a < A ? (am ? (double)expf(apu[a]) : kEps) : 0.0;
            Ssum += alpha[a];
        }
#else
#pragma unroll
        for (int a = 0; a < kAMax; ++a) {
            const bool am = a < A && c.amask[(long)ge * A + a] != 0;
            alpha[a] = a < A ? (am ? (double)expf(c.p[4][(long)ge * A + a]) : kEps) : 0.0;
            Ssum += alpha[a];
        }
#endif
        const double pa0e = c.pi_a0[ge];  // (loaded again behind the barrier, not carried: see above)
        const double rsq = frcp(Ssum) * pa0e;
        // model-side floored concentration c_p (model.py:640-651) for - d log p / d pi
        const double rSe = frcp(Ssum + kEps) * pa0e;
        double total = 0.0, proj = 0.0;
#pragma unroll
        for (int a = 0; a < kAMax; ++a) {
            double cqa = alpha[a] * rsq;
            if (SURV && a < A && cqa < 1e-5) cqa = 1e-5;
            total += cqa;
            if (a < A) {
                const double rpi = frcp(pi[a]);
                row_out(kTL + a, flog(pi[a]));
                gpi[a] += (cqa - 1.0) * rpi;  // + d log q / d pi
                const double v = (alpha[a] + kEps / A) * rSe;
                gpi[a] -= ((v < kEps ? kEps : v) - 1.0) * rpi;
                proj += pi[a] * gpi[a];
            }
        }
        const double dgS_t = c.dgq_t[(long)kAMax * G + ge];  // digamma(sum c_q), tabulated by k_param
        // (digamma(c_q a) of every slot asked for now: asked for in front of its call, each was a cold round trip that the
        // call began by waiting for)
        // (the default build; with 16 or 32 slots they would be spilled across the calls)
#if BEAN_AMAX <= 8
        double dgq_a[kAMax];
#pragma unroll
        for (int a = 0; a < kAMax; ++a) dgq_a[a] = c.dgq_t[(long)(a < A ? a : A - 1) * G + ge];
#define BEAN_TL_DGQ(a_) dgq_a[a_]
#else
#define BEAN_TL_DGQ(a_) c.dgq_t[(long)(a_) * G + ge]
#endif
#pragma unroll
        for (int a = 0; a < kAMax; ++a)
            if (a < A) {
                double cqa = alpha[a] * rsq;
                if (SURV && cqa < 1e-5) cqa = 1e-5;
#if BEAN_TL_DIAG == 2
                row_out(kTPath + a, (pi[a] + cqa) * (gpi[a] - proj));
#else
                row_out(kTPath + a,
                        dirichlet_grad_one_pre(pi[a], cqa, total, BEAN_TL_DGQ(a), dgS_t) * (gpi[a] - proj));
#endif
            }
    }
#undef BEAN_TL_DGQ
    BEAN_STAMP_TR(6);
    if (ACC) row_out(kTGnoise, gnoise);
    row_out(kTNrg, 1.0);
    rows_flush();
    const double tot = wave_sum(on ? nll : 0.0);
    if ((lane & 63) == 0) {  // every wave of the workgroup adds its part
        loss_add(c, ctr.slot, tot);
        if (blockIdx.x == 0 && lane == 0) publish_ctr(c, ctr);
    }
    BEAN_STAMP_TR(7);
}

}  // namespace bean

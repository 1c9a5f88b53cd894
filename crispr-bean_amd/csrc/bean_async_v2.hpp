// k_svi_async: ALL the SVI steps of a call in one launch, with no grid-wide boundary between steps
// (variant sorting families whose parameters are all per target or per guide - the metric workload).
//
// What the two launches per step cost ("What one launch costs"): ~25 of the metric step's 55 us are
// not arithmetic but structure - 3.5 us until a launch's first loads arrive, ~10 us in which the SIMDs drain (each SIMD's
// youngest wave finishes alone, latency-bound), and a k_param launch of ~10 us that is a chain of memory round
// trips.  The fused step kernel (bean_step_v2.hpp) removed the second launch and lost: all tiles finish together, so the
// finishing waves run alone at the end of the launch.  The tile-persistent loop (bean_tile_svi.hpp) removed the
// launches and lost: a workgroup owned a tile for the whole call and its waves waited for each other at two barriers
// per step.
//
// Here nothing waits for the chip.  Single-wave workgroups stay resident and PULL work items (step, tile, replicate)
// from a queue; an item is k_guide_wave2's wave work (guide_wave2_tile, the same code, the same bits); the wave that
// completes a tile last finishes it (or, with finisher roles, hands the finish to a wave that does nothing else: below)
// - k_step_wave2's hand-over and k_param's per-target / per-guide code - and the finish publishes "tile k has completed
// step s" (one word for the tile's targets, one for its guides).  An item of step s + 1 waits (a bounded poll) for those
// words of its own tile and for the target words of the two neighbours it may share a target with, and for nothing else:
// one tile's finish chain runs while other tiles' waves keep the SIMDs busy, and a step never drains.
//
// (Measured and not adopted: pulling a wave's NEXT item at the head of the current one, to hide the queue atomic's
// round trip - an item that is claimed but not started delays its tile, and every tile's delay is the next step's wait:
// 49.8 -> 61.0 us per step at 50k guides, 83.6 -> 92.6 at 100k.  A 237-register build at two waves per SIMD (while the
// build still ran MachineLICM, the only way to inline the pieces without scratch): 51.5 -> 53.0.  A work-conserving queue - one ring per group that holds READY items
// in the order they became ready, filled by the finish that completes a tile's three dependencies, so that no wave holds
// an item that cannot start: same bits, and 51.0 us at two waves per SIMD where this form takes 50.6, 65 / 64 at three /
// four against 59.6 / 67.7.  Head-of-line blocking is not what more resident waves lose to; a wave's own chain growing
// with its SIMD's load is, and a tile waits for the slowest of its R waves.  With finisher roles, the next item pulled
// between an item's last stores and the wait for them (the queue atomic under the store drain): 48.8 against 48.4 us at
// 50k guides, 57.6 against 57.9 at 62.5k - nothing.  A table of every tile's first target, target count and offsets
// (one load at the head of a finish instead of two dependent lookups, the boundary counters asked before DevArgs have
// arrived): 48.67 against 48.74, 58.06 against 58.2 - nothing.  A 168-register build (three waves per SIMD: what the
// grids with finisher roles run; 137 - 141 VGPRs, nothing spilled, where the 128-register build spills 3 - 11): 53.5
// against 49.0 and 63.0 against 58.0 - slower, as the 237-register build was.  In the finish: the boundary counters, the R
// waves' loss parts (one word per lane) and the guides' state requested in one batch at the top and the guides' part
// moved in front of the targets' - 18.8 -> 21.2 us per finish (more values live across the chain, 36 B more scratch).
// A 256-register build for the two-waves-per-SIMD grids whose bin loop takes two bins per pass (four lgamma / digamma
// chains side by side instead of two; same bits): 50.4 - 51.5 us against 50.7 - two chains already issue a float64
// instruction every ~5.5 cycles, the pipe's rate for one wave.)
//
// Where a wave's time goes at the metric shape, two waves per SIMD (scripts/async_timeline.py, a -DBEAN_ASYNC_STAMP
// build; medians): an item waits 1.2 us for its dependencies (the poll's own round trip), computes for 19.1 us (a lone
// wave's chain: the SIMD's second wave fills 65 % of its issue slots), drains its stores and arrives in 1.4 us; one
// item in five then finishes its tile in 18.8 us - the call and the boundary counters 3.6, sums / ClippedAdam / draw
// 4.7, Phi tables 2.8, the guides' alpha_pi and digamma tables 5.0, loss parts + store drain + publish 2.8.  A wave is
// busy for 90 - 93 % of a step.
//
// Order and progress.  The queue is one counter per XCD group (blockIdx & 7, a label: the blocks that share a label
// share an L2; nothing depends on it but speed), items in (step, tile, replicate) order; tile k belongs to group
// k & 7, so whatever a tile reads and writes over the steps stays in one L2 except the targets it shares with its
// neighbours.  An item depends only on items of the PREVIOUS step, and a group hands out all items of step s before any
// of step s + 1: a wave that polls waits for waves that hold earlier items and do not wait for it - no cycle, whatever
// the residency.  Every poll is bounded (kAsyncSpinMax) and watches an abort word; a wave that gives up sets it, every
// wave then leaves at its next poll or pull, the grid drains, and the call's loss slots report NaN (the host halts a
// fit on a non-finite loss window).
//
// Visibility (MI355X_MICROARCH.md, "inter-workgroup visibility"; the forms measured valid there).  Per-XCD L2s are not
// coherent with each other and a CU's L1 is never refreshed by another CU's stores.  Everything one wave writes and
// another wave of this launch reads - rows, per-target sums, loss parts (guide waves -> finisher); parameters, moments,
// draws, Phi tables, digamma tables (finisher -> the tile's next step) - is STORED at agent scope (sc1: written
// through) and LOADED at agent scope (sc1: bypasses L1); the writer waits for its stores (s_waitcnt vmcnt(0)) before
// the relaxed agent-scope atomic that signals them (arrival counter, completed-step word); the reader issues its loads
// after the atomic it polled has returned.  No fences (an agent-scope fence per wave was measured at 46 -> 189 us per
// launch).  Data that no wave writes (counts, masks, size factors, a0, offsets) is read with plain loads.
//
// Bit-identical to the two launches per step (tests/test_gpu_async.py): same per-pair math, same summation orders,
// integer loss accumulation.
#pragma once

#include "bean_devargs_sgpr.hpp"

// The two pieces of an item are inlined into the item loop.  That needs the build's -mllvm -disable-machine-licm
// (_lib.HIPCC_FLAGS): with the pass, the loop keeps every hoisted constant live across both pieces (448 B of scratch per
// lane at 128 VGPRs), and out-of-line pieces (-DBEAN_ASYNC_INLINE=__noinline__, how this kernel was first built) save and
// restore ~60 callee-saved registers per call - 82 of the 122 MB the launch then moved per step were that.
#ifndef BEAN_ASYNC_INLINE
#define BEAN_ASYNC_INLINE __forceinline__
#endif
#ifndef BEAN_ASYNC_PRIO
#define BEAN_ASYNC_PRIO 1
#endif
#ifndef BEAN_ASYNC_SLEEP
#define BEAN_ASYNC_SLEEP 8
#endif
#ifndef BEAN_ASYNC_EU
#define BEAN_ASYNC_EU BEAN_WAVE_EU
#endif

namespace bean {

// When it is the default (BEAN_HIP_STEP=async forces it for any eligible screen, =pair switches it off), measured on
// MI355X against the two launches per step (scripts/time_async.py, R = 5; us per step, async at its best grid / pair;
// the build without MachineLICM): 25k guides 39.9 / 37.2, 37.5k 44.3 / 46.4, 50k 48.3 / 55.1, 62.5k 60.7 / 66.1,
// 75k 64.1 / 75.2, 100k 78.7 / 95.8, 150k 110 / 132, 250k 191 / 204 (first build), 500k 387 / 384 (first build): from
// ~2 800 items (tile, replicate) per step upwards.  Below that a step is one tile's dependency chain either way and
// k_param's chip-wide launch is the shorter chain.
constexpr long kAsyncMinItems = 2800;
// ... and up to 24 000 items (~300k guides at R = 5): beyond, both forms sit on the same issue bound in long windows and a
// short call pays the pipeline's ramp (500k guides, one 20-step call: 456 against 407 us per step).
constexpr long kAsyncMaxItems = 24000;
// Resident waves per SIMD.  Fewer waves than the chip holds is FASTER while a step has few items per wave: a wave's
// guide work is a latency chain (20 us alone, 34 us with three neighbours on its SIMD), a tile's next step waits for the
// slowest of its R waves plus the finish, and with ~2 items per wave and step every wave always finds an item whose
// dependencies are met (measured wait: 0.8 us median at two waves per SIMD, 25 us at four - scripts/async_timeline.py).
// 50k guides: 2 / 3 / 4 waves per SIMD 48.3 / 54.9 / 60.2 us per step; 75k: 72.9 / 64.1 / 70.3; 100k: 97.2 / 78.7 / 86.3;
// 150k: 144 / 118 / 110.
// Grids that do not give every SIMD the same number of waves lose 10 - 20 % (2304 blocks: 60.4 us at 50k guides).
// (All of these were measured on 1 024 SIMDs - 256 CUs - and are applied per SIMD: a partition with fewer CUs gets the
// same items per wave.)
__host__ __device__ inline int async_waves_per_simd(long items, long simds = 1024) {
    const double per_simd = (double)items / (double)(simds > 0 ? simds : 1);
    return per_simd <= 5400.0 / 1024.0 ? 2 : (per_simd <= 11000.0 / 1024.0 ? 3 : 4);
}
__host__ __device__ inline bool async_size_in_range(long items, long simds = 1024) {
    const double per_simd = (double)items / (double)(simds > 0 ? simds : 1);
    return per_simd >= (double)kAsyncMinItems / 1024.0 && per_simd <= (double)kAsyncMaxItems / 1024.0;
}
// Finisher roles (round 5, second form).  One more wave per SIMD that does nothing but finish tiles: the wave that
// completes a tile appends two entries to its group's finish ring - the tile's TARGETS (sums, priors, ClippedAdam, draw,
// Phi tables, the R waves' loss parts) and its GUIDES (alpha_pi, noise site, digamma tables): two chains that do not
// depend on each other - and pulls its next item at once; a finisher takes the ring's next position and polls it, so
// the two parts of a tile run on two waves at the same time (11 and 5 us instead of 19 in a row), and an item waits for
// its tile's two completed-step words and its neighbours' target words.  The item waves lose the fifth of their time they
// spent finishing; a finisher is a latency chain that takes few issue slots from its SIMD's item waves.  It pays where an
// item wave has about two items per step or more (us per step, roles / no roles, scripts/time_async.py, same box per
// pair): 43.75k guides 46.9 / 46.4 (1.7 items per wave: no), 50k 47.8 / 49.6, 62.5k 58.4 / 62.2 (1.9, 2.4: yes); at three
// item waves per SIMD 75k 69.0 / 64.8, 87.5k 71.7 / 70.1 (1.9, 2.2: no), 100k 77.8 / 79.9, 125k 96.6 / 99.5 (2.55, 3.2: yes);
// 512 ... 2 048 finishers make no difference.  (With the finish in ONE piece the roles lost at 50k - 50.3 / 48.8: the hop
// through the ring made the tile's chain longer than the item waves' step - and won as much at 62.5k, 57.4 / 61.0.)
// Progress does not depend on the finishers being resident: a wave whose item has waited kAsyncStealEvery polls takes the
// oldest finish nobody has taken, and a wave that finds no items left joins the finishers
// (tests/test_gpu_async.py runs both).
__host__ __device__ inline bool async_finisher_roles(int waves_per_simd, long items, long simds) {
    const double per_wave = (double)items / (double)(waves_per_simd * simds);
    return (waves_per_simd == 2 && per_wave >= 1.8) || (waves_per_simd == 3 && per_wave >= 2.4);
}
constexpr int kAsyncQueueStride = 32;      // ints between two groups' queue counters (separate 128-byte lines)
constexpr int kAsyncStealEvery = 64;       // polls between two looks of a waiting item's wave into the finish ring
constexpr int kAsyncSpinMax = 1 << 21;     // polls before a wave gives up (~0.3 us each: over half a second)

struct AsyncArgs {
    unsigned long long step0, slot0;  // first step of the call and its loss slot
    int n_steps;
    int* queue;                // [8 * kAsyncQueueStride] next item of each group; zero when the call starts
    int* done;                 // [2 n_tiles] steps of THIS call whose finish is complete: [2 k] tile k's targets, [2 k + 1] its guides
    int* abort_flag;           // [1] set by a wave whose poll ran out
    // finisher roles (n_guide_blocks > 0): blocks from n_guide_blocks upwards only finish tiles; the wave that completes a
    // tile appends (step, tile) to its group's finish ring instead of finishing it itself
    int n_guide_blocks;        // 0: every wave pulls items and the last arriver finishes (no roles)
    int fin_split;             // 1: a tile's finish is two ring entries (targets, guides); 0: one
    int* fhead;                // [8 * kAsyncQueueStride] next ring position a finisher takes
    int* ftail;                // [8 * kAsyncQueueStride] next ring position to be filled
    int* fring;                // [8 * fring_stride] ((step + 1) << 18) | (part: 0 targets, 1 guides, 2 both) << 16 | (tile >> 3); 0 = not filled yet
    long fring_stride;
    const float* step_sizes;   // [n_steps] ClippedAdam step size of the update of step0 + i (k_step_sizes)
    unsigned long long* stamps;  // diagnostic builds (-DBEAN_ASYNC_STAMP): kAsyncStampSteps x items x 8 words, or null
};
// diagnostic builds: the items of local steps [kAsyncStampStep0, + kAsyncStampSteps) of a call leave their timeline
// (real-time clock, 100 MHz): pulled, dependencies seen, guide work done, arrived, finished / published
constexpr int kAsyncStampStep0 = 40, kAsyncStampSteps = 4;
constexpr long kAsyncStampFinOff = 1l << 22;  // words between an item's row and its finish-phase row (host: 2 x 2^22 words)
#ifdef BEAN_ASYNC_STAMP
#define BEAN_ASYNC_T(k)                                                                              \
    if (st_row) {                                                                                    \
        unsigned long long u_;                                                                       \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u_)::"memory");               \
        if (lane == 0) st_row[k] = u_;                                                               \
    }
// (finish phases: a second table behind the items', one row per item too)
#define BEAN_ASYNC_TF(k)                                                                             \
    if (st_row) {                                                                                    \
        unsigned long long u_;                                                                       \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(u_)::"memory");               \
        if (lane == 0) st_row[kAsyncStampFinOff + (k)] = u_;                                         \
    }
#define BEAN_ASYNC_ST_ARG , unsigned long long* st_row
#define BEAN_ASYNC_ST_PASS , st_row
#else
#define BEAN_ASYNC_T(k)
#define BEAN_ASYNC_TF(k)
#define BEAN_ASYNC_ST_ARG
#define BEAN_ASYNC_ST_PASS
#endif

// DevArgs as the kernel received them -> their copy in global memory (stream-ordered, no host staging)
__global__ __launch_bounds__(64) void k_put_args(DevArgs c, DevArgs* out) {
    const unsigned int* src = (const unsigned int*)&c;
    unsigned int* dst = (unsigned int*)out;
    for (unsigned i = threadIdx.x; i < sizeof(DevArgs) / 4; i += 64) dst[i] = src[i];
}

// ClippedAdam step sizes of the n updates that follow update `step0` (update t = step + 1), one thread each, with the
// device's exp / pow (adam_coef) so that every path holds the same float32 value
__global__ __launch_bounds__(256) void k_step_sizes(DevArgs c, unsigned long long step0, int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = adam_coef(c, step0 + (unsigned long long)i + 1).step_size;
}

// Everything the launch needs, in ONE small launch in front of it: the queue counters, the abort word, the
// completed-step words and the arrival counters at zero (they run on through a call: a multiple of R arrivals closes a
// tile's step), the ClippedAdam step sizes of the call's updates (the device's exp / pow, adam_coef: every path holds
// the same float32 values), DevArgs as this kernel received them copied to global memory for the out-of-line pieces, and
// the step counters as n {guide, k_param} pairs would LEAVE them - k_svi_async does not read them; the pair path, the loss
// finalize and a later resume continue from there.
__global__ __launch_bounds__(256) void k_async_head(DevArgs c, AsyncArgs a, DevArgs* args_out, int* ws, int ws_ints,
                                                    float* step_sizes) {
    // (a few blocks: the finish rings are n_steps x tiles ints)
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
    for (long i = tid; i < 8 * a.fring_stride; i += nth) a.fring[i] = 0;
    if (blockIdx.x != 0) return;
    for (int i = threadIdx.x; i < ws_ints; i += 256) ws[i] = 0;
    for (int i = threadIdx.x; i < c.n_arrival_ctr; i += 256) c.tile_ctr[i] = 0;
    for (int i = threadIdx.x; i < a.n_steps; i += 256) step_sizes[i] = adam_coef(c, a.step0 + (unsigned long long)i + 1).step_size;
    const unsigned int* src = (const unsigned int*)&c;
    unsigned int* dst = (unsigned int*)args_out;
    for (unsigned i = threadIdx.x; i < sizeof(DevArgs) / 4; i += 256) dst[i] = src[i];
    if (threadIdx.x != 0) return;
    StepCtr last, next;
    last.step = a.step0 + a.n_steps - 1;
    last.slot = a.slot0 + a.n_steps - 1;
    last.step_size = adam_coef(c, last.step + 1).step_size;
    last.pad_ = 0.f;
    next.step = last.step + 1;
    next.slot = last.slot + 1;
    next.step_size = 0.f;
    next.pad_ = 0.f;
    *c.ctrA = last;
    *c.ctrB = next;
}

// A wave whose poll ran out: every wave leaves at its next poll or pull, and the call's loss slots report NaN
// (the host halts a fit on a non-finite loss window).
__device__ __forceinline__ void async_give_up(const DevArgs* cp, const AsyncArgs& a) {
    __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long* acc = cp->loss_acc;
    for (int i = 0; i < a.n_steps; ++i)
        atomicAdd((unsigned long long*)(acc + ((long)(a.slot0 + i) * kLossSub) * kLossWords) + 2, 1ull);
}

// Finish tile `tile` of the step `ctr` describes: k_param's work for the tile's targets and guides (FINISH of this
// step, PREP of the next), by the wave that arrived last.  The structure and the arithmetic are k_step_wave2's
// (bean_step_v2.hpp; its comments explain the lane maps); what differs is that every access to state that lives
// across steps is agent-scope, because the wave that finishes this tile's NEXT step runs on another CU.
// DevArgs come from their copy in global memory into SGPRs (bean_devargs_sgpr.hpp), read again for every item: nothing
// derived from them is loop-invariant to the compiler (with MachineLICM on and DevArgs as a kernel argument, the
// inlined loop spilled 249 VGPRs: 576 B of scratch per lane).
template <int FAM, bool ACC>
__device__ BEAN_ASYNC_INLINE void async_finish_tile(const DevArgs* cp, unsigned long long step, unsigned long long slot,
                                               float step_size, int tile, int part BEAN_ASYNC_ST_ARG) {
    // part: 1 = the tile's targets (sums, priors, ClippedAdam, draw, Phi tables; the R waves' loss parts), 2 = its guides
    // (alpha_pi, the noise site, digamma tables), 3 = both.  The two do not depend on each other: with finisher roles they
    // are two entries of the finish ring and run on two waves at the same time.
    const DevArgs c = dev_args_in_sgprs(cp);
    part = rfl_i(part);
    step = rfl_u64(step);
    slot = rfl_u64(slot);
    step_size = __builtin_bit_cast(float, rfl_i(__builtin_bit_cast(int, step_size)));
    tile = rfl_i(tile);
    int t0, nt;  // the tile's first target and their number (as guide_wave2_tile finds them)
    {
        // (the two lookups ride on lanes 0 and 1 of ONE load - a uniform value's load is waited for where it is made
        // scalar, so one after the other they were two round trips at the head of the tile's chain; the same for the
        // target offsets and the boundary counters below)
        const int gf = tile * 64 - c.g_sh > 0 ? tile * 64 - c.g_sh : 0;
        const int gl = tile * 64 + 63 - c.g_sh < c.G ? tile * 64 + 63 - c.g_sh : c.G - 1;
        const int gv = c.g2t[threadIdx.x == 0 ? gf : gl];
        t0 = __builtin_amdgcn_readlane(gv, 0);
        nt = __builtin_amdgcn_readlane(gv, 1) - t0 + 1;
    }
    StepCtr ctr;
    ctr.step = step;
    ctr.slot = slot;
    ctr.step_size = step_size;
    ctr.pad_ = 0.f;
    constexpr bool MIX = FAM == kMixture;
    constexpr int COH = 2;
    extern __shared__ double tabs[];
    const int lane = threadIdx.x, R = c.R, G = c.G, B = c.B;
    const int t1 = t0 + nt - 1;
    int tof0, tof1;
    {
        const int tv = c.toff[lane == 0 ? t0 : t1 + 1];
        tof0 = __builtin_amdgcn_readlane(tv, 0);
        tof1 = __builtin_amdgcn_readlane(tv, 1);
    }
    const unsigned long long s_prep = ctr.step + 1;
    AdamCoef ak;
    ak.step_size = ctr.step_size;
    ak.clip = (float)c.clip;
    const int g_first = tile * 64 - c.g_sh > 0 ? tile * 64 - c.g_sh : 0;
    const int g_last = tile * 64 + 63 - c.g_sh < G ? tile * 64 + 63 - c.g_sh : G - 1;
    const bool left_str = tof0 < g_first;
    const bool right_str = tof1 > g_last + 1;
    // Which wave finishes a straddling target depends on timing, so the prior / entropy terms are not added up as
    // doubles per wave: every TERM goes into the loss's fixed-point form by itself (fixed_add's split) and the wave
    // adds integers - the loss history is bitwise reproducible run to run, whoever finished what.
    long long loss_hi = 0, loss_lo = 0, loss_bad = 0;
    auto loss_term = [&](double v) {
        if (!(fabs(v) < kLossPartMax)) {
            loss_bad = 1;
            return;
        }
        const double hi = rint(v * 1024.0);
        loss_hi += (long long)hi;
        loss_lo += (long long)rint((v - hi * (1.0 / 1024.0)) * 1099511627776.0);
    };
    if (part & 1) {
    int own_left = left_str ? 0 : 1, own_right = right_str ? 0 : 1;
    {
        // a target that straddles two tiles goes to the tile that completes second (one counter per boundary):
        // lane 0 counts in at the left boundary, lane 1 at the right one - one atomic instruction
        // (running counters, as the tiles': two arrivals per boundary and step, the odd one is the second)
        int* const bp = c.bnd_ctr + (lane == 0 ? (left_str ? tile - 1 : tile) : tile);
        int old = 0;
        if ((lane == 0 && left_str) || (lane == 1 && right_str))
            old = __hip_atomic_fetch_add(bp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int ol = __builtin_amdgcn_readlane(old, 0), orr = __builtin_amdgcn_readlane(old, 1);
        if (left_str && (ol & 1)) own_left = 1;
        if (right_str && (orr & 1)) own_right = 1;
    }
    BEAN_ASYNC_TF(0);
    const int ta = own_left ? t0 : t0 + 1, tb = own_right ? t1 : t1 - 1;  // this wave's targets, ta > tb: none
    double* hmu = tabs;  // drawn mu / y of the targets, hmu[t - ta] (<= 64 targets per tile)
    double* hy = tabs + 64;
    __syncthreads();  // single-wave workgroup: the guide work's LDS is free from here

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
            int2 dsc;
            dsc.x = 0;
            dsc.y = 2;
            if (!c.tsum_direct) dsc = c.tdesc[tc];
            const int n = dsc.y * R, ntm = c.tile_targets;
            const long S = c.tsum_direct ? 2 * (long)c.T : (long)c.n_tiles * ntm;
            float pj = coh_ld<COH>(P + tc), mj = coh_ld<COH>(M + tc), vj = coh_ld<COH>(V + tc);
            const float p1 = coh_ld<COH>(c.p[1] + tc), p3 = coh_ld<COH>(c.p[3] + tc);
            const double eps1 = coh_ld<COH>(c.eps_mu + tc), eps2 = coh_ld<COH>(c.eps_sd + tc);
            const double mu = coh_ld<COH>(c.mu_t + tc), y = coh_ld<COH>(c.y_t + tc);
            // the (part, replicate) sums of the target in k_param's order (16 lanes, xor tree 8, 4, 2, 1)
            double am[4] = {0.0, 0.0, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.0, 0.0};
            const float rR = 1.0f / (float)R;
            for (int i0 = 0; i0 < n; i0 += 32) {
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
            if (act && j == 0) loss_term(lt);
            const double Gd = j < 2 ? gmu - dlogp_mu : gy - dlogp_dy;
            const double grad = tgt_grad(j, Gd, j < 2 ? eps1 : eps2, exp((double)pj));
            adam_update(pj, mj, vj, (float)grad, ak);
            if (act) {
                coh_st<COH>(P + t, pj);
                coh_st<COH>(M + t, mj);
                coh_st<COH>(V + t, vj);
            }
            // PREP of the next step: the draw (Philox keyed by the global target index and the step)
            const float2 nrm = normal2_at(c.seed, ((unsigned long long)kSiteTarget << 48) + (unsigned long long)(c.t_off + tc),
                                          s_prep * 4ull);
            const double en = j < 2 ? (double)nrm.x : (double)nrm.y;
            const float p_scale = __shfl_xor(pj, 1, 64);  // even lanes: the updated log scale of their pair
            const double val = tgt_draw(pj, en, p_scale);
            if (act && (j & 1) == 0) {
                coh_st<COH>((j == 0 ? c.eps_mu : c.eps_sd) + t, en);
                coh_st<COH>((j == 0 ? c.mu_t : c.y_t) + t, val);
                (j == 0 ? hmu : hy)[t - ta] = val;
            }
        }
    }
    __syncthreads();
    BEAN_ASYNC_TF(1);
    // ---- phase C: the Phi tables of the new draws (one lane per distinct finite bin edge, then one per (target, bin))
    {
#pragma clang fp contract(off)
        const int nue = c.ue_idx[2 * B];
        const int nu1 = nue > 0 ? nue : 1, per = 64 / nu1;
        double* const ecdf = tabs + 128;
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
            const int cnt = (tb - base + 1 < per ? tb - base + 1 : per) * B;
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
                coh_st<COH>(c.tabP + o, ch - cl);
                coh_st<COH>(c.tabPmu + o, -(fh - fl) * inv);
                coh_st<COH>(c.tabPy + o, -(uh - ul) * inv * dsig_dy);
            }
            __syncthreads();
        }
    }
    }
    BEAN_ASYNC_TF(2);
    // ---- the tile's guides: alpha_pi (and the accessibility noise site), tables for the next step
    if (MIX && (part & 2)) {
        const int g = tile * 64 + lane - c.g_sh;
        double lg = 0.0;
        if (g >= 0 && g < G) {
            param_guide_mix<true, true, true, COH>(c, g, ak, s_prep, lg);
            loss_term(lg);
        }
    }
    BEAN_ASYNC_TF(3);
    // ---- loss: the R waves' parts of this tile + this wave's prior / entropy terms, integer atomics
    // the R waves' loss parts of this tile (with the targets): lane r asks for replicate r's three words - one round trip
    // for the wave - and the words join this wave's own terms in the integer sums below (integers: any order).  (One lane
    // asking for all of them had become 3 R loads in a row, each waited for before the next: the compiler keeps a
    // uniform lane's values in scalar registers, and a readfirstlane waits for its load.)
    if (part & 1) {
        for (int r0 = 0; r0 < R; r0 += 64) {
            if (r0 + lane < R) {
                const long long* o = c.lpart + 3 * (long)(((tile >> 3) * R + r0 + lane) * 8 + (tile & 7));
                loss_hi += __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                loss_lo += __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                loss_bad += __hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    loss_hi = wave_sum_i64(loss_hi);
    loss_lo = wave_sum_i64(loss_lo);
    loss_bad = wave_sum_i64(loss_bad);
    if (lane == 0) {
        long long a = 0, b = 0, d = 0;
        a += loss_hi;
        b += loss_lo;
        d += loss_bad;
        long long* acc = c.loss_acc + ((long)ctr.slot * kLossSub + (tile & (kLossSub - 1))) * kLossWords;
        atomicAdd((unsigned long long*)acc, (unsigned long long)a);
        atomicAdd((unsigned long long*)acc + 1, (unsigned long long)b);
        if (d) atomicAdd((unsigned long long*)acc + 2, (unsigned long long)d);
    }
}

// One item: k_guide_wave2's wave work on (tile, r) at the step, this wave's loss part, the arrival.  Returns
// {t0, number of the tile's targets} when this wave completed the tile (it then finishes it), {-1, 0} otherwise.
template <int FAM, bool ACC>
__device__ BEAN_ASYNC_INLINE int2 async_guide_item(const DevArgs* cp, unsigned long long step, unsigned long long slot,
                                              float step_size, int tile, int r BEAN_ASYNC_ST_ARG) {
    const DevArgs c = dev_args_in_sgprs(cp);
    tile = rfl_i(tile);
    r = rfl_i(r);
    StepCtr ctr;
    ctr.step = rfl_u64(step);
    ctr.slot = rfl_u64(slot);
    ctr.step_size = __builtin_bit_cast(float, rfl_i(__builtin_bit_cast(int, step_size)));
    ctr.pad_ = 0.f;
    const int lane = threadIdx.x, R = c.R;
    const int wg = ((tile >> 3) * R + r) * 8 + (tile & 7);  // the wave's id in the grid of padded tiles x replicates
    int t0, nt;
    double tot;
    guide_wave2_tile<FAM, ACC, 2>(c, ctr, tile, r, wg, t0, nt, tot);
    BEAN_ASYNC_T(4);
    if (lane == 0) {
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
    }
    // ---- arrival: every row / sum / loss-part store of this wave has completed before it is counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) {
        // (the counter runs on through the call - k_async_head zeroed it - and is never reset: the arrivals of a step
        // cannot begin before all R of the step before have been counted, so a multiple of R closes a step, and there is
        // no store whose order against a later step's atomics would matter)
        const int old = __hip_atomic_fetch_add(c.tile_ctr + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old % R == R - 1) last = 1;
    }
    last = __builtin_amdgcn_readfirstlane(last);
    (void)t0;
    (void)nt;
    int2 res;
    res.x = last ? 1 : -1;
    res.y = 0;
    return res;
}

template <int FAM, bool ACC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BEAN_ASYNC_EU)))
void k_svi_async(const DevArgs* cp, int R, int n_tiles, AsyncArgs a) {
    const int lane = threadIdx.x;
    const int x = blockIdx.x & 7;               // XCD group of this wave and of the tiles it works on
    const int nx = (n_tiles - x + 7) >> 3;      // tiles k with k & 7 == x
    const int per_step = nx * R;
    const long total = (long)per_step * a.n_steps;
    const long total_fin = (a.fin_split ? 2l : 1l) * nx * a.n_steps;  // (ring entries per tile and step)
    int* const queue = a.queue + x * kAsyncQueueStride;
    const bool roles = a.n_guide_blocks > 0;
    bool finisher = roles && (int)blockIdx.x >= a.n_guide_blocks;
    int* const fhead = a.fhead + x * kAsyncQueueStride;
    int* const ftail = a.ftail + x * kAsyncQueueStride;
    int* const fring = a.fring + x * a.fring_stride;
    int item = -1;  // a guide item this wave has pulled and not run yet
    for (;;) {
        int fin_s = -1, fin_tile = 0, fin_part = 3;  // the finish this iteration ends with, if any
#ifdef BEAN_ASYNC_STAMP
        unsigned long long* st_row = nullptr;
#endif
        if (finisher) {
            // ---- a finisher takes the group's next finish and waits until it is there
            int pos = 0;
            if (lane == 0) pos = __hip_atomic_fetch_add(fhead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pos = __builtin_amdgcn_readfirstlane(pos);
            if (pos >= total_fin) return;
            int entry = 0, spins = 0;
            for (;;) {
                int v = 0, ab = 0;
                if (lane == 0) v = __hip_atomic_load(fring + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 1) ab = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__any(ab != 0)) return;
                entry = __builtin_amdgcn_readfirstlane(v);
                if (entry != 0) break;
                if (++spins > kAsyncSpinMax) {
                    if (lane == 0) async_give_up(cp, a);
                    return;
                }
                __builtin_amdgcn_s_sleep(BEAN_ASYNC_SLEEP);
            }
            asm volatile("" ::: "memory");
            fin_s = (int)((unsigned)entry >> 18) - 1;
            fin_part = ((entry >> 16) & 3) + 1;  // 1 targets, 2 guides, 3 both
            fin_tile = (entry & 0xffff) * 8 + x;
        } else {
            // ---- pull the group's next item: (step, tile, replicate) in that order
            if (item < 0) {
                int it = 0;
                if (lane == 0) it = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                item = __builtin_amdgcn_readfirstlane(it);
            }
            if (item >= total) {
                // no items left: without roles the wave is done; with roles it joins the finishers for what is left in the
                // ring (the last steps' finishes - and all of them, should no finisher be resident)
                if (!roles) return;
                finisher = true;
                continue;
            }
            const int s = item / per_step;
            const int rem = item - s * per_step;
            const int jt = rem / R;
            const int r = rem - jt * R;
            const int tile = jt * 8 + x;
#ifdef BEAN_ASYNC_STAMP
            if (a.stamps && s >= kAsyncStampStep0 && s < kAsyncStampStep0 + kAsyncStampSteps) {
                const long all = (long)((n_tiles + 7) / 8 * 8) * R;
                st_row = a.stamps + ((long)(s - kAsyncStampStep0) * all + (long)tile * R + r) * 8;
                if (lane == 0) {
                    st_row[5] = (unsigned long long)blockIdx.x;
                    st_row[6] = (unsigned long long)tile;
                    st_row[7] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
                }
            }
#endif
            BEAN_ASYNC_T(0);
            // ---- wait until the previous step of this tile and of its two neighbours (a target may straddle a tile
            // boundary, and is finished by whichever of the two tiles completes second) has been finished: lanes 0 - 2
            // poll one word each, lane 3 the abort word.  With finisher roles a wave that has waited for a while looks
            // into the finish ring and takes the oldest finish that nobody has taken (the finishers need not be
            // resident for the launch to make progress).
            bool ready = false;
            {
                // (done[2 k]: tile k's targets, done[2 k + 1]: its guides; a neighbour matters through the targets it shares)
                const int tq = (lane == 0 || lane == 3) ? tile : (lane == 1 ? tile - 1 : tile + 1);
                const bool need = s > 0 && lane < 4 && tq >= 0 && tq < n_tiles;
                int spins = 0;
                for (;;) {
                    int v = s;
                    if (need) v = __hip_atomic_load(a.done + 2 * tq + (lane == 3 ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int ab = 0;
                    if (lane == 4) ab = __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__any(ab != 0)) return;
                    if (__all(v >= s)) {
                        ready = true;
                        break;
                    }
                    if (++spins > kAsyncSpinMax) {
                        if (lane == 0) async_give_up(cp, a);
                        return;
                    }
                    if (roles && (spins & (kAsyncStealEvery - 1)) == 0) {
                        int e = 0;
                        if (lane == 0) {
                            const int h = __hip_atomic_load(fhead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (h < total_fin) {
                                const int cand = __hip_atomic_load(fring + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                int expect = h;
                                if (cand != 0 && __hip_atomic_compare_exchange_strong(fhead, &expect, h + 1, __ATOMIC_RELAXED,
                                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                    e = cand;
                            }
                        }
                        e = __builtin_amdgcn_readfirstlane(e);
                        if (e != 0) {
                            fin_s = (int)((unsigned)e >> 18) - 1;
                            fin_part = ((e >> 16) & 3) + 1;
                            fin_tile = (e & 0xffff) * 8 + x;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(BEAN_ASYNC_SLEEP);
                }
                asm volatile("" ::: "memory");  // (nothing below is loaded before the poll has matched)
            }
            BEAN_ASYNC_T(1);
            if (ready) {
                const unsigned long long step = a.step0 + (unsigned long long)s, slot = a.slot0 + (unsigned long long)s;
                const int2 fin = async_guide_item<FAM, ACC>(cp, step, slot, a.step_sizes[s], tile, r BEAN_ASYNC_ST_PASS);
                item = -1;
                BEAN_ASYNC_T(2);
                if (fin.x >= 0) {
                    if (roles) {
                        // the tile is complete: hand its finish to the group's finishers (the rows, sums and loss parts of
                        // all R waves had completed before the arrival that made this wave the last)
                        // (two entries: the targets' part and the guides' part run on two finishers at the same time)
                        if (lane == 0) {
                            const int e = (int)(((unsigned)(s + 1) << 18) | (unsigned)(tile >> 3));
                            if (a.fin_split) {
                                const int p = __hip_atomic_fetch_add(ftail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(fring + p, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                // targets
                                __hip_atomic_store(fring + p + 1, e | 0x10000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // guides
                            } else {
                                const int p = __hip_atomic_fetch_add(ftail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(fring + p, e | 0x20000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // both
                            }
                        }
                    } else {
                        fin_s = s;
                        fin_tile = tile;
                        fin_part = 3;
                    }
                }
            }
        }
        if (fin_s >= 0) {
#ifdef BEAN_ASYNC_STAMP
            // (a finish stamps the row of its tile's replicate 0, whoever runs it: words 4 / 5 of the finish table = start /
            // published of the targets' part - or of the whole finish - 6 / 7 of the guides' part; words 0 - 3 its phases)
            st_row = nullptr;
            if (a.stamps && fin_s >= kAsyncStampStep0 && fin_s < kAsyncStampStep0 + kAsyncStampSteps)
                st_row = a.stamps + ((long)(fin_s - kAsyncStampStep0) * ((long)((n_tiles + 7) / 8 * 8) * R) + (long)fin_tile * R) * 8;
            BEAN_ASYNC_TF(fin_part == 2 ? 6 : 4);
#endif
            const unsigned long long step = a.step0 + (unsigned long long)fin_s, slot = a.slot0 + (unsigned long long)fin_s;
            // (the tile's next step waits for this chain: it goes first on its SIMD)
            if (BEAN_ASYNC_PRIO) __builtin_amdgcn_s_setprio(3);
            async_finish_tile<FAM, ACC>(cp, step, slot, a.step_sizes[fin_s], fin_tile, fin_part BEAN_ASYNC_ST_PASS);
            if (BEAN_ASYNC_PRIO) __builtin_amdgcn_s_setprio(0);
            // ---- publish: every store of the finish has completed before the tile's step counts move
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                if (fin_part & 1) __hip_atomic_store(a.done + 2 * fin_tile, fin_s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (fin_part & 2) __hip_atomic_store(a.done + 2 * fin_tile + 1, fin_s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            BEAN_ASYNC_TF(fin_part == 2 ? 7 : 5);
        }
        __syncthreads();  // (the next item restages the wave's LDS)
    }
}

}  // namespace bean

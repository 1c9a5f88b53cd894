// Device special functions and samplers for the BEAN ELBO kernels (gfx950).
//
// All arithmetic is float64: the reference's likelihood runs in float64 through
// dtype promotion and MI355X issues v_fma_f64 at the same rate as
// non-packed f32.
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>

#ifndef BEAN_NOINLINE
#define BEAN_NOINLINE __noinline__
#endif

namespace bean {

struct DD {
    double d;   // lgamma(a + x) - lgamma(a)
    double dp;  // digamma(a + x) - digamma(a)
};

// ---------------------------------------------------------------------------
// Lean float64 primitives.  The ocml log/division are correctly rounded but cost
// ~100 / ~12 instructions (double-double arithmetic); the ELBO needs ~1e-15
// relative accuracy, which these reach in ~28 / 5 instructions.  Arguments are
// positive finite normal numbers everywhere they are used.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double frcp(double x) {
    const double r = __builtin_amdgcn_rcp(x);  // v_rcp_f64: ~2^-26 relative
    // one third-order step, 1 / x = r (1 + e + e^2 + ...) with e = 1 - x r: the remainder e^3 is 2^-78
    // (three FMAs; two Newton steps took four).  Round 3 backed this out because the A/B library's fused
    // step kernel faulted with it: that was the toolchain's misplaced live-range copy (isa_check.py), which
    // this change happened to provoke, not this arithmetic.
    const double e = fma(-x, r, 1.0);
    return fma(fma(e, e, e), r, r);
}

// natural log of a positive normal double: x = m 2^e, m in [sqrt(1/2), sqrt(2)),
// log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716, series to s^19.
// In two halves, so that a caller with several logs (and other reciprocals) to form can take all its
// reciprocals from ONE v_rcp_f64 (batched inversion: stirling_diff): flog_arg splits x = (1 + f) 2^e, and
// flog_from finishes the log given 1 / (2 + f).
struct LogArg {
    double f;   // m - 1, m in [sqrt(1/2), sqrt(2))
    double ed;  // the exponent e
};
__device__ __forceinline__ LogArg flog_arg(double x) {
#pragma clang fp contract(off)  // every rounding pinned: the same bits wherever this is inlined
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    LogArg a;
    a.f = m - 1.0;
    a.ed = (double)e;
    return a;
}
__device__ __forceinline__ double flog_from(const LogArg& a, double rinv) {  // rinv = 1 / (2 + a.f)
#pragma clang fp contract(off)
    const double s = a.f * rinv;
    const double z = s * s;
    double p = 2.0 / 19.0;
    p = fma(p, z, 2.0 / 17.0);
    p = fma(p, z, 2.0 / 15.0);
    p = fma(p, z, 2.0 / 13.0);
    p = fma(p, z, 2.0 / 11.0);
    p = fma(p, z, 2.0 / 9.0);
    p = fma(p, z, 2.0 / 7.0);
    p = fma(p, z, 2.0 / 5.0);
    p = fma(p, z, 2.0 / 3.0);
    // e*ln2 split so that the leading product is exact for |e| < 2^11
    const double hi = a.ed * 6.93147180369123816490e-01;
    const double lo2 = fma(a.ed, 1.90821492927058770002e-10, s * z * p);
    return hi + fma(2.0, s, lo2);
}
__device__ __forceinline__ double flog(double x) {
#pragma clang fp contract(off)
    const LogArg a = flog_arg(x);
    return flog_from(a, frcp(2.0 + a.f));
}

// Stirling tails, valid to ~1e-14 absolute for z >= 10 (r = 1/z, w = r*r).
__device__ __forceinline__ double stirling_lgamma_tail(double r, double w) {
    // 1/(12 z) - 1/(360 z^3) + 1/(1260 z^5) - 1/(1680 z^7) + 1/(1188 z^9)
    double t = 8.4175084175084175e-4;
    t = fma(w, t, -5.9523809523809524e-4);
    t = fma(w, t, 7.9365079365079365e-4);
    t = fma(w, t, -2.7777777777777778e-3);
    t = fma(w, t, 8.3333333333333333e-2);
    return r * t;
}
__device__ __forceinline__ double stirling_digamma_tail(double w) {
    // 1/(12 z^2) - 1/(120 z^4) + 1/(252 z^6) - 1/(240 z^8) + 1/(132 z^10) - 691/(32760 z^12)
    double t = -2.1092796092796093e-2;
    t = fma(w, t, 7.5757575757575758e-3);
    t = fma(w, t, -4.1666666666666667e-3);
    t = fma(w, t, 3.9682539682539683e-3);
    t = fma(w, t, -8.3333333333333333e-3);
    t = fma(w, t, 8.3333333333333333e-2);
    return w * t;
}

// Longer Stirling tails, valid to ~1e-14 / 3e-14 absolute for z >= 6 (terms through B16): used for
// the small argument of a difference so that its shift loop (divergent, and followed by a log and a
// division) is only needed below 6 instead of below 10.
__device__ __forceinline__ double stirling_lgamma_tail_long(double r, double w) {
    double t = -2.9550653594771242e-2;
    t = fma(w, t, 6.4102564102564103e-3);
    t = fma(w, t, -1.9175269175269175e-3);
    t = fma(w, t, 8.4175084175084175e-4);
    t = fma(w, t, -5.9523809523809524e-4);
    t = fma(w, t, 7.9365079365079365e-4);
    t = fma(w, t, -2.7777777777777778e-3);
    t = fma(w, t, 8.3333333333333333e-2);
    return r * t;
}
__device__ __forceinline__ double stirling_digamma_tail_long(double w) {
    double t = -4.4325980392156863e-1;
    t = fma(w, t, 8.3333333333333333e-2);
    t = fma(w, t, -2.1092796092796093e-2);
    t = fma(w, t, 7.5757575757575758e-3);
    t = fma(w, t, -4.1666666666666667e-3);
    t = fma(w, t, 3.9682539682539683e-3);
    t = fma(w, t, -8.3333333333333333e-3);
    t = fma(w, t, 8.3333333333333333e-2);
    return w * t;
}

constexpr double kShift = 10.0;
constexpr double kShiftLo = 6.0;
constexpr double kHalfLog2Pi = 0.91893853320467274178;

// Raise z to >= kShift, accumulating P = prod(z+i) and Q = dP/dz, so that
// lgamma(z) = lgamma(z') - log P and digamma(z) = digamma(z') - Q / P.
__device__ __forceinline__ void shift_up(double& z, double& P, double& Q, double zmin = kShift) {
    P = 1.0;
    Q = 0.0;
    while (z < zmin) {
        Q = fma(Q, z, P);
        P *= z;
        z += 1.0;
    }
}

// lgamma(z) and digamma(z) for z > 0.
__device__ __forceinline__ void lgamma_digamma(double z, double& lg, double& dg) {
    double P, Q;
    shift_up(z, P, Q);
    const double l = flog(z), r = frcp(z), w = r * r;
    lg = (z - 0.5) * l - z + kHalfLog2Pi + stirling_lgamma_tail(r, w);
    dg = l - 0.5 * r - stirling_digamma_tail(w);
    if (P != 1.0) {
        lg -= flog(P);
        dg -= Q * frcp(P);
    }
}

__device__ __forceinline__ double digamma(double z) {
    double lg, dg;
    lgamma_digamma(z, lg, dg);
    return dg;
}

// The series part of D(a, x) for shifted arguments z1 >= kShiftLo (from a, shift product P1 and its
// derivative Q1) and z2 >= kShift: roundings pinned.  With P1 == 1 (no shift) log P1 = 0 and Q1 = 0 exactly.
// It needs 1 / z and log z of three numbers (z1, z2, P1), and the log has a reciprocal of its own inside;
// a v_rcp_f64 costs 3.3 float64 FMAs of issue time, 7 with its two Newton steps
// (profiles/r01_valu_issue.txt).  log_and_rcp takes both reciprocals from one: 1 / z = d / (z d),
// 1 / d = z / (z d) with d = 2 + f in [1.7, 2.42] (three multiplications + one frcp instead of two frcp).
// (All six from ONE reciprocal - 17 multiplications - was measured too: the twelve values it keeps live
// cost more in spilled registers than the reciprocals it saves.)
struct LogRcp {
    double l, r;  // log z, 1 / z
};
__device__ __forceinline__ LogRcp log_and_rcp(double z) {
#pragma clang fp contract(off)
    const LogArg a = flog_arg(z);
    const double d = 2.0 + a.f;
    const double inv = frcp(z * d);
    LogRcp o;
    o.r = inv * d;
    o.l = flog_from(a, inv * z);
    return o;
}
__device__ __forceinline__ void stirling_diff(double z1, double z2, double P1, double Q1, double& d, double& dp) {
#pragma clang fp contract(off)
    const LogRcp k1 = log_and_rcp(z1), k2 = log_and_rcp(z2);
    const double l1 = k1.l, l2 = k2.l, r1 = k1.r, r2 = k2.r;
    const double w1 = r1 * r1, w2 = r2 * r2;
    const double head = (z2 - 0.5) * l2 - (z1 - 0.5) * l1 - (z2 - z1);
    d = head + (stirling_lgamma_tail(r2, w2) - stirling_lgamma_tail_long(r1, w1));
    dp = (l2 - l1) - 0.5 * (r2 - r1) - (stirling_digamma_tail(w2) - stirling_digamma_tail_long(w1));
    if (P1 != 1.0) {
        const LogRcp kP = log_and_rcp(P1);
        d += kP.l;
        dp += Q1 * kP.r;
    }
}
// two at once: the same operations, written side by side so that the scheduler interleaves the chains
__device__ __forceinline__ void stirling_diff2(double z1a, double z2a, double P1a, double Q1a, double z1b, double z2b,
                                               double P1b, double Q1b, double& da, double& dpa, double& db,
                                               double& dpb) {
#pragma clang fp contract(off)
    const LogRcp k1a = log_and_rcp(z1a), k1b = log_and_rcp(z1b), k2a = log_and_rcp(z2a), k2b = log_and_rcp(z2b);
    const double l1a = k1a.l, l1b = k1b.l, l2a = k2a.l, l2b = k2b.l;
    const double r1a = k1a.r, r1b = k1b.r, r2a = k2a.r, r2b = k2b.r;
    const double w1a = r1a * r1a, w1b = r1b * r1b, w2a = r2a * r2a, w2b = r2b * r2b;
    const double heada = (z2a - 0.5) * l2a - (z1a - 0.5) * l1a - (z2a - z1a);
    const double headb = (z2b - 0.5) * l2b - (z1b - 0.5) * l1b - (z2b - z1b);
    da = heada + (stirling_lgamma_tail(r2a, w2a) - stirling_lgamma_tail_long(r1a, w1a));
    db = headb + (stirling_lgamma_tail(r2b, w2b) - stirling_lgamma_tail_long(r1b, w1b));
    dpa = (l2a - l1a) - 0.5 * (r2a - r1a) - (stirling_digamma_tail(w2a) - stirling_digamma_tail_long(w1a));
    dpb = (l2b - l1b) - 0.5 * (r2b - r1b) - (stirling_digamma_tail(w2b) - stirling_digamma_tail_long(w1b));
    if (__any(P1a != 1.0 || P1b != 1.0)) {  // (adds exact zeros where nothing was shifted)
        if (P1a != 1.0) {
            const LogRcp kP = log_and_rcp(P1a);
            da += kP.l;
            dpa += Q1a * kP.r;
        }
        if (P1b != 1.0) {
            const LogRcp kP = log_and_rcp(P1b);
            db += kP.l;
            dpb += Q1b * kP.r;
        }
    }
}

// D(a, x) = lgamma(a + x) - lgamma(a) and its derivative in a, for a > 0 and
// x >= 0.  Counts are integer-valued, so small x uses the exact product form
// prod_{i<x}(a + i); everything else is a difference of Stirling series, with
// both arguments first raised to >= kShift.
__device__ __forceinline__ DD lgamma_digamma_diff_inl(double a, double x) {
#pragma clang fp contract(off)
    DD out;
    if (x == 0.0) {
        out.d = 0.0;
        out.dp = 0.0;
        return out;
    }
    if (x <= 12.0 && x == floor(x)) {
        double P = 1.0, Q = 0.0, t = a;
        const int n = (int)x;
        for (int i = 0; i < n; ++i) {
            Q = fma(Q, t, P);
            P *= t;
            t += 1.0;
        }
        out.d = flog(P);
        out.dp = Q * frcp(P);
        return out;
    }
    double z1 = a, z2 = a + x, P1, Q1, P2, Q2;
    shift_up(z1, P1, Q1, kShiftLo);  // the concentration: longer tails instead of a longer shift
    shift_up(z2, P2, Q2);
    stirling_diff(z1, z2, P1, Q1, out.d, out.dp);
    if (P2 != 1.0) {
        out.d -= flog(P2);
        out.dp -= Q2 * frcp(P2);
    }
    return out;
}

// Two differences at once, as two INDEPENDENT dependency chains in one straight line of code: a wave
// alone on its SIMD issues a float64 instruction every ~11 cycles along one chain, and a loop over bins
// that evaluates one difference per iteration is one chain.  Same operations per chain as
// lgamma_digamma_diff (bit-identical results); arguments with x <= 12 (product form, or a second
// argument that still needs shifting) send the whole wave through the one-chain function.
struct DD2 {
    DD a, b;
};
__device__ BEAN_NOINLINE DD lgamma_digamma_diff(double a, double x);
__device__ __forceinline__ DD2 lgamma_digamma_diff2(double a0, double x0, double a1, double x1) {
    DD2 out;
    if (__any(x0 <= 12.0 || x1 <= 12.0)) {
        out.a = lgamma_digamma_diff(a0, x0);
        out.b = lgamma_digamma_diff(a1, x1);
        return out;
    }
    double z1a = a0, z1b = a1, P1a = 1.0, Q1a = 0.0, P1b = 1.0, Q1b = 0.0;
    // a wave-uniform loop (the lanes that are done take neither `if`: same values as a per-lane loop): the
    // wave runs for its slowest lane either way, and a scalar branch leaves no exec mask to restore - the
    // join of the per-lane form is where the toolchain once misplaced a copy (isa_check.py)
    while (__any(z1a < kShiftLo || z1b < kShiftLo)) {
        if (z1a < kShiftLo) {
            Q1a = fma(Q1a, z1a, P1a);
            P1a *= z1a;
            z1a += 1.0;
        }
        if (z1b < kShiftLo) {
            Q1b = fma(Q1b, z1b, P1b);
            P1b *= z1b;
            z1b += 1.0;
        }
    }
    // a + x > 12 >= kShift: the second arguments need no shift
    stirling_diff2(z1a, a0 + x0, P1a, Q1a, z1b, a1 + x1, P1b, Q1b, out.a.d, out.a.dp, out.b.d, out.b.dp);
    return out;
}

// out-of-line copy for the kernels that call it from many sites
__device__ BEAN_NOINLINE DD lgamma_digamma_diff(double a, double x) { return lgamma_digamma_diff_inl(a, x); }

// Standard normal cdf / pdf as torch.distributions.Normal computes them
// (torch/distributions/normal.py:105-113): 0.5 * (1 + erf(u / sqrt 2)).
__device__ __forceinline__ double norm_cdf(double u) {
    return 0.5 * (1.0 + erf(u * 0.70710678118654752440));
}
__device__ __forceinline__ double norm_pdf(double u) {
    return 0.39894228040143267794 * exp(-0.5 * u * u);
}

// ---------------------------------------------------------------------------
// Implicit reparameterisation gradient of a Dirichlet component,
// -(d cdf/d alpha) / pdf / (1 - x) for x ~ Beta(alpha, total - alpha): the
// piecewise approximation published in torch (ATen/native/Distributions.h,
// dirichlet_grad_one and helpers; torch 2.10).  The reference reaches it through
// pyro.distributions.Dirichlet.rsample -> torch._dirichlet_grad; parity with the
// reference's alpha_pi gradient requires the same approximation, so the region
// boundaries and fitted coefficients below are torch's.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double fsqrt(double x) { return sqrt(x); }

// x^y for x > 0 through exp(y log x) (used only on rarely taken paths)
__device__ __forceinline__ double fpow(double x, double y) { return exp(y * flog(x)); }

// PRE: digamma(alpha) and digamma(alpha + beta) are supplied by the caller (they depend on the
// concentrations only, which k_param has already evaluated for the guide: see DevArgs::dgq) instead
// of being evaluated here - two of them per call were ~11 % of k_guide_wave2's instructions.
template <bool PRE>
__device__ __forceinline__ double beta_grad_alpha_small_t(double x, double alpha, double beta, double dg_a,
                                                          double dg_ab) {
    const double factor = (PRE ? dg_a - dg_ab : digamma(alpha) - digamma(alpha + beta)) - flog(x);
    const double ra = frcp(alpha);
    double numer = 1.0;
    double series = numer * ra * (factor + ra);
    // (Measured, not adopted - round 5: skipping the loop and the power where EVERY active lane holds a draw on the floor,
    // x < 1e-280, i.e. the slots above a wave's alleles once the guides are ordered by allele count.  No bit changes - the
    // further terms are below 1e-270 of the first, (1 - x)^-beta is exp(-beta log 1) = 1 - and no time either: at
    // BASELINE config 3 a masked allele's concentration is ~1e-4 (pi_a0 / sum alpha ~ 10), 7 % of its draws stay above
    // the floor, and a wave of 64 such lanes is all-floor once in a hundred calls.)
    for (int i = 1; i <= 10; ++i) {
        const double ci = (double)i;
        numer *= (ci - beta) * x * frcp(ci);
        const double rd = frcp(alpha + ci);
        series += numer * rd * (factor + rd);
    }
    const double result = x * fpow(1.0 - x, -beta) * series;
    return isnan(result) ? 0.0 : result;
}

template <bool PRE>
__device__ __forceinline__ double beta_grad_beta_small_t(double x, double alpha, double beta, double dg_ab,
                                                         double dg_b) {
    const double factor = PRE ? dg_ab - dg_b : digamma(alpha + beta) - digamma(beta);
    double numer = 1.0, betas = 1.0, dbetas = 0.0, series = factor * frcp(alpha);
    for (int i = 1; i <= 8; ++i) {
        const double ci = (double)i;
        numer *= -x * frcp(ci);
        dbetas = dbetas * (beta - ci) + betas;
        betas = betas * (beta - ci);
        series += numer * frcp(alpha + ci) * (dbetas + factor * betas);
    }
    const double result = -fpow(1.0 - x, 1.0 - beta) * series;
    return isnan(result) ? 0.0 : result;
}

__device__ inline double beta_grad_alpha_mid(double x, double alpha, double beta) {
    const double total = alpha + beta;
    const double rtot = frcp(total);
    const double mean = alpha * rtot;
    const double sd = fsqrt(alpha * beta * frcp(total + 1.0)) * rtot;
    if (mean - 0.1 * sd <= x && x <= mean + 0.1 * sd) {
        const double b2 = beta * beta;
        const double poly =
            47.0 * x * b2 * b2 +
            alpha * ((43.0 + 20.0 * (16.0 + 27.0 * beta) * x) * b2 * beta +
                     alpha * (3.0 * (59.0 + 180.0 * beta - 90.0 * x) * b2 +
                              alpha * ((453.0 + 1620.0 * beta * (1.0 - x) - 455.0 * x) * beta +
                                       alpha * (8.0 * (1.0 - x) * (135.0 * beta - 11.0)))));
        const double pre_num = (1.0 + 12.0 * alpha) * (1.0 + 12.0 * beta) * rtot * rtot;
        const double pre_den =
            12960.0 * alpha * alpha * alpha * beta * beta * (1.0 + 12.0 * total);
        return pre_num * poly * frcp((1.0 - x) * pre_den);
    }
    const double ra = frcp(alpha), rb = frcp(beta);
    const double prefactor = -x * frcp(fsqrt(2.0 * alpha * beta * rtot));
    const double stirling = (1.0 + ra * (1.0 / 12.0) + ra * ra * (1.0 / 288.0)) *
                            (1.0 + rb * (1.0 / 12.0) + rb * rb * (1.0 / 288.0)) *
                            frcp(1.0 + rtot * (1.0 / 12.0) + rtot * rtot * (1.0 / 288.0));
    const double term1_num =
        2.0 * (alpha * alpha) * (x - 1.0) + alpha * beta * (x - 1.0) - x * (beta * beta);
    const double axbx = alpha * (x - 1.0) + beta * x;
    const double term1_den = fsqrt(2.0 * alpha * rb) * (total * fsqrt(total)) * axbx * axbx;
    const double term1 = term1_num * frcp(term1_den);
    const double la = flog(alpha * frcp(total * x));
    const double lb = flog(beta * frcp(total * (1.0 - x)));
    const double term2 = 0.5 * la;
    const double term3 = fsqrt(8.0 * alpha * beta * rtot) * frcp(axbx);
    const double term4_base = beta * lb + alpha * la;
    const double term4 = frcp(term4_base * fsqrt(term4_base));
    const double term1234 = term1 + term2 * (term3 + (x < mean ? term4 : -term4));
    return stirling * prefactor * term1234;
}

// c[num/den][u^i][a^j][b^k] of the rational correction in dirichlet_grad_one
__device__ __constant__ const double kDirGradC[2][3][3][4] = {
    {{{1.003668233, -0.01061107488, -0.0657888334, 0.01201642863},
      {0.6336835991, -0.3557432599, 0.05486251648, -0.001465281033},
      {-0.03276231906, 0.004474107445, 0.002429354597, -0.0001557569013}},
     {{0.221950385, -0.3187676331, 0.01799915743, 0.01074823814},
      {-0.2951249643, 0.06219954479, 0.01535556598, 0.001550077057},
      {0.02155310298, 0.004170831599, 0.001292462449, 6.976601077e-05}},
     {{-0.05980841433, 0.008441916499, 0.01085618172, 0.002319392565},
      {0.02911413504, 0.01400243777, -0.002721828457, 0.000751041181},
      {0.005900514878, -0.001936558688, -9.495446725e-06, 5.385558597e-05}}},
    {{{1, -0.02924021934, -0.04438342661, 0.007285809825},
      {0.6357567472, -0.3473456711, 0.05454656494, -0.002407477521},
      {-0.03301322327, 0.004845219414, 0.00231480583, -0.0002307248149}},
     {{0.5925320577, -0.1757678135, 0.01505928619, 0.000564515273},
      {0.1014815858, -0.06589186703, 0.01272886114, -0.0007316646956},
      {-0.007258481865, 0.001096195486, 0.0003934994223, -4.12701925e-05}},
     {{0.06469649321, -0.0236701437, 0.002902096474, -5.896963079e-05},
      {0.001925008108, -0.002869809258, 0.0008000589141, -6.063713228e-05},
      {-0.0003477407336, 6.959756487e-05, 1.097287507e-05, -1.650964693e-06}}},
};

template <bool PRE>
__device__ __forceinline__ double dirichlet_grad_one_t(double x, double alpha, double total, double dg_alpha,
                                                       double dg_total) {
    const double beta = total - alpha;
    const double boundary = total * x * (1.0 - x);
    if (x <= 0.5 && boundary < 2.5) return beta_grad_alpha_small_t<PRE>(x, alpha, beta, dg_alpha, dg_total);
    if (x >= 0.5 && boundary < 0.75) return -beta_grad_beta_small_t<PRE>(1.0 - x, beta, alpha, dg_total, dg_alpha);
    if (alpha > 6.0 && beta > 6.0) return beta_grad_alpha_mid(x, alpha, beta);
    // rational correction to an analytic approximation (kDirGradC below)
    const auto& c = kDirGradC;
    const double u = flog(x);
    const double a = flog(alpha) - u;
    const double b = flog(total) - a;
    const double pow_u[3] = {1.0, u, u * u};
    const double pow_a[3] = {1.0, a, a * a};
    double p = 0.0, q = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double ua = pow_u[i] * pow_a[j];
            p += ua * (c[0][i][j][0] + b * (c[0][i][j][1] + b * (c[0][i][j][2] + b * c[0][i][j][3])));
            q += ua * (c[1][i][j][0] + b * (c[1][i][j][1] + b * (c[1][i][j][2] + b * c[1][i][j][3])));
        }
    }
    const double dd = PRE ? dg_total - dg_alpha : digamma(total) - digamma(alpha);
    const double approx = x * dd * frcp(beta);
    return p * approx * frcp(q);
}
__device__ __forceinline__ double dirichlet_grad_one_inl(double x, double alpha, double total) {
    return dirichlet_grad_one_t<false>(x, alpha, total, 0.0, 0.0);
}
// out-of-line copies for the kernels that call them from several sites
__device__ BEAN_NOINLINE double dirichlet_grad_one(double x, double alpha, double total) {
    return dirichlet_grad_one_inl(x, alpha, total);
}
// with digamma(alpha) and digamma(total) supplied by the caller
__device__ BEAN_NOINLINE double dirichlet_grad_one_pre(double x, double alpha, double total, double dg_alpha,
                                                       double dg_total) {
    return dirichlet_grad_one_t<true>(x, alpha, total, dg_alpha, dg_total);
}

// ---------------------------------------------------------------------------
// Counter-based RNG: rocRAND Philox4x32-10 device API, keyed by
// (seed, site, element) through the subsequence and by the SVI step through the
// offset, so draws do not depend on grid shape or on the number of GPUs.
// ---------------------------------------------------------------------------
enum RngSite : unsigned long long { kSiteTarget = 1, kSitePi = 2, kSiteNoise = 3, kSiteAux = 4, kSiteQ0 = 5, kSiteCov = 6 };

// Stateless view of one Philox subsequence: draw k is counter (offset/4 + k), so
// the generator lives in 7 registers and never touches memory.
struct Rng {
    unsigned long long seed, sub, off;
    unsigned int k;
    __device__ __forceinline__ Rng(unsigned long long seed_, RngSite site, unsigned long long element,
                                   unsigned long long offset)
        : seed(seed_), sub(((unsigned long long)site << 48) + element), off(offset), k(0) {}
    // four fresh 32-bit words
    // (0, 1] from 53 random bits
    static __device__ __forceinline__ double to_unit(unsigned int a, unsigned int b) {
        const unsigned long long bits = (((unsigned long long)a << 32) | b) >> 11;
        return ((double)bits + 1.0) * 1.1102230246251565e-16;  // 2^-53
    }
};

struct Pair {
    double a, b;
};

// Philox4x32-10 block at an absolute position; out of line so that its ~35
// registers are not multiplied by inlining into the rejection loop.
// = rocrand4 of a rocrand_state_philox4x32_10 set up by rocrand_init(seed, sub, offset), for an offset that is a
// multiple of four words (every offset of this library is: a block per draw round): the ten Philox rounds on
// counter (offset / 4, subsequence) with key seed (rocrand_philox4x32_10.h; Random123), written out so that no
// generator state object exists - held by address it was the one user of scratch memory in k_param.
__device__ __forceinline__ uint4 philox_block(unsigned long long seed, unsigned long long sub,
                                              unsigned long long offset) {
    const unsigned long long blk = offset >> 2;
    unsigned int c0 = (unsigned int)blk, c1 = (unsigned int)(blk >> 32), c2 = (unsigned int)sub,
                 c3 = (unsigned int)(sub >> 32);
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long m0 = (unsigned long long)0xD2511F53u * c0, m1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(m1 >> 32) ^ c1 ^ k0, n2 = (unsigned int)(m0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned int)m1;
        c3 = (unsigned int)m0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
// What rocrand_normal2 (.x: rocrand_normal) returns from a generator set up by rocrand_init(seed, sub, offset):
// the Box-Muller pair of the first two words of its first Philox block (rocrand_normal.h) - the same values,
// without a generator state held by address (the cached second normal and its flag made that state the one user
// of scratch memory in k_param).
__device__ __forceinline__ float2 normal2_at(unsigned long long seed, unsigned long long sub, unsigned long long offset) {
    const uint4 v = philox_block(seed, sub, offset);
    return rocrand_device::detail::box_muller(v.x, v.y);
}
__device__ __noinline__ uint4 philox_at(unsigned long long seed, unsigned long long sub,
                                        unsigned long long offset) {
    return philox_block(seed, sub, offset);
}

// two standard normals (Box-Muller on two 53-bit uniforms) from one Philox counter
__device__ __noinline__ Pair normal_pair_at(unsigned long long seed, unsigned long long sub,
                                            unsigned long long offset) {
    const uint4 v = philox_at(seed, sub, offset);
    const double u1 = Rng::to_unit(v.x, v.y), u2 = Rng::to_unit(v.z, v.w);
    const double rad = sqrt(-2.0 * flog(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    Pair p;
    p.a = rad * cs;
    p.b = rad * sn;
    return p;
}
__device__ __forceinline__ Pair normal_pair(Rng& rng) {
    const Pair p = normal_pair_at(rng.seed, rng.sub, rng.off + 4ull * rng.k);
    ++rng.k;
    return p;
}
__device__ __forceinline__ Pair uniform_pair(Rng& rng) {
    const uint4 v = philox_at(rng.seed, rng.sub, rng.off + 4ull * rng.k);
    ++rng.k;
    Pair p;
    p.a = Rng::to_unit(v.x, v.y);
    p.b = Rng::to_unit(v.z, v.w);
    return p;
}

// Random inputs of one Marsaglia-Tsang rejection round for two components, from ONE Philox
// counter: words (x, y) give two standard normals by Box-Muller evaluated in float32 (the
// reference draws in float32; a 32-bit radius uniform reaches 6.7 sigma), words (z, w) the two
// acceptance uniforms in (0, 1).
struct GammaRound {
    double na, nb, ua, ub;
};
__device__ __forceinline__ GammaRound gamma_round_from(const uint4 v) {
    const float u1 = ((float)v.x + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)v.y * 2.3283064365386963e-10f;           // [0, 1]
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincospif(2.0f * u2, &sn, &cs);
    GammaRound q;
    q.na = (double)(rad * cs);
    q.nb = (double)(rad * sn);
    q.ua = ((double)v.z + 0.5) * 2.3283064365386963e-10;
    q.ub = ((double)v.w + 0.5) * 2.3283064365386963e-10;
    return q;
}
__device__ __noinline__ GammaRound gamma_round_at(unsigned long long seed, unsigned long long sub,
                                                  unsigned long long offset) {
    return gamma_round_from(philox_block(seed, sub, offset));  // inlined: a leaf function needs no stack frame
}

// log in the Marsaglia-Tsang acceptance test (taken only when the squeeze test fails)
__device__ __forceinline__ double accept_log(double x) { return flog(x); }

struct GammaPair {
    double g0, g1;
    unsigned int k;  // generator position after the draw
};

// Two independent Gamma(alpha_i, 1) draws by Marsaglia & Tsang (2000) with the
// alpha < 1 boost (the method behind torch's sample_gamma), sharing one
// rejection loop: each round costs one Philox counter for both components (gamma_round_at),
// and a wave only iterates until its slowest lane has accepted both.
// `first`: the Philox block at the generator's first counter, where the caller has computed it ahead
// (every lane consumes that block first, as the boost's uniforms or as the inputs of its first round).
// FLOOR32: the caller keeps max((float)g, FLT_MIN) of each draw (a site torch samples in float32).  A
// boosted component whose U^(1/alpha) is below FLT_MIN / 128 gives that floor whatever its
// Marsaglia-Tsang factor turns out to be (the factor is at most d v <= 5/3 (1 + 6.66 / sqrt(6))^3 = 86
// with the 32-bit Box-Muller radius), so it skips the rejection loop: same result, and with
// concentrations of 1 / n_guides most waves skip it altogether.
// FLOOR = 2, the same shortcut for callers that keep max(g, DBL_MIN) (the tiling pi sites: masked alleles are
// components with concentrations of ~1e-6, whose boost underflows for 99.9 % of the draws): a boosted component
// with U^(1/alpha) < e^-716 gives a product below DBL_MIN whatever its Marsaglia-Tsang factor (<= 86) is.
template <int FLOOR = 0>
__device__ __forceinline__ GammaPair sample_gamma_pair_inl(double a0, double a1, Rng rng, const uint4* first = nullptr) {
    constexpr bool FLOOR32 = FLOOR != 0;  // (the code below calls either floor FLOOR32; the thresholds differ)
    constexpr double kLogFloor = FLOOR == 2 ? -716.5 : -92.5, kScaleFloor = FLOOR == 2 ? 1e-311 : 9.18e-41;
    double scale0 = 1.0, scale1 = 1.0;
    if (a0 < 1.0 || a1 < 1.0) {
        Pair u;
        if (first) {
            u.a = Rng::to_unit(first->x, first->y);
            u.b = Rng::to_unit(first->z, first->w);
            ++rng.k;
        } else {
            u = uniform_pair(rng);
        }
        // FLOOR32: log U / alpha < log(FLT_MIN / 128) = -92.19 decides the floor; a float32 log with a margin
        // settles it for nearly every lane without the float64 log and exp (the value of a floored
        // scale does not matter: 0 stands for it)
        bool fl0 = false, fl1 = false;
        if (FLOOR32) {
            fl0 = (double)__logf((float)u.a) < kLogFloor * a0 - 1e-5;
            fl1 = (double)__logf((float)u.b) < kLogFloor * a1 - 1e-5;
        }
        if (a0 < 1.0) {
            if (fl0) scale0 = 0.0;
            else scale0 = a0 == 0.0 ? 0.0 : exp(flog(u.a) * frcp(a0));
            a0 += 1.0;
        }
        if (a1 < 1.0) {
            if (fl1) scale1 = 0.0;
            else scale1 = a1 == 0.0 ? 0.0 : exp(flog(u.b) * frcp(a1));
            a1 += 1.0;
        }
    }
    bool done0 = false, done1 = false;
    if (FLOOR32) {
        done0 = scale0 < kScaleFloor;  // FLT_MIN / 128, or DBL_MIN / 2e3
        done1 = scale1 < kScaleFloor;
        if (done0 && done1) {
            GammaPair out;
            out.g0 = out.g1 = 0.0;  // floored by the caller
            out.k = rng.k;
            return out;
        }
    }
    const double d0 = a0 - 1.0 / 3.0, d1 = a1 - 1.0 / 3.0;
    const double c0 = frcp(sqrt(9.0 * d0)), c1 = frcp(sqrt(9.0 * d1));
    double g0 = d0, g1 = d1;
#pragma unroll 1
    for (int it = 0; it < 64 && !(done0 && done1); ++it) {  // >= 95 % acceptance per round
        GammaRound q;
        if (first && rng.k == 0) q = gamma_round_from(*first);
        else q = gamma_round_at(rng.seed, rng.sub, rng.off + 4ull * rng.k);
        ++rng.k;
        if (!done0) {
            const double y = 1.0 + c0 * q.na;
            if (y > 0.0) {
                const double v = y * y * y, xx = q.na * q.na;
                if (q.ua < 1.0 - 0.0331 * xx * xx || accept_log(q.ua) < 0.5 * xx + d0 * (1.0 - v + accept_log(v))) {
                    g0 = d0 * v;
                    done0 = true;
                }
            }
        }
        if (!done1) {
            const double y = 1.0 + c1 * q.nb;
            if (y > 0.0) {
                const double v = y * y * y, xx = q.nb * q.nb;
                if (q.ub < 1.0 - 0.0331 * xx * xx || accept_log(q.ub) < 0.5 * xx + d1 * (1.0 - v + accept_log(v))) {
                    g1 = d1 * v;
                    done1 = true;
                }
            }
        }
    }
    GammaPair out;
    out.g0 = scale0 * g0;
    out.g1 = scale1 * g1;
    out.k = rng.k;
    return out;
}

// out-of-line copy for the kernels whose call sites have many live registers
__device__ BEAN_NOINLINE GammaPair sample_gamma_pair(double a0, double a1, Rng rng) {
    return sample_gamma_pair_inl(a0, a1, rng);
}
__device__ BEAN_NOINLINE GammaPair sample_gamma_pair_floor32(double a0, double a1, Rng rng) {
    return sample_gamma_pair_inl<1>(a0, a1, rng);
}
__device__ BEAN_NOINLINE GammaPair sample_gamma_pair_floord(double a0, double a1, Rng rng) {
    return sample_gamma_pair_inl<2>(a0, a1, rng);
}

// single draw (second component unused)
__device__ __forceinline__ double sample_gamma(double alpha, Rng& rng) {
    const GammaPair p = sample_gamma_pair(alpha, 1.0, rng);
    rng.k = p.k;
    return p.g0;
}

}  // namespace bean

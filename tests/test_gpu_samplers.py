"""Distribution tests of the in-kernel samplers.  -m gpu.

Exact-noise parity replays whatever the kernels drew, so a biased sampler would pass it; these tests
look at the draws themselves, exported through the C ABI (``dump_noise``), and compare their marginals
with the distributions the reference samples from:

* the A-component Dirichlet of the tiling guide kernels (paired Marsaglia-Tsang gammas, normalised):
  component a of Dirichlet(c) is Beta(c_a, sum(c) - c_a) (bean/model/model.py:942-950);
* the gamma draws of the survival Dirichlet-over-all-guides site, in the reference's float32
  semantics (floor FLT_MIN; survival_model.py:660-669);
* the standard normals of k_param (mu / sd of every target, logit_pi_noise and mu_negctrl of every
  guide; model.py:808-811, utils.py:144-155, survival_model.py:271-274).
"""
import numpy as np
import pytest
import scipy.stats as st
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
P_MIN = 1e-4  # per-test KS threshold (about 100 KS tests in this file)


@pytest.fixture(scope="module")
def engine():
    from bean_amd import engine as eng

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return eng


@pytest.mark.parametrize("A,seed", [(3, 1), (5, 2), (8, 3), (11, 4), (16, 5)])
def test_tiling_dirichlet_component_marginals(engine, A, seed):
    G, R, steps = 4096, 4, 4
    data = syn.make_sorting_tiling_screen(G, R, seed=30 + A, n_max_alleles=A)
    rng = np.random.default_rng(seed)
    conc = rng.permutation(np.geomspace(0.05, 6.0, A))
    data.allele_mask[:] = True          # every component is a free parameter
    data.repguide_mask[:] = True        # masked replicates export 1 / A instead of a draw
    data.pi_a0[:] = float(conc.sum())   # concentration of guide g: alpha / sum(alpha) * pi_a0
    eng = engine.HipSVI("MultiMixtureNormal", data.to(DEV), dump_noise=True, num_steps=steps + 1)
    eng.unconstrained["alpha_pi"].copy_(torch.as_tensor(np.log(conc), dtype=torch.float32).expand(G, A))
    draws = []
    for s in range(steps):
        eng.elbo_grad(step=s, seed=900 + A)
        draws.append(eng.drawn_noise()["pi"].cpu().numpy().reshape(-1, A))
    eng.close()
    pi = np.concatenate(draws)  # (steps * R * G, A)
    assert pi.shape[0] == steps * R * G and np.all(pi > 0) and np.all(pi < 1)
    np.testing.assert_allclose(pi.sum(1), 1.0, atol=1e-12)
    assert not np.array_equal(draws[0], draws[1])  # a new draw every step
    S = conc.sum()
    for a in range(A):
        ks = st.kstest(pi[:, a], st.beta(conc[a], S - conc[a]).cdf)
        assert ks.pvalue > P_MIN, (A, a, conc[a], ks)
    # second moments: Cov(pi_a, pi_b) = -c_a c_b / (S^2 (S + 1))
    a, b = int(np.argmax(conc)), int(np.argsort(conc)[-2])
    cov = np.cov(pi[:, a], pi[:, b])[0, 1]
    want = -conc[a] * conc[b] / (S * S * (S + 1))
    sd = np.std((pi[:, a] - pi[:, a].mean()) * (pi[:, b] - pi[:, b].mean())) / np.sqrt(pi.shape[0])
    assert abs(cov - want) < 5 * sd, (cov, want, sd)


def test_survival_q0_gamma_draws_with_float32_floor(engine):
    """The site's draw is exported normalised (x = gamma / sum gamma, float32 semantics); with the
    normaliser read back from the exchange buffer, gamma = x * sum is compared with Gamma(q0, 1).
    A concentration of 1e-3 puts ~92 % of the mass below float32's smallest normal: those draws sit on
    the floor exactly as torch's float32 sampler leaves them."""
    G, R, steps = 6000, 3, 6
    flt_min = float(np.finfo(np.float32).tiny)
    data = syn.make_survival_variant_screen(G, R, seed=41)
    eng = engine.HipSVI("MixtureNormal", data.to(DEV), dump_noise=True, num_steps=steps + 1)
    conc = np.empty(G)
    conc[:2000], conc[2000:4000], conc[4000:] = 0.5, 3.0, 1e-3
    eng.unconstrained["q0"].copy_(torch.as_tensor(np.log(conc), dtype=torch.float32))
    gsum = eng.exchange_buffers()["gsum"]  # (R + 1): sum_g gamma[r, g], sum_g q0
    gam, tot = [], []
    for s in range(steps):
        eng.elbo_grad(step=s, seed=77)
        x = eng.drawn_noise()["initial_abundance"].cpu().numpy()  # (R, G)
        t = gsum.cpu().numpy()[:R]
        gam.append(x * t[:, None])
        tot.append(np.broadcast_to(t[:, None], x.shape).copy())
        np.testing.assert_allclose(x.sum(1), 1.0, rtol=1e-5)
        assert x.min() >= flt_min and x.max() <= 1 - 2.0 ** -24
    eng.close()
    gam, tot = np.stack(gam), np.stack(tot)  # (steps, R, G)
    assert not np.array_equal(gam[0], gam[1])
    for lo, hi, a in ((0, 2000, 0.5), (2000, 4000, 3.0)):
        ks = st.kstest(gam[:, :, lo:hi].ravel(), st.gamma(a).cdf)
        assert ks.pvalue > P_MIN, (a, ks)
    # floored group: x == FLT_MIN wherever gamma / sum < FLT_MIN (covers the FLT_MIN floor of the
    # gamma itself: sum > 1)
    g3, t3 = gam[:, :, 4000:], tot[:, :, 4000:]
    floored = g3 <= flt_min * t3 * (1 + 1e-6)
    p_floor = st.gamma(1e-3).cdf(flt_min * t3).mean()
    n = floored.size
    assert 0.90 < p_floor < 0.95
    assert abs(floored.mean() - p_floor) < 4 * np.sqrt(p_floor * (1 - p_floor) / n), (floored.mean(), p_floor)
    free, cut = g3[~floored], (flt_min * t3)[~floored]
    d = st.gamma(1e-3)
    u = (d.cdf(free) - d.cdf(cut)) / d.sf(cut)  # probability transform of the truncated law
    assert st.kstest(u, "uniform").pvalue > P_MIN


def test_k_param_standard_normals(engine):
    G, steps = 20_000, 12
    data = syn.make_sorting_variant_screen(G, 2, seed=42, with_accessibility=True)
    eng = engine.HipSVI("MixtureNormal", data.to(DEV), dump_noise=True, scale_by_accessibility=True,
                        num_steps=steps + 1)
    cols = {"eps_mu": [], "eps_sd": [], "eps_noise": []}
    for s in range(steps):
        eng.elbo_grad(step=s, seed=5)
        d = eng.drawn_noise()
        for k in cols:
            cols[k].append(d[k].cpu().numpy().ravel())
    eng.close()
    for k, v in cols.items():
        x = np.concatenate(v)
        assert x.size >= 48_000
        ks = st.kstest(x, "norm")
        assert ks.pvalue > P_MIN, (k, ks)
        assert abs(st.kurtosis(x)) < 0.1 and abs(x.mean()) < 5 / np.sqrt(x.size)
        # consecutive steps are independent draws
        r = np.corrcoef(v[0], v[1])[0, 1]
        assert abs(r) < 5 / np.sqrt(v[0].size), (k, r)
    r = np.corrcoef(np.concatenate(cols["eps_mu"]), np.concatenate(cols["eps_sd"]))[0, 1]
    assert abs(r) < 5 / np.sqrt(steps * data.n_targets)


def test_survival_baseline_draws_are_standard_normal(engine):
    G, steps = 20_000, 6
    data = syn.make_survival_variant_screen(G, 2, seed=43)
    m0, s0 = 0.05, 0.2
    eng = engine.HipSVI("MixtureNormal", data.to(DEV), dump_noise=True, num_steps=steps + 1, mu_negctrl=(m0, s0))
    xs = []
    for s in range(steps):
        eng.elbo_grad(step=s, seed=6)
        u = eng.drawn_noise()["mu_negctrl"].cpu().numpy().ravel()
        xs.append((u - float(np.float32(m0))) / float(np.float32(s0)))
    eng.close()
    x = np.concatenate(xs)
    assert st.kstest(x, "norm").pvalue > P_MIN
    assert abs(np.corrcoef(xs[0], xs[1])[0, 1]) < 5 / np.sqrt(G)

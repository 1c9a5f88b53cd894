"""Known-answer tests that pin the oracle's building blocks.

The reference cannot run here (no Pyro), so each restated piece is checked
against an independent implementation of the same published formula (scipy) or
against an analytic identity.
"""
import itertools
import math

import numpy as np
import pytest
import scipy.special as sp
import scipy.stats as st
import torch

import bean_amd  # noqa: F401
from oracle import elbo, svi

torch.set_num_threads(4)


def test_dirichlet_multinomial_matches_scipy():
    rng = np.random.default_rng(0)
    alpha = np.exp(rng.normal(0, 1.5, (50, 5)))
    x = rng.integers(0, 40, (50, 5)).astype(float)
    got = elbo.dirichlet_multinomial_log_prob(torch.tensor(alpha), torch.tensor(x)).numpy()
    want = np.array([st.dirichlet_multinomial.logpmf(x[i].astype(int), alpha[i], int(x[i].sum())) for i in range(50)])
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)


def test_dirichlet_multinomial_pmf_sums_to_one():
    alpha = torch.tensor([0.7, 2.5, 1.2], dtype=torch.float64)
    n = 6
    support = [c for c in itertools.product(range(n + 1), repeat=3) if sum(c) == n]
    lp = elbo.dirichlet_multinomial_log_prob(alpha.expand(len(support), 3), torch.tensor(support, dtype=torch.float64))
    assert abs(float(lp.exp().sum()) - 1.0) < 1e-12


def test_bin_probabilities_partition_of_unity_and_scipy():
    edges = torch.tensor([0.0, 0.2, 0.4, 0.6, 0.8, 1.0], dtype=torch.float64)
    lq, uq = edges[:-1], edges[1:]
    mu = torch.tensor([-1.3, 0.0, 0.4, 2.0], dtype=torch.float64)
    sd = torch.tensor([0.5, 1.0, 1.7, 0.2], dtype=torch.float64)
    B, G = 5, 4
    p = elbo.std_normal_bin_prob(uq[:, None].expand(B, G), lq[:, None].expand(B, G),
                                 mu[None].expand(B, G), sd[None].expand(B, G))
    np.testing.assert_allclose(p.sum(0).numpy(), 1.0, atol=1e-14)
    z = st.norm.ppf(edges.numpy())
    want = st.norm.cdf((z[1:, None] - mu.numpy()) / sd.numpy()) - st.norm.cdf((z[:-1, None] - mu.numpy()) / sd.numpy())
    np.testing.assert_allclose(p.numpy(), want, atol=1e-12)
    # wild-type component (mu 0, sd 1) recovers the quantile width
    p0 = elbo.std_normal_bin_prob(uq, lq, torch.zeros(5, dtype=torch.float64), torch.ones(5, dtype=torch.float64))
    np.testing.assert_allclose(p0.numpy(), 0.2, atol=1e-14)


def test_bin_probabilities_allele_mask():
    uq = torch.tensor([[0.5, 0.5], [1.0, 1.0]], dtype=torch.float64)
    lq = torch.tensor([[0.0, 0.0], [0.5, 0.5]], dtype=torch.float64)
    mu = torch.zeros(2, 2, dtype=torch.float64)
    sd = torch.ones(2, 2, dtype=torch.float64)
    mask = torch.tensor([[True, False], [True, False]])
    p = elbo.std_normal_bin_prob(uq, lq, mu, sd, mask=mask)
    assert torch.all(p[:, 1] == 0)
    np.testing.assert_allclose(p[:, 0].numpy(), 0.5, atol=1e-14)


def test_bin_probabilities_gradients_are_finite_at_open_edges():
    uq = torch.tensor([0.3, 1.0], dtype=torch.float64)
    lq = torch.tensor([0.0, 0.3], dtype=torch.float64)
    mu = torch.tensor([0.1, 0.1], dtype=torch.float64, requires_grad=True)
    sd = torch.tensor([1.3, 1.3], dtype=torch.float64, requires_grad=True)
    p = elbo.std_normal_bin_prob(uq, lq, mu, sd)
    p[0].backward()
    assert torch.isfinite(mu.grad).all() and torch.isfinite(sd.grad).all()
    assert abs(float(p.detach().sum()) - 1.0) < 1e-14


def test_get_alpha_invariants():
    R, B, G = 2, 4, 7
    g = torch.Generator().manual_seed(1)
    e = torch.rand((R, B, G), generator=g, dtype=torch.float64)
    sf = torch.rand((R, B), generator=g, dtype=torch.float64) + 0.5
    a0 = torch.rand(G, generator=g, dtype=torch.float64) * 100
    mask = torch.ones((R, B), dtype=torch.int64)
    a = elbo.dirmult_concentration(e, sf, mask, a0)
    assert a.shape == (R, G, B)
    # with all samples kept the concentrations sum to a0 (up to the epsilon terms)
    np.testing.assert_allclose(a.sum(-1).numpy(), a0[None].expand(R, G).numpy(), rtol=1e-4)
    mask[1, 2] = 0
    a = elbo.dirmult_concentration(e, sf, mask, a0)
    assert torch.all(a[1, :, 2] == elbo.EPS)
    # literal formula
    p = e.permute(0, 2, 1) * sf[:, None, :]
    want = ((p + 1e-5 / B) / (p.sum(-1, keepdim=True) + 1e-5) * a0[None, :, None] * mask[:, None, :]).clamp(min=1e-5)
    np.testing.assert_allclose(a.numpy(), want.numpy(), rtol=1e-14)


def test_torch_distributions_match_scipy():
    # the oracle evaluates these sites with torch.distributions, the classes Pyro wraps
    x = torch.tensor([0.2, 0.5, 0.3], dtype=torch.float64)
    c = torch.tensor([1.5, 0.8, 3.0], dtype=torch.float64)
    assert abs(float(torch.distributions.Dirichlet(c).log_prob(x)) - st.dirichlet.logpdf(x.numpy(), c.numpy())) < 1e-12
    counts = torch.tensor([3.0, 0.0, 5.0], dtype=torch.float64)
    got = torch.distributions.Multinomial(probs=x, validate_args=False).log_prob(counts)
    assert abs(float(got) - st.multinomial.logpmf(counts.numpy(), 8, x.numpy())) < 1e-12
    assert abs(float(torch.distributions.Laplace(0.0, 1.0).log_prob(torch.tensor(0.7))) - st.laplace.logpdf(0.7)) < 1e-6
    d64 = lambda v: torch.tensor(v, dtype=torch.float64)
    got = torch.distributions.LogNormal(d64(0.0), d64(0.01)).log_prob(d64(1.01))
    assert abs(float(got) - st.lognorm.logpdf(1.01, 0.01)) < 1e-9


def test_given_sample_gradient_equals_rsample_gradient():
    conc = torch.tensor([[0.4, 2.0], [5.0, 7.0], [30.0, 3.0]], dtype=torch.float64, requires_grad=True)
    w = torch.tensor([[1.0, -2.0], [0.3, 0.9], [-1.0, 4.0]], dtype=torch.float64)
    torch.manual_seed(5)
    x = torch.distributions.Dirichlet(conc).rsample()
    (x * w).sum().backward()
    g1 = conc.grad.clone()
    conc.grad = None
    y = elbo.dirichlet_rsample(conc, x.detach())
    (y * w).sum().backward()
    np.testing.assert_allclose(conc.grad.numpy(), g1.numpy(), rtol=1e-12)


def test_dirichlet_implicit_gradient_against_finite_difference_of_cdf():
    # -(dF/dalpha)/pdf for Beta(a, b): torch's approximation is accurate to ~1e-3
    a, b, x = 2.3, 4.1, 0.31
    h = 1e-5
    dF = (st.beta.cdf(x, a + h, b) - st.beta.cdf(x, a - h, b)) / (2 * h)
    exact = -dF / st.beta.pdf(x, a, b)
    xs = torch.tensor([[x, 1 - x]], dtype=torch.float64)
    conc = torch.tensor([[a, b]], dtype=torch.float64)
    g = torch._dirichlet_grad(xs, conc, conc.sum(-1, True).expand_as(conc))
    # torch returns the gradient scaled by 1/(1-x)
    assert abs(float(g[0, 0]) * (1 - x) - exact) < 2e-3 * abs(exact)


def test_clipped_adam_hand_rolled():
    p = {"w": torch.tensor([1.0, -2.0, 0.5], requires_grad=True)}
    opt = svi.ClippedAdam(p, lr=0.01, lrd=0.9)
    grads = [torch.tensor([0.5, -30.0, 2.0]), torch.tensor([-1.0, 4.0, 50.0]), torch.tensor([0.1, 0.2, -0.3])]
    w = np.array([1.0, -2.0, 0.5])
    m = np.zeros(3)
    v = np.zeros(3)
    lr = 0.01
    for t, g in enumerate(grads, 1):
        p["w"].grad = g.clone()
        opt.step()
        lr *= 0.9
        gc = np.clip(g.numpy().astype(float), -10, 10)
        m = 0.9 * m + 0.1 * gc
        v = 0.999 * v + 0.001 * gc * gc
        w = w - lr * math.sqrt(1 - 0.999**t) / (1 - 0.9**t) * m / (np.sqrt(v) + 1e-8)
        np.testing.assert_allclose(p["w"].detach().numpy(), w, rtol=2e-6)


@pytest.mark.parametrize("family", ["MixtureNormal", "Normal", "ControlNormal"])
def test_elbo_gradient_finite_difference(small_screen, family):
    """d loss / d mu_loc from autograd agrees with a central difference in float64."""
    data = elbo.as_float64(small_screen)
    torch.manual_seed(3)
    params = {k: v.detach().double() + 0.1 for k, v in elbo.init_params(family, data).items()}
    params = {k: v.requires_grad_(True) for k, v in params.items()}
    R, G, T = data.n_reps, data.n_guides, data.n_targets
    shape = () if family == "ControlNormal" else (T, 1)
    noise = {"eps_mu": torch.randn(shape, dtype=torch.float64), "eps_sd": torch.randn(shape, dtype=torch.float64)}
    if family == "MixtureNormal":
        noise["pi"] = torch.distributions.Dirichlet(torch.tensor([3.0, 5.0], dtype=torch.float64)).sample((R, 1, G))
    fn = elbo.LOSSES[family]
    _, grads, _ = svi.loss_and_grads(fn, data, params, noise=noise)
    for name in ("mu_loc", "sd_scale"):
        idx = (0, 0) if family != "ControlNormal" else ()
        h = 1e-6
        with torch.no_grad():
            base = params[name][idx].item()
            params[name][idx] = base + h
            lp = float(fn(data, params, noise=noise))
            params[name][idx] = base - h
            lm = float(fn(data, params, noise=noise))
            params[name][idx] = base
        fd = (lp - lm) / (2 * h)
        assert abs(fd - float(grads[name][idx])) < 1e-4 * max(1.0, abs(fd)), (name, fd, float(grads[name][idx]))


def test_param_store_layout_matches_shipped_example():
    """Names, shapes and log-space storage of the fitted parameters follow the
    reference's example output (docs/example_run_output/variant/*.result.pkl:
    mu_loc/mu_scale/sd_loc/sd_scale (T,1), alpha_pi (G,2), noise_* (G,))."""
    from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

    data = make_sorting_variant_screen(120, 2, seed=3, with_accessibility=True)
    p = elbo.init_params("MixtureNormal", data, fit_noise=True, scale_by_acc=True)
    T, G = data.n_targets, data.n_guides
    assert {k: tuple(v.shape) for k, v in p.items()} == {
        "mu_loc": (T, 1), "mu_scale": (T, 1), "sd_loc": (T, 1), "sd_scale": (T, 1),
        "alpha_pi": (G, 2), "noise_loc": (G,), "noise_scale": (G,)}
    # unconstrained storage: log(1) = 0 for scales, log(0.655) for noise_scale
    assert float(p["mu_scale"].abs().max()) == 0.0
    np.testing.assert_allclose(p["noise_scale"].detach().numpy(), math.log(0.655), rtol=1e-6)
    assert all(v.dtype == torch.float32 for v in p.values())


def test_oracle_fit_decreases_loss_and_recovers_effects():
    from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

    torch.manual_seed(101)
    data = make_sorting_variant_screen(400, 3, seed=5, frac_effect=0.5)
    params = elbo.init_params("MixtureNormal", data)
    params, losses = svi.run_svi(elbo.mixture_normal_loss, data, params, num_steps=150)
    assert losses[-1] < 0.8 * losses[0]
    assert isinstance(losses[0], float)
    mu = params["mu_loc"].detach().numpy().ravel()
    truth = data.truth["mu"]
    big = np.abs(truth) > 0.8
    # the fit has moved the strong effects in the right direction
    assert np.mean(np.sign(mu[big]) == np.sign(truth[big])) > 0.8


@pytest.mark.parametrize("n_alleles,mode", [(4, "ref"), (8, "f64"), (13, "f64")])
def test_sparse_allele_moments_equal_the_dense_form(n_alleles, mode):
    """The gather / index_add form of allele_to_edit @ mu_edits and ||allele_to_edit * sd_edits||
    (bean/model/model.py:618-622) used at BASELINE config 3's size equals the reference's dense
    form: loss and every gradient."""
    from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen

    data = make_sorting_tiling_screen(150, 2, seed=21, n_max_alleles=n_alleles, mask_fraction=0.05)
    torch.manual_seed(3)
    params = elbo.init_params("MultiMixtureNormal", data)
    if mode == "f64":
        data = elbo.as_float64(data)
        params = {k: v.detach().double().requires_grad_(True) for k, v in params.items()}
    with torch.no_grad():
        for k, v in params.items():
            noise = 0.3 * torch.randn_like(v)
            v.add_(noise * data.allele_mask if k == "alpha_pi" else noise)
    g = torch.Generator().manual_seed(5)
    conc = torch.rand((data.n_reps, 1, data.n_guides, n_alleles), generator=g, dtype=torch.float64) + 0.2
    draws = {"eps_mu": torch.randn(data.n_edits, generator=g), "eps_sd": torch.randn(data.n_edits, generator=g),
             "pi": torch.distributions.Dirichlet(conc).sample()}
    dense = svi.loss_and_grads(elbo.multi_mixture_normal_loss, data, params, noise=draws)
    sparse = svi.loss_and_grads(elbo.multi_mixture_normal_loss, data, params, noise=draws, sparse=True)
    tol = 1e-12 if mode == "f64" else 2e-6
    assert abs(dense[0] - sparse[0]) <= tol * abs(dense[0])
    for k in dense[1]:
        ref = dense[1][k].double()
        err = (sparse[1][k].double() - ref).abs().max().item()
        assert err <= max(tol * 10, 1e-11) * (ref.abs().max().item() + 1e-30), (k, err)

"""One-particle negative ELBO of the sorting-screen model families (oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Each ``*_loss`` function runs
the guide first, then replays the model on the guide's draws, and returns
``-(sum log p - sum log q)`` exactly as ``pyro.infer.Trace_ELBO`` does for fully
reparameterised latents.  Tensor dtypes are left to
PyTorch's promotion rules on inputs that carry the reference's dtypes (counts
f32, size factors / ``a0`` / ``pi_a0`` / bounds f64), so float32/float64 mixing
matches the reference op for op.

``params`` always holds the *unconstrained* leaves the optimiser updates
(``log`` of every positive parameter, Appendix A.6 item 4).  ``noise`` lets a
test inject the random draws (standard-normal ``eps_*`` and the Dirichlet sample
``pi``); anything missing is drawn from torch's global RNG.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.distributions as tdist

EPS = 1e-5
ACC_A, ACC_B, PI_NOISE_SD = 0.2513, -1.9458, 0.655


# ------------------------------------------------------------------ primitives
def masked_sum(logp: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """``poutine.mask`` + ``log_prob_sum`` (Appendix A.6 item 2)."""
    if mask is not None:
        logp = torch.where(mask, logp, logp.new_zeros(()))
    return logp.sum()


def dirichlet_multinomial_log_prob(conc: torch.Tensor, value: torch.Tensor):
    """Pyro's dense ``DirichletMultinomial.log_prob`` (pyro-ppl 1.8/1.9,
    ``pyro/distributions/conjugate.py``; call sites ``bean/model/model.py:531-547``):
    ``logB1(sum a, sum x) - sum_k logB1(a_k, x_k)`` with
    ``logB1(a, x) = lgamma(1+x) + lgamma(a) - lgamma(x+a)``."""

    def log_beta_1(a, x):
        return torch.lgamma(1 + x) + torch.lgamma(a) - torch.lgamma(x + a)

    return log_beta_1(conc.sum(-1), value.sum(-1)) - log_beta_1(conc, value).sum(-1)


class _DirichletGivenSample(torch.autograd.Function):
    """Identity on a supplied Dirichlet draw ``x`` whose backward is the implicit
    reparameterisation gradient torch uses for ``Dirichlet.rsample``
    (``torch/distributions/dirichlet.py:17-36``)."""

    @staticmethod
    def forward(ctx, concentration, x):
        ctx.save_for_backward(x, concentration)
        return x.clone()

    @staticmethod
    def backward(ctx, grad_output):
        x, concentration = ctx.saved_tensors
        total = concentration.sum(-1, True).expand_as(concentration)
        grad = torch._dirichlet_grad(x, concentration, total)
        return grad * (grad_output - (x * grad_output).sum(-1, True)), None


def dirichlet_rsample(concentration: torch.Tensor, given: Optional[torch.Tensor]):
    if given is None:
        return tdist.Dirichlet(concentration).rsample()
    return _DirichletGivenSample.apply(
        concentration.contiguous(), given.to(concentration.dtype).contiguous()
    )


def normal_rsample(loc, scale, eps: Optional[torch.Tensor]):
    if eps is None:
        eps = torch.randn(loc.shape, dtype=loc.dtype)
    return loc + eps.to(loc.dtype) * scale


def std_normal_bin_prob(uq, lq, mu, sd, mask=None):
    """P(lq-quantile < N(mu, sd) < uq-quantile of N(0,1))
    (``bean/model/utils.py:34-76``).  Edges at quantile exactly 1.0 / 0.0 give
    cdf 1 / 0; with ``mask`` invalid alleles get ``sd + 100`` then probability 0."""
    top, bottom = uq == 1.0, lq == 0.0
    std = tdist.Normal(0.0, 1.0)
    half = torch.full_like(uq, 0.5)
    z_hi = std.icdf(torch.where(top, half, uq))
    z_lo = std.icdf(torch.where(bottom, half, lq))
    if mask is not None:
        sd = sd + (~mask).long() * 100
    dist = tdist.Normal(mu, sd, validate_args=False)
    c_hi = torch.where(top, torch.ones_like(uq), dist.cdf(z_hi))
    c_lo = torch.where(bottom, torch.zeros_like(lq), dist.cdf(z_lo))
    res = c_hi - c_lo
    if mask is not None:
        res = torch.where(mask, res, torch.zeros_like(res))
    return res


def dirmult_concentration(expected_guide_p, size_factor, sample_mask, a0, eps=EPS):
    """``get_alpha`` (``bean/model/utils.py:10-31``): (R,B,G) -> (R,G,B)."""
    p = expected_guide_p.permute(0, 2, 1) * size_factor[:, None, :]
    a = (p + eps / p.shape[-1]) / (p.sum(-1)[:, :, None] + eps) * a0[None, :, None]
    return (a * sample_mask[:, None, :]).clamp(min=eps)


def scale_pi_by_accessibility(pi, acc, noise):
    """``scale_pi_by_accessibility`` + ``add_noise_to_pi`` on a drawn
    ``logit_pi_noise`` (``bean/model/utils.py:79-178``)."""
    scaled = (
        pi[..., 1:] * torch.exp(torch.tensor(ACC_B)) * torch.pow(acc, ACC_A).unsqueeze(-1)
    )
    ctrl = torch.ones(pi[..., 0].shape) - scaled.sum(-1)
    pi = torch.cat([ctrl.unsqueeze(-1), scaled], -1)
    pi = pi / pi.sum(-1).clamp(min=1.0)[..., None]
    n_reps, _, n_guides, n_alleles = pi.shape
    logit = torch.logit(pi[..., 1:].clamp(min=1e-3, max=1 - 1e-3))
    logit = logit + noise[None, None, :, None].expand(n_reps, 1, -1, n_alleles - 1)
    e = torch.exp(logit)
    noised = (e / (1 + e)).clamp(min=1e-3, max=1 - 1e-3)
    first = torch.ones(pi[:, :, :, 0].shape) - noised.sum(-1)
    return torch.cat([first.unsqueeze(-1), noised], -1)


def _count_likelihoods(data, expected_guide_p, use_bcmatch, mask_thres=10):
    """The two DirMult observation sites shared by every family
    (e.g. ``bean/model/model.py:506-547``)."""
    out = {}
    a = dirmult_concentration(
        expected_guide_p, data.size_factor, data.sample_mask, data.a0
    )
    obs = data.X_masked.permute(0, 2, 1)
    m = torch.logical_and(obs.sum(-1) > mask_thres, data.repguide_mask)
    out["guide_counts"] = masked_sum(dirichlet_multinomial_log_prob(a, obs), m)
    if use_bcmatch:
        a_bc = dirmult_concentration(
            expected_guide_p, data.size_factor_bcmatch, data.sample_mask, data.a0_bcmatch
        )
        obs_bc = data.X_bcmatch_masked.permute(0, 2, 1)
        m_bc = torch.logical_and(obs_bc.sum(-1) > mask_thres, data.repguide_mask)
        out["guide_bcmatch_counts"] = masked_sum(
            dirichlet_multinomial_log_prob(a_bc, obs_bc), m_bc
        )
    return out


def _priors(shape, sd_scale, prior_params):
    """Prior distributions of mu/sd (``model.py:405-428``)."""
    sd_loc = torch.zeros(shape)
    sd_sc = torch.ones(shape) * sd_scale
    mu_dist = tdist.Laplace(0.0, 1.0)
    if prior_params is not None:
        sd_loc = prior_params.get("sd_loc", sd_loc)
        sd_sc = prior_params.get("sd_scale", sd_sc)
        if "mu_loc" in prior_params or "mu_scale" in prior_params:
            mu_dist = tdist.Normal(
                prior_params.get("mu_loc", 0.0), prior_params.get("mu_scale", 1.0)
            )
    return mu_dist, tdist.LogNormal(sd_loc, sd_sc)


def _finish(model_lp: Dict[str, torch.Tensor], guide_lp: Dict[str, torch.Tensor], record):
    """``Trace_ELBO``: loss = -(sum model log p - sum guide log q)."""
    elbo = 0.0
    for v in model_lp.values():
        elbo = elbo + v.double()
    for v in guide_lp.values():
        elbo = elbo - v.double()
    if record is not None:
        record["model"] = {k: float(v.detach()) for k, v in model_lp.items()}
        record["guide"] = {k: float(v.detach()) for k, v in guide_lp.items()}
    return -elbo


def as_float64(data):
    """Copy of ``data`` with every floating tensor promoted to float64 (the
    "f64" oracle mode used to check the kernel algebra below float32 rounding;
    the default mode keeps the reference's mixed dtypes)."""
    import copy

    out = copy.copy(data)
    for k, v in list(vars(data).items()):
        if isinstance(v, torch.Tensor) and v.dtype == torch.float32:
            setattr(out, k, v.double())
    return out


def _noise(noise, key):
    return None if noise is None else noise.get(key)


# ------------------------------------------------------------------- families
def init_params(family: str, data, fit_noise: bool = True, scale_by_acc: bool = False):
    """Unconstrained initial parameters (``pyro.param`` initial values at
    ``model.py:757-769,800-830,866-873,893-937``; positives stored as log)."""
    f = torch.float32
    if family == "ControlNormal":
        p = {k: torch.tensor(0.0, dtype=f) for k in ("mu_loc", "mu_scale", "sd_loc", "sd_scale")}
    elif family == "MultiMixtureNormal":
        E = data.n_edits
        p = {k: torch.zeros(E, dtype=f) for k in ("mu_loc", "mu_scale", "sd_loc", "sd_scale")}
    else:
        T = data.n_targets
        p = {k: torch.zeros((T, 1), dtype=f) for k in ("mu_loc", "mu_scale", "sd_loc", "sd_scale")}
    if family == "Normal" and getattr(data, "sample_covariates", None) is not None:
        p["mu_cov_loc"] = torch.zeros(data.n_sample_covariates, dtype=f)
        p["mu_cov_scale"] = torch.zeros(data.n_sample_covariates, dtype=f)  # log(1)
    if family in ("MixtureNormal", "MultiMixtureNormal"):
        A = data.n_max_alleles
        a0 = torch.ones((data.n_guides, A), dtype=f)
        if family == "MultiMixtureNormal":
            a0[~data.allele_mask] = EPS
        p["alpha_pi"] = a0.log()
        if scale_by_acc and fit_noise:
            p["noise_loc"] = torch.zeros(data.n_guides, dtype=f)
            p["noise_scale"] = torch.full((data.n_guides,), PI_NOISE_SD, dtype=f).log()
    return {k: v.clone().requires_grad_(True) for k, v in p.items()}


POSITIVE = ("mu_scale", "sd_scale", "alpha_pi", "noise_scale", "mu_cov_scale", "q0")


def constrained(params):
    return {k: (v.exp() if k in POSITIVE else v) for k, v in params.items()}


def normal_loss(data, params, noise=None, use_bcmatch=True, sd_scale=0.01,
                prior_params=None, record=None):
    """``NormalModel`` + ``NormalGuide`` (``model.py:19-165,754-782``): one
    component per guide, scale ``sqrt(sd_target)`` (``model.py:92-98``)."""
    P = constrained(params)
    R, B, G = data.n_reps, data.n_condits, data.n_guides
    q_mu = tdist.Normal(P["mu_loc"], P["mu_scale"])
    q_sd = tdist.LogNormal(P["sd_loc"], P["sd_scale"])
    mu_t = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    sd_t = normal_rsample(P["sd_loc"], P["sd_scale"], _noise(noise, "eps_sd")).exp()
    guide_lp = {"mu_targets": q_mu.log_prob(mu_t).sum(), "sd_targets": q_sd.log_prob(sd_t).sum()}

    p_mu, p_sd = _priors((data.n_targets, 1), sd_scale, prior_params)
    model_lp = {"mu_targets": p_mu.log_prob(mu_t).sum(), "sd_targets": p_sd.log_prob(sd_t).sum()}
    mu = torch.repeat_interleave(mu_t, data.target_lengths, dim=0)
    sd = torch.repeat_interleave(sd_t, data.target_lengths, dim=0)
    mu = mu[None, None].expand(R, B, -1, -1)
    if getattr(data, "sample_covariates", None) is not None:
        # sample covariates (model.py:73-76, 88-91; guide 771-782): one Normal site per covariate,
        # but only the FIRST column of rep_by_cov * mu_cov shifts the replicates' means
        # (`(data.rep_by_cov * mu_cov)[:, 0]`)
        mu_cov = normal_rsample(P["mu_cov_loc"], P["mu_cov_scale"], _noise(noise, "eps_cov"))
        guide_lp["mu_cov"] = tdist.Normal(P["mu_cov_loc"], P["mu_cov_scale"]).log_prob(mu_cov).sum()
        model_lp["mu_cov"] = tdist.Normal(0.0, 1.0).log_prob(mu_cov).sum()
        mu = mu + (data.rep_by_cov * mu_cov)[:, 0][:, None, None, None].expand(-1, B, G, 1)
    sd = torch.sqrt(sd[None, None].expand(R, B, -1, -1))
    uq = data.upper_bounds[None, :, None, None].expand(R, -1, G, 1)
    lq = data.lower_bounds[None, :, None, None].expand(R, -1, G, 1)
    expected_guide_p = std_normal_bin_prob(uq, lq, mu, sd).sum(-1)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch))
    return _finish(model_lp, guide_lp, record)


def control_normal_loss(data, params, noise=None, use_bcmatch=True, record=None):
    """``ControlNormalModel`` + ``ControlNormalGuide`` (``model.py:168-252,861-875``):
    one scalar mu / sd shared by all guides, prior sd ~ LogNormal(0, 1)."""
    P = constrained(params)
    R, B, G = data.n_reps, data.n_condits, data.n_guides
    mu_t = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    sd_t = normal_rsample(P["sd_loc"], P["sd_scale"], _noise(noise, "eps_sd")).exp()
    guide_lp = {
        "mu_targets": tdist.Normal(P["mu_loc"], P["mu_scale"]).log_prob(mu_t).sum(),
        "sd_targets": tdist.LogNormal(P["sd_loc"], P["sd_scale"]).log_prob(sd_t).sum(),
    }
    model_lp = {
        "mu_targets": tdist.Laplace(0.0, 1.0).log_prob(mu_t).sum(),
        "sd_targets": tdist.LogNormal(0.0, 1.0).log_prob(sd_t).sum(),
    }
    mu = mu_t.repeat(G).unsqueeze(-1)
    sd = sd_t.repeat(G).unsqueeze(-1)
    uq = data.upper_bounds[:, None, None].expand(-1, G, 1)
    lq = data.lower_bounds[:, None, None].expand(-1, G, 1)
    p_bin = std_normal_bin_prob(uq, lq, mu[None].expand(B, -1, -1), sd[None].expand(B, -1, -1))
    expected_guide_p = p_bin[None].expand(R, -1, -1, -1).sum(-1)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch))
    return _finish(model_lp, guide_lp, record)


def mixture_normal_loss(data, params, noise=None, use_bcmatch=True, sd_scale=0.01,
                        scale_by_accessibility=False, fit_noise=True,
                        prior_params=None, record=None):
    """``MixtureNormalModel`` + ``MixtureNormalGuide`` (``model.py:378-547,785-858``).

    ``fit_noise`` is the *guide's* flag; the model always uses the fixed
    ``N(0, 0.655)`` prior for ``logit_pi_noise`` because ``identify_model_guide``
    forwards ``fit_noise`` only to the guide (``bean/model/run.py:447-456``).
    """
    P = constrained(params)
    R, B, G = data.n_reps, data.n_condits, data.n_guides
    T = data.n_targets
    # ---- guide
    mu_t = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    sd_t = normal_rsample(P["sd_loc"], P["sd_scale"], _noise(noise, "eps_sd")).exp()
    guide_lp = {
        "mu_targets": tdist.Normal(P["mu_loc"], P["mu_scale"]).log_prob(mu_t).sum(),
        "sd_targets": tdist.LogNormal(P["sd_loc"], P["sd_scale"]).log_prob(sd_t).sum(),
    }
    alpha_pi = P["alpha_pi"]
    pi_a_scaled = alpha_pi / alpha_pi.sum(-1)[:, None] * data.pi_a0[:, None]
    conc_q = pi_a_scaled[None, None].expand(R, 1, -1, -1).clamp(1e-5)
    pi = dirichlet_rsample(conc_q, _noise(noise, "pi"))
    guide_lp["pi"] = tdist.Dirichlet(conc_q, validate_args=False).log_prob(pi).sum()
    model_lp = {}
    pi_eff = pi
    if scale_by_accessibility:
        if fit_noise:
            q_noise = tdist.Normal(P["noise_loc"], P["noise_scale"])
            lpn = normal_rsample(P["noise_loc"], P["noise_scale"], _noise(noise, "eps_noise"))
        else:
            q_noise = tdist.Normal(torch.zeros(G), torch.full((G,), PI_NOISE_SD))
            lpn = normal_rsample(q_noise.loc, q_noise.scale, _noise(noise, "eps_noise"))
        guide_lp["logit_pi_noise"] = q_noise.log_prob(lpn).sum()
        model_lp["logit_pi_noise"] = (
            tdist.Normal(0.0, PI_NOISE_SD).log_prob(lpn).sum()
        )
        pi_eff = scale_pi_by_accessibility(pi, data.guide_accessibility, lpn)
    # ---- model replay
    p_mu, p_sd = _priors((T, 1), sd_scale, prior_params)
    model_lp["mu_targets"] = p_mu.log_prob(mu_t).sum()
    model_lp["sd_targets"] = p_sd.log_prob(sd_t).sum()
    mu = torch.repeat_interleave(
        torch.cat([torch.zeros((T, 1)), mu_t], -1), data.target_lengths, dim=0
    )
    sd = torch.repeat_interleave(
        torch.cat([torch.ones((T, 1)), sd_t], -1), data.target_lengths, dim=0
    )
    rg = data.repguide_mask.unsqueeze(1)
    conc_p = pi_a_scaled[None, None].expand(R, 1, -1, -1)
    model_lp["pi"] = masked_sum(
        tdist.Dirichlet(conc_p, validate_args=False).log_prob(pi), rg
    )
    model_lp["bulk_allele_count"] = masked_sum(
        tdist.Multinomial(probs=pi, validate_args=False).log_prob(
            data.allele_counts_control
        ),
        rg,
    )
    uq = data.upper_bounds[:, None, None].expand(-1, G, 2)
    lq = data.lower_bounds[:, None, None].expand(-1, G, 2)
    p_bin = std_normal_bin_prob(uq, lq, mu[None].expand(B, -1, -1), sd[None].expand(B, -1, -1))
    expected_guide_p = (pi_eff.expand(R, B, -1, -1) * p_bin[None]).sum(-1)
    # use_bcmatch is a 1-tuple at the call site => always truthy (run.py:450, F5)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch))
    return _finish(model_lp, guide_lp, record)


def allele_moments(data, mu_e, sd_e, sparse: bool):
    """Per-allele mean and scale from the per-edit draws (``model.py:618-622``):
    ``matmul(allele_to_edit, mu_edits)`` and ``linalg.norm(allele_to_edit * sd_edits, dim=-1)``.

    ``sparse=False`` is the reference's dense 0/1 ``(G, A-1, E)`` form.  ``sparse=True`` evaluates
    the same two sums through the CSR map with ``index_add`` (one term per nonzero instead of
    ``G (A-1) E`` products), which is what lets the oracle run at BASELINE config 3's size (193k
    allele slots x 30k edits would be 46 GB dense); ``tests/test_oracle_kat.py`` pins it to the
    dense form at small sizes.  Rows without edits have scale 0 in both forms (the reference
    masks them through ``allele_mask``)."""
    G, A1 = data.n_guides, data.n_max_alleles - 1
    if not sparse:
        a2e = data.allele_to_edit_dense().to(mu_e.dtype)  # f32 in the reference
        return torch.matmul(a2e, mu_e), torch.linalg.norm(a2e * sd_e[None, None, :], dim=-1)
    ptr = data.a2e_ptr.to(torch.int64)
    idx = data.a2e_idx.to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(G * A1, dtype=torch.int64), ptr[1:] - ptr[:-1])
    mu_a = torch.zeros(G * A1, dtype=mu_e.dtype).index_add(0, rows, mu_e[idx])
    var_a = torch.zeros(G * A1, dtype=sd_e.dtype).index_add(0, rows, sd_e[idx] ** 2)
    # d sqrt(v) / dv at v = 0 is infinite: keep empty rows out of the graph as norm() does (its
    # subgradient at 0 is 0)
    live = var_a > 0
    sd_a = torch.where(live, torch.sqrt(torch.where(live, var_a, torch.ones_like(var_a))), torch.zeros_like(var_a))
    return mu_a.reshape(G, A1), sd_a.reshape(G, A1)


def multi_mixture_normal_loss(data, params, noise=None, use_bcmatch=True, sd_scale=0.01,
                              scale_by_accessibility=False, fit_noise=True,
                              prior_params=None, record=None, eps=EPS, sparse=False):
    """``MultiMixtureNormalModel`` + ``MultiMixtureNormalGuide``
    (``model.py:550-751,878-962``): per-edit latents, allele = sum of edits."""
    P = constrained(params)
    R, B, G, A, E = data.n_reps, data.n_condits, data.n_guides, data.n_max_alleles, data.n_edits
    mu_e = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    sd_e = normal_rsample(P["sd_loc"], P["sd_scale"], _noise(noise, "eps_sd")).exp()
    guide_lp = {
        "mu_targets": tdist.Normal(P["mu_loc"], P["mu_scale"]).log_prob(mu_e).sum(),
        "sd_targets": tdist.LogNormal(P["sd_loc"], P["sd_scale"]).log_prob(sd_e).sum(),
    }
    # alpha_pi[~allele_mask] = eps, written in place on the constrained value
    alpha_pi = torch.where(data.allele_mask, P["alpha_pi"], torch.full_like(P["alpha_pi"], eps))
    rg = data.repguide_mask.unsqueeze(1)
    conc_q = (alpha_pi / alpha_pi.sum(-1)[:, None] * data.pi_a0[:, None])[None, None].expand(R, 1, -1, -1)
    pi = dirichlet_rsample(conc_q, _noise(noise, "pi"))
    guide_lp["pi"] = masked_sum(tdist.Dirichlet(conc_q, validate_args=False).log_prob(pi), rg)
    model_lp = {}
    pi_eff = pi
    if scale_by_accessibility:
        # fit_noise=~args.dont_fit_noise is always truthy in tiling (run.py:416, F5)
        q_noise = tdist.Normal(P["noise_loc"], P["noise_scale"])
        lpn = normal_rsample(P["noise_loc"], P["noise_scale"], _noise(noise, "eps_noise"))
        guide_lp["logit_pi_noise"] = q_noise.log_prob(lpn).sum()
        model_lp["logit_pi_noise"] = tdist.Normal(0.0, PI_NOISE_SD).log_prob(lpn).sum()
        pi_eff = scale_pi_by_accessibility(pi, data.guide_accessibility, lpn)
    p_mu, p_sd = _priors((E,), sd_scale, prior_params)
    model_lp["mu_targets"] = p_mu.log_prob(mu_e).sum()
    model_lp["sd_targets"] = p_sd.log_prob(sd_e).sum()
    mu_a, sd_a = allele_moments(data, mu_e, sd_e, sparse)
    mu = torch.cat([torch.zeros((G, 1)), mu_a], -1)
    sd = torch.cat([torch.ones((G, 1)), sd_a], -1)
    conc_p = (alpha_pi + eps / A) / (alpha_pi.sum(-1)[:, None] + eps) * data.pi_a0[:, None]
    conc_p = torch.where(conc_p < eps, torch.full_like(conc_p, eps), conc_p)
    conc_p = conc_p[None, None].expand(R, 1, -1, -1)
    model_lp["pi"] = masked_sum(tdist.Dirichlet(conc_p, validate_args=False).log_prob(pi), rg)
    model_lp["bulk_allele_count"] = masked_sum(
        tdist.Multinomial(probs=pi, validate_args=False).log_prob(data.allele_counts_control), rg
    )
    uq = data.upper_bounds[:, None, None].expand(-1, G, A)
    lq = data.lower_bounds[:, None, None].expand(-1, G, A)
    p_bin = std_normal_bin_prob(
        uq, lq, mu[None].expand(B, -1, -1), sd[None].expand(B, -1, -1),
        mask=data.allele_mask[None].expand(B, -1, -1),
    )
    expected_guide_p = (pi_eff.expand(R, B, -1, -1) * p_bin[None]).sum(-1)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch))
    return _finish(model_lp, guide_lp, record)


LOSSES = {
    "Normal": normal_loss,
    "ControlNormal": control_normal_loss,
    "MixtureNormal": mixture_normal_loss,
    "MultiMixtureNormal": multi_mixture_normal_loss,
}

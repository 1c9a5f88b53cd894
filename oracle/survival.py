"""One-particle negative ELBO of the survival-screen model families (oracle).

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates
``bean/model/survival_model.py``: the sorting models' Normal-CDF bin
probabilities are replaced by exponential growth ``exp(mu * t)`` over the
(max-normalised) timepoints, there is no ``sd`` latent, and the likelihood is the
same Dirichlet-Multinomial over conditions (there is no Negative-Binomial
likelihood anywhere in the reference).

Quirks reproduced:
* ``MixtureNormalModel``: the per-guide baseline ``mu_negctrl ~ N(m0, s0)`` is
  sampled in the model only (no guide site), i.e. a fresh prior draw each step
  (``survival_model.py:271-274``);
* ``initial_abundance ~ Dirichlet(q0)`` is *observed* in the model at the
  normalised t0 counts (306-311) but *sampled* in the guide (665-669): the draw
  is unused by the model yet contributes ``-log q``;
* the parameter ``q0`` is created by the guide with shape ``(G,)`` (660-664).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributions as tdist

from .elbo import (
    PI_NOISE_SD,
    _count_likelihoods,
    _finish,
    _noise,
    constrained,
    dirichlet_rsample,
    masked_sum,
    normal_rsample,
    scale_pi_by_accessibility,
)


def init_params(family: str, data, fit_noise: bool = True, scale_by_acc: bool = False):
    """Initial unconstrained parameters (``survival_model.py:630-648,660-711,747-752``)."""
    f = torch.float32
    G = data.n_guides
    if family == "ControlNormal":
        p = {k: torch.tensor(0.0, dtype=f) for k in ("mu_loc", "mu_scale")}
    elif family == "MultiMixtureNormal":
        E, A = data.n_edits, data.n_max_alleles
        p = {k: torch.zeros((E,), dtype=f) for k in ("mu_loc", "mu_scale")}
        ap = torch.zeros((G, A), dtype=f)
        ap[~data.allele_mask] = float(torch.log(torch.tensor(1e-5)))
        p["alpha_pi"] = ap
        if scale_by_acc:
            p["noise_loc"] = torch.zeros(G, dtype=f)
            p["noise_scale"] = torch.full((G,), PI_NOISE_SD, dtype=f).log()
        return {k: v.clone().requires_grad_(True) for k, v in p.items()}
    else:
        T = data.n_targets
        p = {k: torch.zeros((T, 1), dtype=f) for k in ("mu_loc", "mu_scale")}
    if family == "Normal":
        p["initial_abundance"] = (torch.ones(G, dtype=f) / G).log()
    if family == "MixtureNormal":
        p["q0"] = (torch.ones(G, dtype=f) / G).log()
        p["alpha_pi"] = torch.zeros((G, 2), dtype=f)
        if scale_by_acc and fit_noise:
            p["noise_loc"] = torch.zeros(G, dtype=f)
            p["noise_scale"] = torch.full((G,), PI_NOISE_SD, dtype=f).log()
    return {k: v.clone().requires_grad_(True) for k, v in p.items()}


def _constrained(params):
    out = constrained(params)
    if "initial_abundance" in params:
        out["initial_abundance"] = params["initial_abundance"].exp()
    return out


def _mu_prior(prior_params):
    if prior_params is not None and ("mu_loc" in prior_params or "mu_scale" in prior_params):
        return tdist.Normal(prior_params.get("mu_loc", 0.0), prior_params.get("mu_scale", 1.0))
    return tdist.Laplace(0.0, 1.0)


def control_normal_loss(data, params, noise=None, use_bcmatch=True, mask_thres=10, record=None):
    """``ControlNormalModel`` / ``ControlNormalGuide`` (133-212, 742-756)."""
    P = _constrained(params)
    R, B, G = data.n_reps, data.n_condits, data.n_guides
    mu_t = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    guide_lp = {"mu_targets": tdist.Normal(P["mu_loc"], P["mu_scale"]).log_prob(mu_t).sum()}
    model_lp = {"mu_targets": tdist.Normal(0.0, 1.0).log_prob(mu_t).sum()}
    mu = mu_t.repeat(G)
    p_time = torch.exp(mu[None].expand(B, -1) * data.timepoints[:, None].expand(-1, G))
    expected_guide_p = p_time[None].expand(R, -1, -1)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch, mask_thres))
    return _finish(model_lp, guide_lp, record)


def mixture_normal_loss(data, params, noise=None, use_bcmatch=True, scale_by_accessibility=False,
                        fit_noise=True, mask_thres=10, prior_params=None, mu_negctrl=(0.0, 0.1),
                        record=None):
    """``MixtureNormalModel`` / ``MixtureNormalGuide`` (215-424, 651-739)."""
    P = _constrained(params)
    R, B, G, T = data.n_reps, data.n_condits, data.n_guides, data.n_targets
    # ---- guide
    q0 = P["q0"]
    x0 = dirichlet_rsample(q0[None].expand(R, -1), _noise(noise, "initial_abundance"))
    guide_lp = {"initial_abundance": tdist.Dirichlet(q0, validate_args=False).log_prob(x0).sum()}
    mu_t = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    guide_lp["mu_targets"] = tdist.Normal(P["mu_loc"], P["mu_scale"]).log_prob(mu_t).sum()
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
        model_lp["logit_pi_noise"] = tdist.Normal(0.0, PI_NOISE_SD).log_prob(lpn).sum()
        pi_eff = scale_pi_by_accessibility(pi, data.guide_accessibility, lpn)
    # ---- model replay
    model_lp["mu_targets"] = _mu_prior(prior_params).log_prob(mu_t).sum()
    neg = tdist.Normal(mu_negctrl[0], mu_negctrl[1])
    u = _noise(noise, "mu_negctrl")
    if u is None:
        u = neg.sample((G,))
    u = u.to(neg.loc.dtype)
    model_lp["mu_negctrl"] = neg.log_prob(u).sum()
    mu_edit = torch.repeat_interleave(mu_t, data.target_lengths, dim=0)
    mu = torch.cat([u.unsqueeze(-1), mu_edit + u.unsqueeze(-1)], -1)  # (G, 2)
    x_t0 = data.X[:, 0, :] + 1
    obs0 = x_t0 / x_t0.sum(-1, keepdims=True)
    model_lp["initial_abundance"] = tdist.Dirichlet(q0, validate_args=False).log_prob(obs0).sum()
    rg = data.repguide_mask.unsqueeze(1)
    conc_p = pi_a_scaled[None, None].expand(R, 1, -1, -1)
    model_lp["pi"] = masked_sum(tdist.Dirichlet(conc_p, validate_args=False).log_prob(pi), rg)
    tc = data.control_timepoint
    n_c = len(tc)
    growth_c = torch.exp(mu[None, None].expand(R, n_c, -1, -1) * tc[None, :, None, None].expand(R, -1, G, 2))
    model_lp["control_allele_count"] = masked_sum(
        tdist.Multinomial(probs=pi.expand(-1, n_c, -1, -1) * growth_c, validate_args=False).log_prob(
            data.allele_counts_control),
        rg,
    )
    p_time = torch.exp(mu[None].expand(B, -1, -1) * data.timepoints[:, None, None].expand(-1, G, 1))
    expected_guide_p = (pi_eff.expand(-1, B, -1, -1) * p_time[None]).sum(-1)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch, mask_thres))
    return _finish(model_lp, guide_lp, record)


def normal_loss(data, params, noise=None, use_bcmatch=True, mask_thres=10, prior_params=None, record=None):
    """``NormalModel`` / ``NormalGuide`` (15-130, 629-648): the drawn initial guide
    abundance ``q_0`` multiplies the growth term inside the likelihood."""
    P = _constrained(params)
    R, B, G = data.n_reps, data.n_condits, data.n_guides
    ia = P["initial_abundance"]
    q_0 = dirichlet_rsample(ia[None].expand(R, -1), _noise(noise, "q_0"))
    mu_t = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    guide_lp = {
        "initial_guide_abundance": tdist.Dirichlet(ia, validate_args=False).log_prob(q_0).sum(),
        "mu_targets": tdist.Normal(P["mu_loc"], P["mu_scale"]).log_prob(mu_t).sum(),
    }
    prior_ia = torch.ones(G) / G  # float32 values, as the reference builds them
    if prior_params is not None and "initial_abundance" in prior_params:
        prior_ia = prior_params["initial_abundance"]
    prior_ia = prior_ia.to(ia.dtype)  # the float64 checking mode evaluates the same values in float64
    model_lp = {
        "mu_targets": _mu_prior(prior_params).log_prob(mu_t).sum(),
        "initial_guide_abundance": tdist.Dirichlet(prior_ia[None].expand(R, -1), validate_args=False)
        .log_prob(q_0).sum(),
    }
    mu = torch.repeat_interleave(mu_t, data.target_lengths, dim=0)
    idx = getattr(data, "negctrl_guide_idx", None)
    # `mu[data.negctrl_guide_idx, :] = 0.0` (59-60); with idx None this zeroes every row
    keep = torch.ones(G, 1, dtype=torch.bool)
    if idx is None:
        keep[:] = False
    else:
        keep[torch.as_tensor(idx, dtype=torch.int64)] = False
    mu = torch.where(keep, mu, torch.zeros_like(mu))
    p_time = torch.exp(mu[None].expand(B, -1, -1) * data.timepoints[:, None, None].expand(-1, G, 1))
    expected_guide_p = (p_time[None].expand(R, -1, -1, -1) * q_0[:, None, :, None].expand(-1, B, -1, -1)).sum(-1)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch, mask_thres))
    return _finish(model_lp, guide_lp, record)


def multi_mixture_normal_loss(data, params, noise=None, use_bcmatch=True, scale_by_accessibility=False,
                              fit_noise=True, mask_thres=10, prior_params=None, mu_negctrl=(0.0, 0.1),
                              record=None, eps=1e-5):
    """Tiling survival screens: ``MultiMixtureNormalModel`` / ``MultiMixtureNormalGuide``
    (427-626, 759-833).  Per-edit growth effects, allele = sum of its edits on top of the per-guide
    baseline ``mu_negctrl`` (model-only draw, as in ``MixtureNormalModel``); masked alleles get zero
    growth probability (561-567) but stay in the control Multinomial (535-548).  The guide declares
    an ``initial_abundance`` parameter it never uses (770-774): it has no gradient and is not part of
    the loss."""
    P = _constrained({k: v for k, v in params.items() if k != "initial_abundance"})
    R, B, G, A, E = data.n_reps, data.n_condits, data.n_guides, data.n_max_alleles, data.n_edits
    a2e = data.allele_to_edit_dense().to(P["mu_loc"].dtype)
    mu_e = normal_rsample(P["mu_loc"], P["mu_scale"], _noise(noise, "eps_mu"))
    guide_lp = {"mu_targets": tdist.Normal(P["mu_loc"], P["mu_scale"]).log_prob(mu_e).sum()}
    alpha_pi = torch.where(data.allele_mask, P["alpha_pi"], torch.full_like(P["alpha_pi"], eps))
    rg = data.repguide_mask.unsqueeze(1)
    conc_q = (alpha_pi / alpha_pi.sum(-1)[:, None] * data.pi_a0[:, None])[None, None].expand(R, 1, -1, -1)
    conc_q = conc_q.clamp(1e-5)
    pi = dirichlet_rsample(conc_q, _noise(noise, "pi"))
    guide_lp["pi"] = masked_sum(tdist.Dirichlet(conc_q, validate_args=False).log_prob(pi), rg)
    model_lp = {}
    pi_eff = pi
    if scale_by_accessibility:
        q_noise = tdist.Normal(P["noise_loc"], P["noise_scale"])
        lpn = normal_rsample(P["noise_loc"], P["noise_scale"], _noise(noise, "eps_noise"))
        guide_lp["logit_pi_noise"] = q_noise.log_prob(lpn).sum()
        model_lp["logit_pi_noise"] = tdist.Normal(0.0, PI_NOISE_SD).log_prob(lpn).sum()
        pi_eff = scale_pi_by_accessibility(pi, data.guide_accessibility, lpn)
    model_lp["mu_targets"] = _mu_prior(prior_params).log_prob(mu_e).sum()
    neg = tdist.Normal(mu_negctrl[0], mu_negctrl[1])
    u = _noise(noise, "mu_negctrl")
    if u is None:
        u = neg.sample((G,))
    u = u.to(neg.loc.dtype)
    model_lp["mu_negctrl"] = neg.log_prob(u).sum()
    mu_a = torch.matmul(a2e, mu_e)  # (G, A - 1)
    mu = torch.cat([u.unsqueeze(-1), u.unsqueeze(-1) + mu_a], -1)  # (G, A)
    conc_p = (alpha_pi + eps / A) / (alpha_pi.sum(-1)[:, None] + eps) * data.pi_a0[:, None]
    conc_p = torch.where(conc_p < eps, torch.full_like(conc_p, eps), conc_p)[None, None].expand(R, 1, -1, -1)
    model_lp["pi"] = masked_sum(tdist.Dirichlet(conc_p, validate_args=False).log_prob(pi), rg)
    tc = data.control_timepoint
    n_c = len(tc)
    growth_c = torch.exp(mu[None, None].expand(R, n_c, -1, -1) * tc[None, :, None, None].expand(R, -1, G, A))
    model_lp["control_allele_count"] = masked_sum(
        tdist.Multinomial(probs=pi.expand(-1, n_c, -1, -1) * growth_c, validate_args=False).log_prob(
            data.allele_counts_control),
        rg,
    )
    p_time = torch.exp(data.timepoints[:, None, None].expand(-1, G, 1) * mu[None].expand(B, -1, -1))
    p_time = p_time * data.allele_mask[None].expand(B, -1, -1)
    expected_guide_p = (pi_eff.expand(R, B, -1, -1) * p_time[None]).sum(-1)
    model_lp.update(_count_likelihoods(data, expected_guide_p, use_bcmatch, mask_thres))
    return _finish(model_lp, guide_lp, record)


LOSSES = {
    "MultiMixtureNormal": multi_mixture_normal_loss,
    "Normal": normal_loss,
    "ControlNormal": control_normal_loss,
    "MixtureNormal": mixture_normal_loss,
}

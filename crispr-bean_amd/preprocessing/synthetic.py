"""Deterministic synthetic screens of the BASELINE.json shapes.

There is no network for datasets, so ``bench.py`` and the parity tests draw
screens from the generative model itself: guides grouped
``guides_per_target`` per target, true effects ``mu ~ 0.9*delta_0 + 0.1*N(0,1)``,
editing rate ``pi ~ Beta(2,5)``, Dirichlet-Multinomial counts with the
reference's fallback precision trend ``a0 = exp(-1.510 + 0.7861 log n)``
(``bean/preprocessing/get_alpha0.py:105``).  The derived tensors (size factors,
``a0``, ``a0_bcmatch``, ``pi_a0``, masks) are then computed from the counts with
the same preprocessing the reference applies to a real screen, so the object
returned honours the full ``ScreenData`` attribute contract.

Sorting screens contain ``n_bins`` sorted samples plus one control ("bulk",
quantiles (0, 1)) sample per replicate, and - as in the reference, where the
control sample always stays inside ``screen_selected`` (``bean/cli/run.py:114``,
SURVEY.md F4) - the control is one of the ``n_condits = n_bins + 1`` conditions
of ``X``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
from scipy.special import ndtr, ndtri

from .alpha0 import fitted_alpha0, fitted_pi_alpha0, pred_alpha0
from .data_class import ScreenTensors

BASE_SEED = 20240501
DEFAULT_BINS = ((0.0, 0.2), (0.2, 0.4), (0.6, 0.8), (0.8, 1.0))


def _size_factor(X_gs: np.ndarray) -> np.ndarray:
    """Normalised column means (data_class.py:237-251). X_gs is (G, n_samples)."""
    sf = X_gs.mean(axis=0)
    return sf / sf.mean()


def _dirmult_counts(rng, n_total, alpha):
    """Counts (..., K) ~ DirMult(n_total (...,), alpha (..., K))."""
    g = rng.standard_gamma(alpha)
    g = np.maximum(g, 1e-300)
    p = g / g.sum(-1, keepdims=True)
    flat_p = p.reshape(-1, p.shape[-1])
    flat_n = n_total.reshape(-1)
    out = rng.multinomial(flat_n, flat_p)
    return out.reshape(p.shape)


def _target_layout(n_guides: int, guides_per_target: int):
    n_targets = -(-n_guides // guides_per_target)
    lengths = np.full(n_targets, guides_per_target, dtype=np.int64)
    lengths[-1] = n_guides - guides_per_target * (n_targets - 1)
    g2t = np.repeat(np.arange(n_targets), lengths)
    return n_targets, lengths, g2t


def make_sorting_variant_screen(
    n_guides: int = 50_000,
    n_reps: int = 5,
    bins=DEFAULT_BINS,
    guides_per_target: int = 5,
    depth_per_guide: float = 500.0,
    seed: int = BASE_SEED,
    with_accessibility: bool = False,
    frac_effect: float = 0.1,
    mask_fraction: float = 0.0,
) -> ScreenTensors:
    """Variant sorting screen (BASELINE configs 2/4 and the metric shape).

    Returns CPU tensors; call ``.to("cuda")`` to move them.
    ``mask_fraction`` > 0 knocks out that fraction of (rep, guide) pairs through
    ``repguide_mask`` and zeroes one sample via ``sample_mask`` to exercise the
    masking paths in tests.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    R, G = n_reps, n_guides
    lo = np.array([b[0] for b in bins] + [0.0])
    hi = np.array([b[1] for b in bins] + [1.0])
    # conditions sorted by (upper, lower) as data_class.py:948-964 does
    order = np.lexsort((lo, hi))
    lo, hi = lo[order], hi[order]
    B = len(lo)

    T, lengths, g2t = _target_layout(G, guides_per_target)
    mu_true = np.where(rng.random(T) < frac_effect, rng.normal(0.0, 1.0, T), 0.0)
    sd_true = np.ones(T)
    is_negctrl = rng.random(T) < 0.05
    mu_true[is_negctrl] = 0.0
    pi_true = rng.beta(2.0, 5.0, G)
    abundance = np.exp(rng.normal(0.0, 0.5, G))
    abundance /= abundance.mean()
    acc = np.exp(rng.normal(1.0, 0.8, G)) if with_accessibility else None
    pi_endo = pi_true
    if with_accessibility:
        # endogenous editing rate ~ reporter rate scaled by accessibility
        # (bean/model/utils.py:79-103), used only to generate data
        pi_endo = np.clip(pi_true * np.exp(-1.9458) * acc**0.2513, 1e-3, 1 - 1e-3)

    # bin probabilities of the two mixture components
    with np.errstate(invalid="ignore"):
        z_hi = np.where(hi >= 1.0, np.inf, ndtri(np.clip(hi, 1e-300, 1)))
        z_lo = np.where(lo <= 0.0, -np.inf, ndtri(np.clip(lo, 1e-300, 1)))
    p_wt = ndtr(z_hi) - ndtr(z_lo)  # (B,)
    mu_g, sd_g = mu_true[g2t], sd_true[g2t]
    p_ed = ndtr((z_hi[:, None] - mu_g[None, :]) / sd_g[None, :]) - ndtr(
        (z_lo[:, None] - mu_g[None, :]) / sd_g[None, :]
    )  # (B, G)
    e = (1 - pi_endo)[None, :] * p_wt[:, None] + pi_endo[None, :] * p_ed  # (B, G)

    depth = rng.uniform(0.7, 1.3, (R, B))  # per-sample sequencing depth factor
    # the control sample holds all cells (p = 1): sequence it as deep as one bin
    is_ctrl = (lo == 0.0) & (hi == 1.0)
    depth[:, is_ctrl] *= 0.2
    prop = e[None, :, :] * depth[:, :, None]  # (R, B, G)
    prop_g = prop / prop.sum(1, keepdims=True)
    n_rg = rng.poisson(depth_per_guide * abundance[None, :] * prop.sum(1) / 0.2)
    a0_true = np.exp(-1.510 + 0.7861 * np.log(np.maximum(n_rg, 1)))
    alpha = np.moveaxis(prop_g, 1, -1) * a0_true[:, :, None]  # (R, G, B)
    X = np.moveaxis(_dirmult_counts(rng, n_rg, alpha), -1, 1).astype(np.float64)
    X_bc = rng.binomial(X.astype(np.int64), 0.8).astype(np.float64)

    ctrl = int(np.nonzero(is_ctrl)[0][0])
    X_bc_ctrl = X_bc[:, ctrl : ctrl + 1, :]  # (R, 1, G)
    edited = rng.binomial(X_bc_ctrl.astype(np.int64), pi_true[None, None, :]).astype(
        np.float64
    )
    allele_counts_control = np.stack([X_bc_ctrl - edited, edited], axis=-1)

    sample_mask = np.ones((R, B), dtype=np.int64)
    repguide = np.ones((R, G), dtype=bool)
    if mask_fraction > 0:
        repguide &= rng.random((R, G)) >= mask_fraction
        sample_mask[R - 1, 0] = 0

    # size factors over all samples of the screen, as ScreenData.__init__ does
    sf = _size_factor(X.reshape(R * B, G).T).reshape(R, B)
    sf_bc = _size_factor(X_bc.reshape(R * B, G).T).reshape(R, B)

    a0, popt = fitted_alpha0(X, sf, sample_mask)
    a0_bc = pred_alpha0(X_bc, sf_bc, popt, sample_mask)
    pi_a0, _ = fitted_pi_alpha0(allele_counts_control, sf[:, ctrl : ctrl + 1])

    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32)
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)
    sm = torch.as_tensor(sample_mask)
    Xt, Xbt = f32(X), f32(X_bc)
    repguide_t = torch.as_tensor(repguide) & ~(Xt == 0).any(dim=1)
    data = ScreenTensors(
        n_reps=R,
        n_condits=B,
        n_guides=G,
        n_targets=T,
        n_max_alleles=2,
        X=Xt,
        X_masked=Xt * sm[:, :, None],
        X_bcmatch=Xbt,
        X_bcmatch_masked=Xbt * sm[:, :, None],
        X_control=Xt[:, ctrl : ctrl + 1, :].clone(),
        X_bcmatch_control=Xbt[:, ctrl : ctrl + 1, :].clone(),
        sample_mask=sm,
        control_sample_mask=sm[:, ctrl : ctrl + 1].clone(),
        repguide_mask=repguide_t,
        size_factor=f64(sf),
        size_factor_bcmatch=f64(sf_bc),
        size_factor_control=f64(sf[:, ctrl : ctrl + 1]),
        size_factor_bcmatch_control=f64(sf_bc[:, ctrl : ctrl + 1]),
        a0=f64(a0),
        a0_bcmatch=f64(a0_bc),
        pi_a0=f64(pi_a0),
        allele_counts_control=f32(allele_counts_control),
        upper_bounds=f64(hi),
        lower_bounds=f64(lo),
        target_lengths=torch.as_tensor(lengths),
        guide_accessibility=f64(acc) if acc is not None else None,
        popt=popt,
    )
    data.selection, data.library_design = "sorting", "variant"
    data.truth = {
        "mu": mu_true,
        "sd": sd_true,
        "pi": pi_true,
        "negctrl_target": is_negctrl,
    }
    data.negctrl_guide_idx = np.nonzero(is_negctrl[g2t])[0]
    data.validate()
    return data


def make_sorting_tiling_screen(
    n_guides: int = 50_000,
    n_reps: int = 5,
    n_max_alleles: int = 8,
    bins=DEFAULT_BINS,
    depth_per_guide: float = 500.0,
    seed: int = BASE_SEED + 3,
    with_accessibility: bool = False,
    edits_per_guide: float = 0.6,
    mask_fraction: float = 0.0,
    alleles_mean: float = 3.0,
) -> ScreenTensors:
    """Tiling sorting screen (BASELINE config 3): every guide produces up to
    ``n_max_alleles - 1`` edited alleles, each a set of 1-3 edits drawn from a
    window of 10 positions around the guide, so edits are shared by neighbouring
    guides.  ``allele_to_edit`` is returned in CSR form (rows = ``G * (A - 1)``
    allele slots; the reference builds the dense 0/1 tensor,
    ``data_class.py:656-699``)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R, G, A = n_reps, n_guides, n_max_alleles
    A1 = A - 1
    lo = np.array([b[0] for b in bins] + [0.0])
    hi = np.array([b[1] for b in bins] + [1.0])
    order = np.lexsort((lo, hi))
    lo, hi = lo[order], hi[order]
    B = len(lo)
    E = max(int(G * edits_per_guide) + 10, 12)
    mu_e = np.where(rng.random(E) < 0.1, rng.normal(0.0, 1.0, E), 0.0)
    sd_e = np.ones(E)
    n_al = np.minimum(1 + rng.poisson(alleles_mean, G), A1)  # alleles_mean >> 3: an unfiltered allele table
    base = np.minimum((np.arange(G) * (E - 10) / max(G, 1)).astype(np.int64), E - 10)
    ptr = np.zeros(G * A1 + 1, dtype=np.int64)
    idx = []
    allele_mask = np.zeros((G, A), dtype=bool)
    allele_mask[:, 0] = True
    mu_a = np.zeros((G, A))
    sd_a = np.ones((G, A))
    for g in range(G):
        seen = set()
        k = 0
        for _ in range(int(n_al[g])):
            ne = int(rng.integers(1, 4 if alleles_mean <= 10 else 5))
            es = tuple(sorted(base[g] + rng.choice(10, size=ne, replace=False)))
            if es in seen:
                continue
            seen.add(es)
            slot = g * A1 + k
            idx.extend(es)
            ptr[slot + 1] = len(es)
            allele_mask[g, k + 1] = True
            mu_a[g, k + 1] = mu_e[list(es)].sum()
            sd_a[g, k + 1] = np.sqrt((sd_e[list(es)] ** 2).sum())
            k += 1
    ptr = np.cumsum(ptr)
    idx = np.asarray(idx, dtype=np.int64)
    # allele frequencies: wild type heavy
    conc = np.where(allele_mask, 1.0, 0.0)
    conc[:, 0] = 5.0
    gam = rng.standard_gamma(np.where(conc > 0, conc, 1.0)) * (conc > 0)
    pi_true = gam / gam.sum(1, keepdims=True)
    abundance = np.exp(rng.normal(0.0, 0.5, G))
    abundance /= abundance.mean()
    acc = np.exp(rng.normal(1.0, 0.8, G)) if with_accessibility else None

    with np.errstate(invalid="ignore"):
        z_hi = np.where(hi >= 1.0, np.inf, ndtri(np.clip(hi, 1e-300, 1)))
        z_lo = np.where(lo <= 0.0, -np.inf, ndtri(np.clip(lo, 1e-300, 1)))
    p_bin = ndtr((z_hi[:, None, None] - mu_a[None]) / sd_a[None]) - ndtr(
        (z_lo[:, None, None] - mu_a[None]) / sd_a[None]
    )  # (B, G, A)
    e = (pi_true[None] * p_bin).sum(-1)  # (B, G)
    depth = rng.uniform(0.7, 1.3, (R, B))
    is_ctrl = (lo == 0.0) & (hi == 1.0)
    depth[:, is_ctrl] *= 0.2
    prop = e[None] * depth[:, :, None]
    prop_g = prop / prop.sum(1, keepdims=True)
    n_rg = rng.poisson(depth_per_guide * abundance[None, :] * prop.sum(1) / 0.2)
    a0_true = np.exp(-1.510 + 0.7861 * np.log(np.maximum(n_rg, 1)))
    alpha = np.moveaxis(prop_g, 1, -1) * a0_true[:, :, None]
    X = np.moveaxis(_dirmult_counts(rng, n_rg, alpha), -1, 1).astype(np.float64)
    X_bc = rng.binomial(X.astype(np.int64), 0.8).astype(np.float64)
    ctrl = int(np.nonzero(is_ctrl)[0][0])
    X_bc_ctrl = X_bc[:, ctrl, :].astype(np.int64)  # (R, G)
    ac = rng.multinomial(X_bc_ctrl.reshape(-1), np.tile(pi_true, (R, 1))).reshape(R, 1, G, A).astype(np.float64)

    sample_mask = np.ones((R, B), dtype=np.int64)
    repguide = np.ones((R, G), dtype=bool)
    if mask_fraction > 0:
        repguide &= rng.random((R, G)) >= mask_fraction
        sample_mask[R - 1, 0] = 0
    sf = _size_factor(X.reshape(R * B, G).T).reshape(R, B)
    sf_bc = _size_factor(X_bc.reshape(R * B, G).T).reshape(R, B)
    a0, popt = fitted_alpha0(X, sf, sample_mask)
    a0_bc = pred_alpha0(X_bc, sf_bc, popt, sample_mask)
    pi_a0, _ = fitted_pi_alpha0(ac, sf[:, ctrl : ctrl + 1])

    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32)
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)
    sm = torch.as_tensor(sample_mask)
    Xt, Xbt = f32(X), f32(X_bc)
    data = ScreenTensors(
        n_reps=R, n_condits=B, n_guides=G, n_targets=E, n_edits=E, n_max_alleles=A,
        X=Xt, X_masked=Xt * sm[:, :, None], X_bcmatch=Xbt, X_bcmatch_masked=Xbt * sm[:, :, None],
        X_control=Xt[:, ctrl : ctrl + 1, :].clone(), X_bcmatch_control=Xbt[:, ctrl : ctrl + 1, :].clone(),
        sample_mask=sm, control_sample_mask=sm[:, ctrl : ctrl + 1].clone(),
        repguide_mask=torch.as_tensor(repguide) & ~(Xt == 0).any(dim=1),
        size_factor=f64(sf), size_factor_bcmatch=f64(sf_bc), size_factor_control=f64(sf[:, ctrl : ctrl + 1]),
        size_factor_bcmatch_control=f64(sf_bc[:, ctrl : ctrl + 1]),
        a0=f64(a0), a0_bcmatch=f64(a0_bc), pi_a0=f64(pi_a0), allele_counts_control=f32(ac),
        upper_bounds=f64(hi), lower_bounds=f64(lo), target_lengths=None,
        guide_accessibility=f64(acc) if acc is not None else None,
        allele_mask=torch.as_tensor(allele_mask),
        a2e_ptr=torch.as_tensor(ptr.astype(np.int32)), a2e_idx=torch.as_tensor(idx.astype(np.int32)),
        popt=popt,
    )
    data.selection, data.library_design = "sorting", "tiling"
    data.truth = {"mu_edits": mu_e, "pi": pi_true}
    data.validate()
    return data


def make_survival_variant_screen(
    n_guides: int = 100_000,
    n_reps: int = 3,
    times=(0.0, 3.0, 6.0, 9.0, 12.0, 15.0),
    control_index: int = 1,
    guides_per_target: int = 5,
    depth_per_guide: float = 500.0,
    seed: int = BASE_SEED + 5,
    with_accessibility: bool = False,
    frac_effect: float = 0.1,
    mask_fraction: float = 0.0,
) -> ScreenTensors:
    """Variant survival / proliferation screen (BASELINE config 5): guide
    abundance grows as ``exp(mu * t)`` over timepoints normalised by the last one
    (``data_class.py:1034-1053``); the control condition is timepoint
    ``control_index`` and, as in the reference, stays among the ``n_condits``
    timepoints of ``X``.  The likelihood is Dirichlet-Multinomial over
    timepoints (the reference has no Negative-Binomial model)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R, G = n_reps, n_guides
    t = np.asarray(times, dtype=np.float64)
    t = t / t.max()
    B = len(t)
    T, lengths, g2t = _target_layout(G, guides_per_target)
    mu_true = np.where(rng.random(T) < frac_effect, rng.normal(0.0, 1.0, T), 0.0)
    is_negctrl = rng.random(T) < 0.05
    mu_true[is_negctrl] = 0.0
    pi_true = rng.beta(2.0, 5.0, G)
    base = rng.normal(0.0, 0.1, G)  # per-guide baseline growth of unedited cells
    abundance = np.exp(rng.normal(0.0, 0.5, G))
    abundance /= abundance.mean()
    acc = np.exp(rng.normal(1.0, 0.8, G)) if with_accessibility else None
    mu_g = mu_true[g2t]
    growth = (1 - pi_true)[None, :] * np.exp(base[None, :] * t[:, None]) + pi_true[None, :] * np.exp(
        (base + mu_g)[None, :] * t[:, None]
    )  # (B, G)
    depth = rng.uniform(0.7, 1.3, (R, B))
    prop = growth[None] * depth[:, :, None]
    prop_g = prop / prop.sum(1, keepdims=True)
    n_rg = rng.poisson(depth_per_guide * abundance[None, :] * prop.sum(1))
    a0_true = np.exp(-1.510 + 0.7861 * np.log(np.maximum(n_rg, 1)))
    alpha = np.moveaxis(prop_g, 1, -1) * a0_true[:, :, None]
    X = np.moveaxis(_dirmult_counts(rng, n_rg, alpha), -1, 1).astype(np.float64)
    X_bc = rng.binomial(X.astype(np.int64), 0.8).astype(np.float64)
    c = control_index
    X_bc_ctrl = X_bc[:, c : c + 1, :]
    # edited fraction at the control timepoint reflects selection up to t_c
    w1 = pi_true * np.exp((base + mu_g) * t[c])
    w0 = (1 - pi_true) * np.exp(base * t[c])
    edited = rng.binomial(X_bc_ctrl.astype(np.int64), (w1 / (w0 + w1))[None, None, :]).astype(np.float64)
    ac = np.stack([X_bc_ctrl - edited, edited], axis=-1)

    sample_mask = np.ones((R, B), dtype=np.int64)
    repguide = np.ones((R, G), dtype=bool)
    if mask_fraction > 0:
        repguide &= rng.random((R, G)) >= mask_fraction
        sample_mask[R - 1, B - 1] = 0
    sf = _size_factor(X.reshape(R * B, G).T).reshape(R, B)
    sf_bc = _size_factor(X_bc.reshape(R * B, G).T).reshape(R, B)
    a0, popt = fitted_alpha0(X, sf, sample_mask)
    a0_bc = pred_alpha0(X_bc, sf_bc, popt, sample_mask)
    pi_a0, _ = fitted_pi_alpha0(ac, sf[:, c : c + 1])

    f32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32)
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)
    sm = torch.as_tensor(sample_mask)
    Xt, Xbt = f32(X), f32(X_bc)
    data = ScreenTensors(
        n_reps=R, n_condits=B, n_guides=G, n_targets=T, n_max_alleles=2,
        X=Xt, X_masked=Xt * sm[:, :, None], X_bcmatch=Xbt, X_bcmatch_masked=Xbt * sm[:, :, None],
        X_control=Xt[:, c : c + 1, :].clone(), X_bcmatch_control=Xbt[:, c : c + 1, :].clone(),
        sample_mask=sm, control_sample_mask=sm[:, c : c + 1].clone(),
        repguide_mask=torch.as_tensor(repguide) & ~(Xt == 0).any(dim=1),
        size_factor=f64(sf), size_factor_bcmatch=f64(sf_bc), size_factor_control=f64(sf[:, c : c + 1]),
        size_factor_bcmatch_control=f64(sf_bc[:, c : c + 1]),
        a0=f64(a0), a0_bcmatch=f64(a0_bc), pi_a0=f64(pi_a0), allele_counts_control=f32(ac),
        timepoints=f64(t), control_timepoint=f64(t[c : c + 1]),
        upper_bounds=None, lower_bounds=None, target_lengths=torch.as_tensor(lengths),
        guide_accessibility=f64(acc) if acc is not None else None, popt=popt,
    )
    data.selection, data.library_design = "survival", "variant"
    data.truth = {"mu": mu_true, "pi": pi_true, "negctrl_target": is_negctrl}
    data.negctrl_guide_idx = np.nonzero(is_negctrl[g2t])[0]
    data.validate()
    return data


def make_survival_tiling_screen(n_guides: int = 2000, n_reps: int = 3, times=(0.0, 7.0, 14.0),
                                control_index: int = 1, seed: int = BASE_SEED + 6, **tiling_kw) -> ScreenTensors:
    """Tiling survival screen for parity tests: the allele / edit structure, allele counts and guide
    counts of :func:`make_sorting_tiling_screen` with its conditions relabelled as timepoints
    (normalised by the last one, ``data_class.py:1034-1053``) and the control condition moved to
    timepoint ``control_index``.  The counts are not drawn from a growth model - the tiling survival
    tests compare arithmetic, not recovery."""
    B = len(times)
    bins = tuple((i / B, (i + 1) / B) for i in range(B - 1))  # B - 1 sort bins + the control pseudo-bin
    data = make_sorting_tiling_screen(n_guides, n_reps, seed=seed, bins=bins, **tiling_kw)
    assert data.n_condits == B, (data.n_condits, B)
    t = np.asarray(times, dtype=np.float64)
    data.selection = "survival"
    data.timepoints = torch.as_tensor(t / t.max())
    data.control_timepoint = data.timepoints[control_index:control_index + 1].clone()
    data.upper_bounds = data.lower_bounds = None
    return data


def variant_reporter_screen(data: ScreenTensors):
    """The ``ReporterScreen`` a variant sorting ``ScreenTensors`` of this module would have come from: counts as a
    (guides x samples) matrix with layers ``X_bcmatch`` and ``edits``, a guide table (``target``, ``target_group``
    with the negative controls, ``accessibility`` when the screen has it) and a sample table (``replicate``,
    ``condition``, quantiles, ``mask``) - what ``bean run sorting variant`` reads from an ``.h5ad``.  For the
    ``bench.py`` leg that times the whole command on the one shape the reference publishes a run time for
    (README.md:83) and for CLI tests on screens larger than the reference's 30-guide fixture."""
    import pandas as pd

    from ..framework.ReporterScreen import ReporterScreen

    R, B, G = data.n_reps, data.n_condits, data.n_guides
    lo, hi = data.lower_bounds.numpy(), data.upper_bounds.numpy()
    names = []
    for l, h in zip(lo, hi):
        names.append("bulk" if (l == 0.0 and h == 1.0) else f"q{int(round(l * 100)):02d}_{int(round(h * 100)):02d}")
    samples = pd.DataFrame({
        "replicate": [f"rep{r + 1}" for r in range(R) for _ in range(B)],
        "condition": [names[b] for _ in range(R) for b in range(B)],
        "lower_quantile": [float(lo[b]) for _ in range(R) for b in range(B)],
        "upper_quantile": [float(hi[b]) for _ in range(R) for b in range(B)],
        "mask": 1,
    }, index=[f"rep{r + 1}_{names[b]}" for r in range(R) for b in range(B)])
    to_gs = lambda t: t.double().numpy().reshape(R * B, G).T.copy()  # (R, B, G) -> (G, R * B)
    X, Xbc = to_gs(data.X), to_gs(data.X_bcmatch)
    edits = np.zeros_like(X)
    ctrl = names.index("bulk")
    edited = data.allele_counts_control[:, 0, :, 1].double().numpy()  # (R, G): edited reads of the bulk sample
    for r in range(R):
        edits[:, r * B + ctrl] = edited[r]
    g2t = data.guide_to_target.numpy()
    neg = np.asarray(data.truth["negctrl_target"])[g2t]
    width = len(str(int(g2t.max())))
    guides = pd.DataFrame({
        "target": [f"t{t:0{width}d}" for t in g2t],
        "target_group": np.where(neg, "NegCtrl", "Variant"),
    }, index=pd.Index([f"g{i:0{len(str(G))}d}" for i in range(G)], name="name"))
    if data.guide_accessibility is not None:
        guides["accessibility"] = data.guide_accessibility.numpy()
    return ReporterScreen(X.astype(np.float32), guides, samples, {"X_bcmatch": Xbc, "edits": edits},
                          {"target_base_changes": "A>G", "tiling": False})

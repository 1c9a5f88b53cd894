"""Result tables: fitted parameters -> ``bean_element_result.*.csv`` /
``bean_sgRNA_result.*.csv``.

Restates ``bean/model/readwrite.py:9-246`` (same function names, arguments,
column names and row order) so that the HIP fit can be consumed exactly like the
reference's output.  Pinned by fixtures produced with the reference's own file
(``tests/golden/make_readwrite_golden.py``).  Quirk kept: ``"negctrl" in
param_hist_dict.keys()`` is never true, so the ``_adj`` columns always derive
from the unscaled ``mu`` / ``mu_sd``.
"""
from __future__ import annotations

from statistics import NormalDist
from typing import List, Optional, Sequence, Union

import numpy as np
import pandas as pd
from scipy.special import expit, logit
from scipy.stats import norm

ACC_A, ACC_B = 0.2513, -1.9458  # bean/model/readwrite.py:221-222


def get_novl(df: pd.DataFrame, mu_col: str, mu_sd_col: str) -> pd.Series:
    """1 - overlap of N(mu, mu_sd) with the standard normal, row by row."""
    std = NormalDist(mu=0, sigma=1)
    return df.apply(lambda r: 1 - NormalDist(mu=r[mu_col], sigma=r[mu_sd_col]).overlap(std), axis=1)


def get_quantile(mu, sd, q):
    return norm(mu, sd).ppf(q)


def add_credible_interval(df: pd.DataFrame, mu_col: str, mu_sd_col: str, alpha: float = 0.05) -> pd.DataFrame:
    out = df.copy()
    out[f"CI[{alpha/2}"] = get_quantile(out[mu_col], out[mu_sd_col], alpha / 2)
    out[f"{1-alpha/2}]"] = get_quantile(out[mu_col], out[mu_sd_col], 1 - alpha / 2)
    return out


def adjust_normal_params_by_control(param_df: pd.DataFrame, sd0: float, suffix: str = "_adj",
                                    mu_adjusted_col="mu", mu_sd_adjusted_col="mu_sd", mu0: float = 0.0):
    """Rescale the z-scores by the spread ``sd0`` of the negative-control z-scores."""
    param_df[f"mu{suffix}"] = param_df[mu_adjusted_col] - mu0
    param_df[f"mu_sd{suffix}"] = param_df[mu_sd_adjusted_col] * sd0
    param_df[f"mu_z{suffix}"] = param_df[f"mu{suffix}"] / param_df[f"mu_sd{suffix}"]
    param_df[f"novl{suffix}"] = get_novl(param_df, f"mu{suffix}", f"mu_sd{suffix}")
    return param_df


def _np(t):
    return t.detach().cpu().numpy()


def _scale_edited_pi(pi, guide_accessibility, a: float = ACC_A, b: float = ACC_B):
    return pi * np.exp(b) * guide_accessibility**a


def _add_noise_to_pi(pi, fitted_noise_logit):
    return expit(logit(pi.clip(min=1e-3, max=1 - 1e-3)) + fitted_noise_logit).clip(min=1e-3, max=1 - 1e-3)


def _scale_pi(pi, guide_acc, fitted_noise_logit=None):
    scaled = _scale_edited_pi(pi, guide_acc)
    return scaled if fitted_noise_logit is None else _add_noise_to_pi(scaled, fitted_noise_logit)


def _fitted_columns(P, sd_is_fitted: bool, covariates: Optional[List[str]]) -> pd.DataFrame:
    """One row per target: posterior mean / spread of mu, its z-score, the fitted sd, and - with sample covariates - the
    covariate-shifted copies of the three (variances add)."""
    loc = P["mu_loc"]
    if loc.dim() not in (1, 2):
        raise ValueError(f'`mu_loc` has invalid shape of {loc.shape}')
    first = (lambda arr: arr[:, 0]) if loc.dim() == 2 else (lambda arr: arr)
    centre, spread = first(_np(loc)), first(_np(P["mu_scale"]))
    table = {"mu": centre, "mu_sd": spread, "mu_z": centre / spread}
    if sd_is_fitted:
        table["sd"] = first(_np(P["sd_loc"].detach().exp()))
    if covariates is not None:
        assert "mu_cov_loc" in P and "mu_cov_scale" in P, P.keys()
        shift, shift_sd = _np(P["mu_cov_loc"]), _np(P["mu_cov_scale"])
        for i, name in enumerate(covariates):
            m = centre + shift[i]
            s_ = np.sqrt(spread**2 + shift_sd[i] ** 2)
            table.update({f"mu_{name}": m, f"mu_sd_{name}": s_, f"mu_z_{name}": m / s_})
    return pd.DataFrame(table)


def _rescale_by_control_fit(table: pd.DataFrame, negctrl_params, sd_is_fitted: bool,
                            covariates: Optional[List[str]]) -> None:
    """`_scaled` columns: the targets' posteriors in units of the common negative-control distribution (a separate
    ControlNormal fit: its mean and, when sd is fitted, its sd).  In place."""
    print("Normalizing with common negative control distribution")
    centre0 = _np(negctrl_params["mu_loc"]).mean()
    unit = _np(negctrl_params["sd_loc"].detach().exp()) if sd_is_fitted else 1.0
    print(f"Fitted mu0={centre0}" + (f", sd0={unit}." if sd_is_fitted else ""))
    table["mu_scaled"] = (table["mu"].values - centre0) / unit
    table["mu_sd_scaled"] = table["mu_sd"].values / unit
    table["mu_z_scaled"] = table.mu_scaled / table.mu_sd_scaled
    if sd_is_fitted:
        table["sd_scaled"] = table["sd"].values / unit
    table["novl_scaled"] = get_novl(table, "mu_scaled", "mu_sd_scaled")
    for name in covariates or ():
        table[f"mu_{name}_scaled"] = (table[f"mu_{name}"] - centre0) / unit
        table[f"mu_sd_{name}_scaled"] = table[f"mu_sd_{name}"] / unit
        table[f"mu_z_{name}_scaled"] = table[f"mu_{name}_scaled"] / table["mu_sd_scaled"]


def _ranked(table: pd.DataFrame, z_col: str) -> pd.DataFrame:
    """Rows by decreasing |z| (the reference's order: argsort of -|z|, missing values as pandas places them)."""
    return table.iloc[(-table[z_col].abs()).argsort()]


def _calibrated_by_negatives(table: pd.DataFrame, negatives, covariates: Optional[List[str]], from_scaled: bool) -> pd.DataFrame:
    """`_adj` columns: posterior spreads stretched by the spread of the negative-control VARIANTS' z-scores (a zero-mean
    normal fitted to them), then credible intervals and the |z| order.  Fewer than ten negatives: no calibration."""
    if len(negatives) < 10:
        print("Cannot adjust confidence by negative control due to too small number "
              f"({len(negatives)}) of negatives.")
        return _ranked(add_credible_interval(table, "mu", "mu_sd"), "mu_z")
    controls = table.iloc[negatives]
    if "mu_z_scaled" in controls.columns:
        print("Using mu_z_scaled for normalization input..")
        z_centre, z_spread = norm.fit(controls.mu_z_scaled, floc=0)
    else:
        z_centre, z_spread = norm.fit(controls.mu_z, floc=0)
    pick = (lambda base: base + "_scaled") if from_scaled else (lambda base: base)
    table = adjust_normal_params_by_control(table, z_spread, suffix="_adj", mu_adjusted_col=pick("mu"),
                                            mu_sd_adjusted_col=pick("mu_sd"), mu0=z_centre)
    table = _ranked(add_credible_interval(table, "mu_adj", "mu_sd_adj"), "mu_z_adj")
    for name in covariates or ():
        table = adjust_normal_params_by_control(table, z_spread, suffix=f"_{name}_adj",
                                                mu_adjusted_col=pick(f"mu_{name}"), mu_sd_adjusted_col=pick(f"mu_sd_{name}"))
        table = add_credible_interval(table, f"mu_{name}_adj", f"mu_sd_{name}_adj")
    return table


def _guide_editing_columns(guide_info_df: pd.DataFrame, P, guide_acc) -> None:
    """The sgRNA table's accessibility columns (in place): the fitted editing rate of every guide - the edited share of
    its `alpha_pi` - scaled by accessibility as the model scales it."""
    if guide_acc is None:
        return
    if "alpha_pi" in P.keys():
        conc = _np(P["alpha_pi"])
        edited_share = conc[..., 1:].sum(axis=1) / conc.sum(axis=1)
    else:
        edited_share = 1.0
    logit_noise = _np(P["noise_scale"]) if "noise_scale" in P.keys() else None
    guide_info_df.insert(1, "accessibility", guide_acc)
    guide_info_df.insert(2, "scaled_edit_eff", _scale_pi(edited_share, guide_acc, fitted_noise_logit=logit_noise))


def write_result_table(
    target_info_df: pd.DataFrame,
    guide_info_df: pd.DataFrame,
    param_hist_dict,
    model_label: str,
    prefix: str = "",
    suffix: str = "",
    negctrl_params=None,
    adjust_confidence_by_negative_control: bool = True,
    adjust_confidence_negatives: Optional[np.ndarray] = None,
    guide_acc: Optional[Sequence] = None,
    sd_is_fitted: bool = True,
    sample_covariates: Optional[List[str]] = None,
    return_result: bool = False,
    is_survival_screen: bool = False,
) -> Union[pd.DataFrame, None]:
    """Combine target information and fitted scores into the element table (written
    or returned) and write the sgRNA table (``bean/model/readwrite.py:49-215``: same arguments, columns, row order
    and files; the steps are the helpers above)."""
    fitted = _fitted_columns(param_hist_dict, sd_is_fitted, sample_covariates)
    if negctrl_params is not None:
        _rescale_by_control_fit(fitted, negctrl_params, sd_is_fitted, sample_covariates)
    element = pd.concat([target_info_df.reset_index(), fitted.reset_index(drop=True)], axis=1)
    if adjust_confidence_by_negative_control:
        assert adjust_confidence_negatives is not None
        # (the reference asks the PARAMETER STORE for a "negctrl" key, which it never has: the `_adj` columns always
        # derive from the unscaled mu / mu_sd - SURVEY.md Appendix C item 4; kept, and pinned by the reference's own run)
        element = _calibrated_by_negatives(element, adjust_confidence_negatives, sample_covariates,
                                           from_scaled="negctrl" in param_hist_dict.keys())
    else:
        element = _ranked(add_credible_interval(element, "mu", "mu_sd"), "mu_z")
    _guide_editing_columns(guide_info_df, param_hist_dict, guide_acc)
    guide_info_df.to_csv(f"{prefix}bean_sgRNA_result.{model_label}{suffix}.csv")
    if return_result:
        return element
    element.to_csv(f"{prefix}bean_element_result.{model_label}{suffix}.csv")

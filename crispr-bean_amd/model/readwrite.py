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
    or returned) and write the sgRNA table (``bean/model/readwrite.py:49-215``)."""
    P = param_hist_dict
    ndim = P["mu_loc"].dim()
    if ndim not in (1, 2):
        raise ValueError(f'`mu_loc` has invalid shape of {P["mu_loc"].shape}')
    col = (lambda a: a[:, 0]) if ndim == 2 else (lambda a: a)
    mu, mu_sd = col(_np(P["mu_loc"])), col(_np(P["mu_scale"]))
    cols = {"mu": mu, "mu_sd": mu_sd, "mu_z": mu / mu_sd}
    sd = None
    if sd_is_fitted:
        sd = col(_np(P["sd_loc"].detach().exp()))
        cols["sd"] = sd
    if sample_covariates is not None:
        assert "mu_cov_loc" in P and "mu_cov_scale" in P, P.keys()
        cov_loc, cov_scale = _np(P["mu_cov_loc"]), _np(P["mu_cov_scale"])
        for i, name in enumerate(sample_covariates):
            cols[f"mu_{name}"] = mu + cov_loc[i]
            cols[f"mu_sd_{name}"] = np.sqrt(mu_sd**2 + cov_scale[i] ** 2)
            cols[f"mu_z_{name}"] = cols[f"mu_{name}"] / cols[f"mu_sd_{name}"]
    fit_df = pd.DataFrame(cols)

    if negctrl_params is not None:
        print("Normalizing with common negative control distribution")
        mu0 = _np(negctrl_params["mu_loc"]).mean()
        sd0 = _np(negctrl_params["sd_loc"].detach().exp()) if sd_is_fitted else 1.0
        print(f"Fitted mu0={mu0}" + (f", sd0={sd0}." if sd_is_fitted else ""))
        fit_df["mu_scaled"] = (mu - mu0) / sd0
        fit_df["mu_sd_scaled"] = mu_sd / sd0
        fit_df["mu_z_scaled"] = fit_df.mu_scaled / fit_df.mu_sd_scaled
        if sd_is_fitted:
            fit_df["sd_scaled"] = sd / sd0
        fit_df["novl_scaled"] = get_novl(fit_df, "mu_scaled", "mu_sd_scaled")
        if sample_covariates is not None:
            for name in sample_covariates:
                fit_df[f"mu_{name}_scaled"] = (fit_df[f"mu_{name}"] - mu0) / sd0
                fit_df[f"mu_sd_{name}_scaled"] = fit_df[f"mu_sd_{name}"] / sd0
                fit_df[f"mu_z_{name}_scaled"] = fit_df[f"mu_{name}_scaled"] / fit_df["mu_sd_scaled"]

    fit_df = pd.concat([target_info_df.reset_index(), fit_df.reset_index(drop=True)], axis=1)

    by_abs_z = lambda df, c: df.iloc[(-df[c].abs()).argsort()]
    scaled_inputs = "negctrl" in P.keys()  # never true for a param store (kept as in the reference)
    if adjust_confidence_by_negative_control:
        assert adjust_confidence_negatives is not None
        if len(adjust_confidence_negatives) < 10:
            print("Cannot adjust confidence by negative control due to too small number "
                  f"({len(adjust_confidence_negatives)}) of negatives.")
            fit_df = by_abs_z(add_credible_interval(fit_df, "mu", "mu_sd"), "mu_z")
        else:
            ncvar = fit_df.iloc[adjust_confidence_negatives]
            if "mu_z_scaled" in ncvar.columns:
                print("Using mu_z_scaled for normalization input..")
                z_mean, z_std = norm.fit(ncvar.mu_z_scaled, floc=0)
            else:
                z_mean, z_std = norm.fit(ncvar.mu_z, floc=0)
            fit_df = adjust_normal_params_by_control(
                fit_df, z_std, suffix="_adj",
                mu_adjusted_col="mu_scaled" if scaled_inputs else "mu",
                mu_sd_adjusted_col="mu_sd_scaled" if scaled_inputs else "mu_sd",
                mu0=z_mean,
            )
            fit_df = by_abs_z(add_credible_interval(fit_df, "mu_adj", "mu_sd_adj"), "mu_z_adj")
            if sample_covariates is not None:
                for name in sample_covariates:
                    fit_df = adjust_normal_params_by_control(
                        fit_df, z_std, suffix=f"_{name}_adj",
                        mu_adjusted_col=f"mu_{name}_scaled" if scaled_inputs else f"mu_{name}",
                        mu_sd_adjusted_col=f"mu_sd_{name}_scaled" if scaled_inputs else f"mu_sd_{name}",
                    )
                    fit_df = add_credible_interval(fit_df, f"mu_{name}_adj", f"mu_sd_{name}_adj")
    else:
        fit_df = by_abs_z(add_credible_interval(fit_df, "mu", "mu_sd"), "mu_z")

    # sgRNA table
    if "alpha_pi" in P.keys():
        a_fitted = _np(P["alpha_pi"])
        pi = a_fitted[..., 1:].sum(axis=1) / a_fitted.sum(axis=1)
    else:
        pi = 1.0
    if guide_acc is not None:
        guide_info_df.insert(1, "accessibility", guide_acc)
        noise = _np(P["noise_scale"]) if "noise_scale" in P.keys() else None
        guide_info_df.insert(2, "scaled_edit_eff", _scale_pi(pi, guide_acc, fitted_noise_logit=noise))
    guide_info_df.to_csv(f"{prefix}bean_sgRNA_result.{model_label}{suffix}.csv")
    if return_result:
        return fit_df
    fit_df.to_csv(f"{prefix}bean_element_result.{model_label}{suffix}.csv")

"""Quality masks of a ReporterScreen, as ``bean qc`` computes them.

The reference runs a notebook (``bean/notebooks/sample_quality_report.ipynb``, executed by
``bean/cli/qc.py:9-58`` through papermill) whose cells 38-47 turn per-sample quality metrics into
``samples["mask"]`` and outlier guides into ``uns["repguide_mask"]`` (``bean/qc/guide_qc.py:5-46``);
``bean run`` then reads both (``bean/preprocessing/data_class.py:138-187``).  The thresholds and the
masking logic below follow those cells.  The METRICS themselves come from ``perturb-tools``
(``pt.qc.plot_correlation`` -> ``median_corr_X``, ``pt.qc.plot_lfc_correlation`` ->
``median_lfc_corr.<c1>_<c2>``, ``perturb_tools._qc.qc.get_outlier_guides``), a dependency that is not
vendored in the reference and absent here: they are restated from that package's published behaviour
and are NOT pinned by a reference-held fixture (stated in DESIGN.md):

* ``median_corr_X``: Spearman correlation of the log-normalised counts (log2(CPM + 1)) between
  samples; per sample the median over the OTHER samples;
* ``median_lfc_corr``: per-replicate log fold change ``cond1 - cond2`` of the selected (positive
  control) guides, Spearman correlation between replicates; per replicate the median over the other
  replicates, given to each of its samples;
* outlier guides: within a condition, a guide is an outlier in a sample when its RPM exceeds the
  median of that guide over the condition's samples by more than ``mad_z_thres`` median absolute
  deviations (scaled by 1.4826) and is larger than ``abs_RPM_thres``.

Editing rates are recomputed from the ``edits`` / ``X_bcmatch`` layers
(``ReporterScreen.get_guide_edit_rate``, reference ``ReporterScreen.py:448-530``); re-deriving the
``edits`` layer from the allele table (notebook cell 27) belongs to the allele-calling side of the
pipeline and is not done here (the notebook's ``--dont-recalculate-edits`` behaviour).
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import pandas as pd
from scipy.stats import spearmanr


def _rc_labels(samples: pd.DataFrame, replicate_col: Union[str, Sequence[str]]) -> pd.Series:
    if isinstance(replicate_col, str):
        return samples[replicate_col].astype(str)
    return samples[list(replicate_col)].astype(str).agg(".".join, axis=1)


def sample_count_correlation(screen) -> pd.Series:
    """median_corr_X of every sample."""
    ln = screen.log_norm(1.0)
    n = ln.shape[1]
    if n < 2:
        return pd.Series(np.nan, index=screen.samples.index)
    with np.errstate(all="ignore"):
        corr = np.atleast_2d(spearmanr(ln, nan_policy="omit")[0])
    if corr.shape != (n, n):  # two samples: spearmanr returns a scalar
        c = float(corr.reshape(-1)[0])
        corr = np.array([[1.0, c], [c, 1.0]])
    corr = corr.copy()
    np.fill_diagonal(corr, np.nan)
    with np.errstate(all="ignore"):
        return pd.Series(np.nanmedian(corr, axis=1), index=screen.samples.index)


def replicate_lfc_correlation(screen, guide_sel: np.ndarray, cond1: str, cond2: str,
                              replicate_col: Union[str, Sequence[str]], condition_col: str) -> pd.Series:
    """median_lfc_corr.<cond1>_<cond2> of every sample (NaN where its replicate lacks a condition)."""
    reps = _rc_labels(screen.samples, replicate_col)
    cond = screen.samples[condition_col].astype(str)
    ln = screen.log_norm(1.0)
    lfc = {}
    for rep in pd.unique(reps):
        a = np.where((reps == rep).values & (cond == cond1).values)[0]
        b = np.where((reps == rep).values & (cond == cond2).values)[0]
        if len(a) == 1 and len(b) == 1:
            lfc[rep] = (ln[:, a[0]] - ln[:, b[0]])[guide_sel]
    out = pd.Series(np.nan, index=screen.samples.index)
    names = list(lfc)
    if len(names) < 2 or int(np.sum(guide_sel)) < 3:
        return out
    mat = np.column_stack([lfc[r] for r in names])
    with np.errstate(all="ignore"):
        corr = np.atleast_2d(spearmanr(mat, nan_policy="omit")[0])
    if corr.shape != (len(names), len(names)):
        c = float(corr.reshape(-1)[0])
        corr = np.array([[1.0, c], [c, 1.0]])
    corr = corr.copy()
    np.fill_diagonal(corr, np.nan)
    with np.errstate(all="ignore"):
        med = dict(zip(names, np.nanmedian(corr, axis=1)))
    return reps.map(med).astype(float)


def outlier_guides_and_mask(screen, condition_col: str, replicate_col: Union[str, Sequence[str]] = "replicate",
                            mad_z_thres: float = 5.0, abs_RPM_thres: float = 10000.0):
    """(outlier table with columns name / sample / replicate, mask (guides x replicates) of 0/1) -
    ``bean/qc/guide_qc.py:5-46``."""
    X = np.asarray(screen.X, dtype=np.float64)
    with np.errstate(all="ignore"):
        rpm = X / X.sum(axis=0, keepdims=True) * 1e6
    cond = screen.samples[condition_col].astype(str).values
    reps = _rc_labels(screen.samples, replicate_col)
    rows = []
    for cnd in pd.unique(cond):
        cols = np.where(cond == cnd)[0]
        sub = rpm[:, cols]
        with np.errstate(all="ignore"):
            med = np.nanmedian(sub, axis=1, keepdims=True)
            mad = np.nanmedian(np.abs(sub - med), axis=1, keepdims=True) * 1.4826
            z = (sub - med) / mad
        gi, si = np.where((z > mad_z_thres) & (sub > abs_RPM_thres))
        for g, s in zip(gi, si):
            rows.append((screen.guides.index[g], screen.samples.index[cols[s]], reps.iloc[cols[s]], float(sub[g, s])))
    outliers = pd.DataFrame(rows, columns=["name", "sample", "replicate", "RPM"])
    mask = pd.DataFrame(1, index=screen.guides.index, columns=list(pd.unique(reps)))
    for _, row in outliers.iterrows():
        mask.loc[row["name"], row["replicate"]] = 0
    return outliers, mask


def qc_masks(screen, *, replicate_col: Union[str, Sequence[str]] = "replicate", condition_col: str = "condition",
             count_correlation_thres: float = 0.7, edit_rate_thres: float = 0.1, lfc_thres: float = -0.1,
             posctrl_col: Optional[str] = "target_group", posctrl_val: str = "PosCtrl",
             lfc_cond1: str = "top", lfc_cond2: str = "bot", control_condition: str = "bulk",
             base_edit_data: bool = True, remove_bad_replicates: bool = False,
             edit_start_pos: int = 2, edit_end_pos: int = 7):
    """Return a copy of ``screen`` with ``samples["mask"]``, ``uns["repguide_mask"]``,
    ``guides["edit_rate"]`` (when the editing layers exist) and the per-sample metric columns; guides that
    are outliers in more than two samples are dropped (notebook cells 38-47)."""
    scr = screen.copy()
    s = scr.samples
    if not isinstance(replicate_col, str):
        scr.uns["sample_covariates"] = list(replicate_col[1:])  # notebook cell 6
    s["replicate"] = _rc_labels(s, replicate_col) if isinstance(replicate_col, str) else s[replicate_col[0]].astype(str)
    if posctrl_col:
        if posctrl_col not in scr.guides.columns:
            raise ValueError(f"--posctrl-col argument '{posctrl_col}' is not present in the input "
                             f"ReporterScreen.guides.columns {scr.guides.columns}. If you do not want to use positive "
                             "control gRNA annotation for LFC calculation, feed --posctrl-col='' instead.")
        if posctrl_val not in scr.guides[posctrl_col].astype(str).tolist():
            raise ValueError(f"--posctrl-val argument '{posctrl_val}' is not present in the input "
                             f"ReporterScreen.guides[{posctrl_col}]. If you do not want to use positive control gRNA "
                             "annotation for LFC calculation, feed --posctrl-col='' instead.")
        sel = (scr.guides[posctrl_col].astype(str) == posctrl_val).values
    else:
        sel = np.ones(scr.n_obs, dtype=bool)
    lfc_col = f"median_lfc_corr.{lfc_cond1}_{lfc_cond2}"
    for col in ("gini_X", "median_corr_X", lfc_col, "mean_editing_rate", "mask"):
        if col in s.columns:
            del s[col]
    n_cols = len(s.columns)
    s["median_corr_X"] = sample_count_correlation(scr)
    s[lfc_col] = replicate_lfc_correlation(scr, sel, lfc_cond1, lfc_cond2, replicate_col, condition_col)
    has_edits = base_edit_data and "edits" in scr.layers and "X_bcmatch" in scr.layers
    if has_edits:
        scr.get_guide_edit_rate(condition_col=condition_col, unsorted_condition_label=control_condition)
        with np.errstate(all="ignore"):
            s["mean_editing_rate"] = np.nansum(scr.layers["edits"], axis=0) / np.nansum(scr.layers["X_bcmatch"], axis=0)
    # ---- cells 38-40: one pass / fail flag per metric, the mask is their conjunction
    ok = pd.DataFrame(1.0, index=s.index, columns=list(s.columns))
    ok.loc[s["median_corr_X"].isnull() | (s["median_corr_X"] < count_correlation_thres), "median_corr_X"] = 0.0
    if "mean_editing_rate" in s.columns:
        ok.loc[s["mean_editing_rate"] < edit_rate_thres, "mean_editing_rate"] = 0.0
    ok.loc[s[lfc_col] < lfc_thres, lfc_col] = 0.0
    if posctrl_col:
        ok.loc[s[lfc_col].isnull(), lfc_col] = 0.0
    s["mask"] = ok.iloc[:, n_cols:].astype(int).all(axis=1).astype(int).tolist()
    if remove_bad_replicates:
        reps = _rc_labels(s, replicate_col)
        n_good = s.groupby(reps.values)["mask"].sum()
        bad = n_good.loc[n_good < 2].index.tolist()
        keep = ~reps.isin(bad).values
        scr = scr[:, keep]
        if _rc_labels(scr.samples, replicate_col).nunique() <= 1:
            raise ValueError("Too small number of replicate left after QC. Check the input data or adjust the QC "
                             "metric thresholds.")
    # ---- cells 43-47: outlier guides
    outliers, mask = outlier_guides_and_mask(scr, condition_col, replicate_col)
    n_out = outliers["name"].value_counts()
    exclude = n_out.loc[n_out > 2].index
    scr.uns["repguide_mask"] = mask
    keep_g = ~scr.guides.index.isin(exclude)
    if not keep_g.all():
        scr = scr[keep_g, :]
        scr.uns["repguide_mask"] = mask.loc[scr.guides.index]
    return scr

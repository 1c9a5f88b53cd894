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

The ``edits`` layer is re-derived from the allele table inside the quantification window by default
(notebook cell 27: ``ReporterScreen.get_edit_from_allele`` / ``get_edit_mat_from_uns``, pinned by the
``edit_counts`` table and ``edits`` layer stored in the reference's own mini-screen files), editing rates
are recomputed from ``edits`` / ``X_bcmatch`` (``get_guide_edit_rate``, ``ReporterScreen.py:448-530``), and a
replicate that lacks a condition gets a dummy all-zero sample (cell 8, ``bean/qc/utils.py:90-157``).
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import numpy as np
import pandas as pd
from scipy.stats import spearmanr

import logging

logger = logging.getLogger(__name__)


def _rc_labels(samples: pd.DataFrame, replicate_col: Union[str, Sequence[str]]) -> pd.Series:
    if isinstance(replicate_col, str):
        return samples[replicate_col].astype(str)
    return samples[list(replicate_col)].astype(str).agg(".".join, axis=1)


def _add_dummy_sample(screen, rep, cond, condition_label: str, replicate_label: Union[str, Sequence[str]]):
    """One all-zero sample ``{rep}_{cond}`` appended to the screen (``bean/qc/utils.py:90-116``): its row of
    ``samples`` is the first sample of that condition with the replicate column(s) overwritten.  What
    the reference's ``concat`` does to the rest is reproduced: ``X`` and ``X_bcmatch`` get a zero column,
    every other layer is dropped (the dummy screen has none; ``edits`` is re-derived afterwards),
    allele / edit count tables get a zero column for the new sample."""
    from ..framework.ReporterScreen import ReporterScreen

    sample_id = f"{rep}_{cond}"
    s = screen.samples
    rows = s.loc[s[condition_label] == cond, :]
    row = rows.iloc[[0], :].copy()
    row.index = pd.Index([sample_id], name=s.index.name)
    if isinstance(replicate_label, str):
        row[replicate_label] = rep
    else:
        for col, val in zip(replicate_label, rep):
            row[col] = val
    samples = pd.concat([s, row])
    zero = np.zeros((screen.n_obs, 1), dtype=screen.X.dtype)
    layers = {}
    if "X_bcmatch" in screen.layers:
        layers["X_bcmatch"] = np.concatenate([screen.layers["X_bcmatch"], zero.astype(screen.layers["X_bcmatch"].dtype)], 1)
    uns = {}
    for k, v in screen.uns.items():
        if isinstance(v, pd.DataFrame) and ("edit" in k or "allele" in k):
            v = v.copy()
            v[sample_id] = 0
        uns[k] = v
    return ReporterScreen(np.concatenate([screen.X, zero], 1), screen.guides, samples, layers, uns)


def fill_in_missing_samples(screen, condition_label: str, replicate_label: Union[str, Sequence[str]]):
    """If a replicate lacks a condition, add a dummy sample for it (``bean/qc/utils.py:118-157``; notebook
    cell 8): ``bean run`` needs every replicate to carry every condition, the dummy is masked by its zero
    counts.  With several replicate columns the samples are then sorted by them and the condition, as
    the reference does in that case only."""
    added = False
    if isinstance(replicate_label, str):
        rep_list = list(pd.unique(screen.samples[replicate_label]))
    else:
        # lists, as the reference makes them before it formats the dummy's id (bean/qc/utils.py:139, 96):
        # "['rep1', 'x']_bulk"
        rep_list = [list(r) for r in screen.samples[list(replicate_label)].drop_duplicates().values.tolist()]
    for rep in rep_list:
        for cond in pd.unique(screen.samples[condition_label]):
            if isinstance(replicate_label, str):
                in_rep = screen.samples[replicate_label] == rep
            else:
                in_rep = (screen.samples[list(replicate_label)] == list(rep)).all(axis=1)
            if int((in_rep & (screen.samples[condition_label] == cond)).sum()) != 1:
                print(f"Adding dummy samples for {rep}, {cond}")
                screen = _add_dummy_sample(screen, rep, cond, condition_label, replicate_label)
                added = True
    if added and not isinstance(replicate_label, str):
        order = screen.samples.sort_values(list(replicate_label) + [condition_label]).index
        screen = screen[:, order]
    return screen


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
        else:
            logger.warning("replicate %s has %d sample(s) of condition `%s` and %d of `%s`: no LFC correlation for it "
                           "(its samples are masked when positive controls are in use)", rep, len(a), cond1, len(b), cond2)
    out = pd.Series(np.nan, index=screen.samples.index)
    names = list(lfc)
    if len(names) < 2 or int(np.sum(guide_sel)) < 3:
        logger.warning("LFC correlation needs at least 2 replicates with both conditions and 3 selected guides "
                       "(have %d replicates, %d guides): every sample gets NaN", len(names), int(np.sum(guide_sel)))
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
             edit_start_pos: int = 2, edit_end_pos: int = 7, recalculate_edits: bool = True,
             target_pos_col: str = "target_pos", rel_pos_is_reporter: bool = False,
             reporter_length: Optional[int] = None, reporter_right_flank_length: Optional[int] = None,
             ignore_missing_samples: bool = False):
    """Return a copy of ``screen`` with ``samples["mask"]``, ``uns["repguide_mask"]``, the ``edits`` layer
    re-derived from the allele table inside ``[edit_start_pos, edit_end_pos)`` (unless
    ``recalculate_edits`` is off), ``guides["edit_rate"]`` and the per-sample metric columns; replicates
    that lack a condition get a dummy sample (unless ``ignore_missing_samples``); guides that are outliers
    in more than two samples are dropped (notebook cells 5-8, 27-29, 38-47)."""
    scr = screen.copy()
    # ---- cell 5
    if "target_base_change" in scr.uns and "target_base_changes" not in scr.uns:
        scr.uns["target_base_changes"] = scr.uns["target_base_change"]
    if reporter_length is None:  # bean/qc/utils.py:77-87
        reporter_length = int(scr.uns["reporter_length"]) if "reporter_length" in scr.uns else 32
    if reporter_right_flank_length is None:
        reporter_right_flank_length = (int(scr.uns["reporter_right_flank_length"])
                                       if "reporter_right_flank_length" in scr.uns else 6)
    scr.uns["reporter_length"] = reporter_length
    scr.uns["reporter_right_flank_length"] = reporter_right_flank_length
    # ---- cell 6
    s = scr.samples
    if not isinstance(replicate_col, str):
        scr.uns["sample_covariates"] = list(replicate_col[1:])
        for col in replicate_col:
            s[col] = s[col].astype(str)
        s["replicate"] = s[replicate_col[0]]
    else:
        s["replicate"] = s[replicate_col] = s[replicate_col].astype(str)
    s["condition"] = s[condition_col]
    # ---- cell 8
    if not ignore_missing_samples:
        scr = fill_in_missing_samples(scr, condition_col, replicate_col)
        s = scr.samples
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
    # ---- cell 27: the edits layer, from the allele table inside the quantification window
    if "target_base_changes" not in scr.uns or not scr.uns["target_base_changes"] or not base_edit_data:
        base_edit_data = False
        print("Not a base editing data or target base change not provided. Passing editing-related QC")
    elif recalculate_edits or "edits" not in scr.layers or float(np.max(scr.layers["edits"])) == 0:
        if "allele_counts" in scr.uns:
            ac = scr.uns["allele_counts"]
            scr.uns["allele_counts"] = ac.loc[ac["allele"].map(str) != ""]
            scr.get_edit_from_allele()
            scr.get_edit_mat_from_uns(rel_pos_start=edit_start_pos, rel_pos_end=edit_end_pos,
                                      target_pos_col=target_pos_col, rel_pos_is_reporter=rel_pos_is_reporter)
    # ---- cell 29
    has_edits = base_edit_data and "edits" in scr.layers and "X_bcmatch" in scr.layers
    if has_edits:
        scr.get_guide_edit_rate(editable_base_start=edit_start_pos, editable_base_end=edit_end_pos,
                                condition_col=condition_col, unsorted_condition_label=control_condition)
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
    for name, row in ok.iloc[:, n_cols:].iterrows():
        failed = [c for c, v in row.items() if v == 0]
        if failed:
            logger.warning("sample %s masked: fails %s", name, ", ".join(
                f"{c} ({s.loc[name, c]:.3g})" if pd.notnull(s.loc[name, c]) else f"{c} (not available)" for c in failed))
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

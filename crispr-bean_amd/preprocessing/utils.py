"""Screen formatting helpers of ``bean run`` (``bean/preprocessing/utils.py:24-67,
233-251``, ``bean/qc/guide_qc.py:49-74``)."""
from __future__ import annotations

import numpy as np
import pandas as pd


def filter_no_info_target(bdata, condit_col: str, control_condition: str, target_col: str = "target",
                          write_no_support_targets: bool = False, no_support_target_write_path: str = None):
    """Drop the guides of targets whose guides have no count in any sample."""
    per_target = pd.DataFrame(bdata.X, index=bdata.guides.index).groupby(bdata.guides[target_col].astype(str).values).sum()
    empty = per_target.index[per_target.sum(axis=1) == 0]
    if write_no_support_targets:
        pd.Series(empty, name=target_col).to_csv(no_support_target_write_path, index=False)
    keep = ~bdata.guides[target_col].isin(empty).values
    return len(empty), bdata[keep, :].copy()


def prepare_bdata(bdata, args, warn, prefix: str, write_files: bool = True):
    """Format a screen for the fit: category replicate column, drop zero-count guides,
    sort guides by target, drop targets without counts."""
    bdata = bdata.copy()
    bdata.samples["replicate"] = bdata.samples[args.replicate_col].astype("category")
    bdata.guides = bdata.guides.loc[:, ~bdata.guides.columns.duplicated()].copy()
    if args.selection == "sorting" or args.exclude_control_condition_for_inference:
        test = bdata[:, (bdata.samples[args.condition_col].astype(str) != args.control_condition).values]
    else:
        test = bdata
    zero = test.X.sum(axis=1) == 0
    if zero.any():
        warn(f"Filtering out {int(zero.sum())} gRNAs without any counts over all samples.")
        bdata = bdata[~zero, :]
    # The reference's zero-count-sample check reads `bdata.samples.mask`, which is the
    # DataFrame.mask *method* (SURVEY.md Appendix C item 11), so it never fires; report
    # the condition it meant to catch without halting.
    if "mask" in bdata.samples.columns:
        kept = (bdata.samples["mask"] == 1).values
        if kept.any() and (bdata.X[:, kept].sum(axis=0) == 0).any():
            warn("Some unmasked samples have 0 counts. Make sure you mask those samples.")
    if args.library_design == "variant":
        if bdata.guides[args.target_col].isnull().any():
            raise ValueError(
                f"Some target column (bdata.guides[{args.target_col}]) value is null. Check your input file.")
        bdata = bdata[np.argsort(bdata.guides[args.target_col].astype(str).values, kind="stable"), :]
        n_bad, bdata = filter_no_info_target(
            bdata, condit_col=args.condition_col, control_condition=args.control_condition,
            target_col=args.target_col, write_no_support_targets=write_files,
            no_support_target_write_path=f"{prefix}/no_support_targets.csv")
        if n_bad > 0:
            warn(f"Ignoring {n_bad} targets with 0 gRNA counts across all non-control samples. "
                 f"Ignored targets are written in {prefix}/no_support_targets.csv.")
    return bdata


def assign_rep_ids_and_sort(screen, rep_col: str, condition_id_col: str = None):
    """Number the replicates in sorted order and sort the samples by (replicate id, condition id)."""
    if rep_col not in screen.samples.columns:
        raise ValueError(f"{rep_col} not in columns of ReporterScreen.samples with following columns: "
                         f"{screen.samples.columns}.")
    reps = sorted(screen.samples[rep_col].astype(str).unique())
    screen.samples[f"{rep_col}_id"] = screen.samples[rep_col].astype(str).map({r: i for i, r in enumerate(reps)})
    keys = [f"{rep_col}_id"] + ([condition_id_col] if condition_id_col else [])
    order = screen.samples.sort_values(keys, kind="stable").index
    return screen[:, order]


def _accessibility_single(pos, track, chrom: str = "chr19", guide_start_pos: int = 0, half_window_size: int = 100):
    """Mean log accessibility signal around a genomic position (``bean/preprocessing/utils.py:70-108``):
    exp(nanmean(log(values + 1))) over the window; NaN when the position is missing or the read fails."""
    if half_window_size < 0:
        raise ValueError("Window size must be non-negative.")
    if pos == "control" or (isinstance(pos, float) and np.isnan(pos)):
        return np.nan
    try:
        v = track.values(chrom, int(guide_start_pos + pos - half_window_size),
                         int(guide_start_pos + pos + half_window_size))
        with np.errstate(all="ignore"):
            return float(np.exp(np.nanmean(np.log(np.asarray(v) + 1.0))))
    except Exception as exc:  # the reference prints and carries on (utils.py:106-108)
        print(exc)
        return np.nan


def get_accessibility_guides(accessibility_bw_path: str, guide_info, half_window_size: int = 100):
    """Guide accessibility from a bigWig track (``bean/preprocessing/utils.py:111-147``): needs
    ``genomic_pos`` and ``chrom`` (or ``chr``) columns; guides without a value get the median."""
    import torch

    from ..framework.bigwig import open_bigwig

    acc = open_bigwig(accessibility_bw_path)
    if "chr" in guide_info.columns and "chrom" not in guide_info.columns:
        guide_info = guide_info.rename(columns={"chr": "chrom"})
    has_chrom = "chrom" in guide_info.columns
    vals = [
        _accessibility_single(row.genomic_pos, acc, chrom=(row.chrom if has_chrom else "chr19"),
                              guide_start_pos=0, half_window_size=half_window_size)
        for row in guide_info.itertuples()
    ]
    acc.close()
    out = torch.as_tensor(np.asarray(vals, dtype=np.float64))
    if torch.isnan(out).all():
        raise ValueError("Cannot retrieve guide accessibility from the bigWig file. Check your inputs.")
    out[torch.isnan(out)] = torch.nanmedian(out)
    return out

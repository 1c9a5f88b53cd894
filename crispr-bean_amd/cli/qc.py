"""``bean qc``: mask low-quality samples and outlier guides (``bean/cli/qc.py:9-58``).

The reference executes a notebook through papermill and converts it to an HTML report; here the same
masking logic runs as a function (``bean_amd.qc.qc_masks``) and the masked screen is written to
``--out-screen-path``; the per-sample metrics go to ``<out-report-prefix>.samples.csv`` instead of a
rendered report.  Every flag of the reference's parser is honoured (missing-sample dummies, the ``edits``
re-derivation window and its switches, reporter geometry)."""
from __future__ import annotations

import argparse


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def attach_args(parser):
    """The flags of ``bean/qc/parser.py:15-172`` (same names, defaults and groups)."""
    parser.add_argument("bdata_path", help="Path to the ReporterScreen object to run QC on", type=str)
    thres = parser.add_argument_group("QC thresholds")
    run = parser.add_argument_group("Run options")
    inp = parser.add_argument_group("Input .h5ad formatting")
    thres.add_argument("--count-correlation-thres", help="Correlation threshold to mask out.", type=float, default=0.7)
    thres.add_argument("--edit-rate-thres", help="Mean editing rate threshold per sample to mask out.", type=float,
                       default=0.1)
    thres.add_argument("--lfc-thres", help="Positive guides' correlation threshold to filter out.", type=float,
                       default=-0.1)
    parser.add_argument("-o", "--out-screen-path", type=str,
                        help="Path where quality-filtered ReporterScreen object to be written to")
    parser.add_argument("-r", "--out-report-prefix", type=str, help="Output prefix of qc report")
    run.add_argument("-b", "--remove-bad-replicates", action="store_true",
                     help="Remove replicates with at least two of its samples meet the QC threshold.")
    run.add_argument("-i", "--ignore-missing-samples", action="store_true",
                     help="Do not add dummy samples for replicates that lack a condition.")
    run.add_argument("--no-editing", action="store_true", help="Ignore QC about editing.")
    run.add_argument("--dont-recalculate-edits", action="store_true",
                     help="Do not recalculate the edit counts from ReporterScreen.uns['allele_count'].")
    inp.add_argument("--tiling", dest="tiling", type=str2bool,
                     help="Specify that the guide library is tiling library without 'n guides per target' design")
    inp.add_argument("--replicate-col", type=str, default="replicate",
                     help="Label of column in `bdata.samples` that describes replicate ID.")
    inp.add_argument("--sample-covariates", type=str, default=None,
                     help="Comma-separated list of column names in `bdata.samples` that describes non-selective "
                          "experimental condition. (drug treatment, etc.)")
    inp.add_argument("--condition-col", type=str, default="condition",
                     help="Label of column in `bdata.samples` that describes experimental condition.")
    inp.add_argument("--target-pos-col", type=str, default="target_pos",
                     help="Target position column in `bdata.guides` specifying target edit position in reporter")
    inp.add_argument("--rel-pos-is-reporter", action="store_true", default=False)
    inp.add_argument("--edit-start-pos", default=2, type=int)
    inp.add_argument("--edit-end-pos", default=7, type=int)
    inp.add_argument("--posctrl-col", type=str, default="target_group",
                     help="Column in ReporterScreen.guides that specifies guide category; '' uses all gRNAs.")
    inp.add_argument("--posctrl-val", type=str, default="PosCtrl")
    inp.add_argument("--lfc-conds", type=str, default="top,bot",
                     help="Two condition labels the LFC is calculated between, delimited by comma")
    inp.add_argument("--control-condition", type=str, default="bulk",
                     help="Condition for which the guide-level editing rate is calculated")
    parser.add_argument("--reporter-length", type=int, default=None)
    parser.add_argument("--reporter-right-flank-length", type=int, default=None)
    return parser


def main(args):
    from ..framework import read_h5ad
    from ..framework.h5ad_io import write_screen
    from ..qc import qc_masks

    print("  \n~~~BEANQC~~~")
    print("-Check guide/sample level quality and mask / discard-")
    if args.out_screen_path is None:
        args.out_screen_path = f"{args.bdata_path.rsplit('.h5ad', 1)[0]}.masked.h5ad"
    if args.out_report_prefix is None:
        args.out_report_prefix = f"{args.bdata_path.rsplit('.h5ad', 1)[0]}.qc_report"
    bdata = read_h5ad(args.bdata_path)
    # bean/qc/utils.py:9-88 (check_args)
    for flag, col in (("--replicate-col", args.replicate_col), ("--condition-col", args.condition_col)):
        if col not in bdata.samples.columns:
            raise ValueError(f"Specified {flag} `{col}` does not exist in ReporterScreen.samples.columns "
                             f"({bdata.samples.columns}). Please check your input.")
    conds = bdata.samples[args.condition_col].astype(str).tolist()
    if args.control_condition not in conds:
        raise ValueError(f"Specified --control-condition `{args.control_condition}` does not exist in "
                         f"ReporterScreen.samples[{args.condition_col}]. Please check your input.")
    lfc = args.lfc_conds.split(",")
    if len(lfc) != 2:
        raise ValueError(f"lfc_conds must be two condition labels delimited by comma. Provided {args.lfc_conds}")
    for c in lfc:
        if c not in conds:
            raise ValueError(f"Specified --lfc-conds `{c}` does not exist in ReporterScreen.samples"
                             f"[{args.condition_col}]. Please check your input.")
    rep = args.replicate_col
    if args.sample_covariates is not None:
        rep = [args.replicate_col] + args.sample_covariates.split(",")
    if args.tiling is not None:
        bdata.uns["tiling"] = args.tiling
    elif "tiling" not in bdata.uns:
        raise ValueError("Ambiguous assignment if the screen is a tiling screen. Provide `--tiling=True` or "
                         "`tiling=False`.")
    # bean/qc/utils.py:19-32: a variant screen of a base editor needs the target position column
    base_change = bdata.uns.get("target_base_changes") or bdata.uns.get("target_base_change")
    if not bdata.uns["tiling"] and base_change and args.target_pos_col not in bdata.guides.columns:
        raise ValueError(f"Specified --target-pos-col `{args.target_pos_col}` does not exist in "
                         f"ReporterScreen.guides.columns ({bdata.guides.columns}). Please check your input. "
                         f"(--tiling {args.tiling}, ReporterScreen.tiling: {bdata.uns['tiling']})")
    out = qc_masks(
        bdata, replicate_col=rep, condition_col=args.condition_col,
        count_correlation_thres=args.count_correlation_thres, edit_rate_thres=args.edit_rate_thres,
        lfc_thres=args.lfc_thres, posctrl_col=args.posctrl_col, posctrl_val=args.posctrl_val,
        lfc_cond1=lfc[0], lfc_cond2=lfc[1], control_condition=args.control_condition,
        base_edit_data=not args.no_editing, remove_bad_replicates=args.remove_bad_replicates,
        edit_start_pos=args.edit_start_pos, edit_end_pos=args.edit_end_pos,
        recalculate_edits=not args.dont_recalculate_edits, target_pos_col=args.target_pos_col,
        rel_pos_is_reporter=args.rel_pos_is_reporter, reporter_length=args.reporter_length,
        reporter_right_flank_length=args.reporter_right_flank_length,
        ignore_missing_samples=args.ignore_missing_samples,
    )
    write_screen(out, args.out_screen_path)
    out.samples.to_csv(f"{args.out_report_prefix}.samples.csv")
    n_masked = int((out.samples["mask"] == 0).sum())
    print(f"{n_masked} of {len(out.samples)} samples masked; "
          f"{int((out.uns['repguide_mask'] == 0).values.sum())} (guide, replicate) pairs masked; "
          f"{bdata.n_obs - out.n_obs} outlier guides removed. Wrote {args.out_screen_path}")
    return args.out_screen_path

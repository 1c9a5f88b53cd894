"""The ``bean run`` command-line contract.

Same positional arguments, flags, short aliases, destinations, defaults and types
as ``bean/model/parser.py:10-262`` (pinned by ``tests/golden/run_flags.json``,
dumped from the reference's own parser), declared here as a table.  Help strings
are this project's.
"""
from __future__ import annotations

import argparse


def none_or_str(value):
    return None if value == "None" else value


_S, _I = str, int
# (group, flags, kwargs)
_FLAGS = [
    ("General run options", ("--uniform-edit", "-p"), dict(action="store_true", default=False,
     help="assume one editing rate for all guides (Normal model, no reporter needed)")),
    ("General run options", ("--scale-by-acc",), dict(action="store_true", default=False,
     help="scale the reporter editing rate by target-site accessibility")),
    ("General run options", ("--acc-bw-path",), dict(type=_S, default=None, help="accessibility bigWig")),
    ("General run options", ("--acc-col",), dict(type=_S, default=None,
     help="column of bdata.guides with the raw accessibility signal")),
    ("General run options", ("--outdir", "-o"), dict(default=".", type=_S, help="output directory")),
    ("General run options", ("--result-suffix",), dict(default="", type=_S, help="suffix of the output files")),
    ("General run options", ("--cuda",), dict(action="store_true", default=False,
     help="accepted for compatibility: this implementation always runs on the GPU")),
    ("General run options", ("--fit-negctrl",), dict(action="store_true", default=False,
     help="fit the shared negative-control distribution and scale the results by it")),
    ("General run options", ("--guide-lfc-pseudocount",), dict(type=_I, default=5,
     help="pseudocount of the per-guide log fold changes in bean_sgRNA_result")),
    ("General run options", ("--dont-fit-noise",), dict(action="store_true")),
    ("General run options", ("--dont-adjust-confidence-by-negative-control",), dict(action="store_true",
     help="do not rescale the z-scores by the negative-control spread")),
    ("General run options", ("--load-existing",), dict(action="store_true", help="load an existing .pkl result")),
    ("General run options", ("--save-raw",), dict(action="store_true", help="write a .pkl with raw input/output")),
    ("General run options", ("--device",), dict(type=_S, default=None, help="GPU device name (e.g. cuda:0)")),
    ("Input .h5ad formatting", ("--condition-col",), dict(default="condition", type=_S,
     help="column of bdata.samples with the experimental condition")),
    ("Input .h5ad formatting", ("--time-col",), dict(default="time", type=_S, help="column with elapsed time")),
    ("Input .h5ad formatting", ("--control-condition",), dict(default="bulk", type=_S,
     help="comma-separated condition values marking the control samples")),
    ("Input .h5ad formatting", ("--plasmid-condition",), dict(default="bulk", type=_S,
     help="condition label of the plasmid library (survival screens)")),
    ("Input .h5ad formatting", ("--replicate-col",), dict(default="replicate", type=_S, help="replicate column")),
    ("Input .h5ad formatting", ("--target-col",), dict(default="target", type=_S,
     help="column of bdata.guides with the target element of each guide")),
    ("Input .h5ad formatting", ("--guide-activity-col", "-a"), dict(type=_S, default=None,
     help="column of bdata.guides with externally estimated editing rates")),
    ("Input .h5ad formatting", ("--sorting-bin-upper-quantile-col", "-uq"), dict(default="upper_quantile",
     help="samples column with the upper quantile of each sorting bin")),
    ("Input .h5ad formatting", ("--sorting-bin-lower-quantile-col", "-lq"), dict(default="lower_quantile",
     help="samples column with the lower quantile of each sorting bin")),
    ("Input .h5ad formatting", ("--sample-mask-col",), dict(type=_S, default="mask",
     help="samples column with the 0/1 sample mask")),
    ("Input .h5ad formatting", ("--negctrl-col",), dict(type=_S, default="target_group",
     help="guides column flagging negative controls")),
    ("Input .h5ad formatting", ("--negctrl-col-value",), dict(type=_S, default="negctrl",
     help="value of --negctrl-col that marks a negative control (case-insensitive)")),
    ("Input .h5ad formatting", ("--repguide-mask",), dict(type=none_or_str, default="repguide_mask",
     help="key of screen.uns with the n_replicate x n_guide outlier mask")),
    ("Input .h5ad formatting", ("--allele-df-key",), dict(type=_S, default=None,
     help="key of screen.uns with the allele counts (tiling)")),
    ("Input .h5ad formatting", ("--splice-site-path",), dict(type=_S, default=None, help="splice-site table")),
    ("Input .h5ad formatting", ("--control-guide-tag",), dict(type=none_or_str, default=None,
     help="guides whose name contains this tag keep guide-specific positions")),
    ("Advanced arguments for model fitting", ("--n-iter",), dict(type=_I, default=2000, help="number of SVI steps")),
    ("Advanced arguments for model fitting", ("--ignore-bcmatch",), dict(action="store_true", default=False,
     help="ignore barcode-matched counts even if present")),
    ("Advanced arguments for model fitting", ("--prior-params",), dict(type=_S, default=None,
     help=".pkl with prior parameters (mu_loc, mu_scale, sd_loc, sd_scale)")),
    ("Advanced arguments for model fitting", ("--rep-pi", "-r"), dict(action="store_true", default=False,
     help="parsed for compatibility (unused by the reference)")),
    ("Advanced arguments for model fitting", ("--const-pi",), dict(default=False, action="store_true",
     help="parsed for compatibility (dead path in the reference)")),
    ("Advanced arguments for model fitting", ("--shrink-alpha",), dict(default=False, action="store_true",
     help="shrink per-guide dispersion estimates towards the fitted trend")),
    ("Advanced arguments for model fitting", ("--exclude-control-condition-for-inference", "-ec"),
     dict(default=False, action="store_true", help="exclude the control condition from inference")),
    ("Advanced arguments for model fitting", ("--alpha-if-overdispersion-fitting-fails", "-af"),
     dict(default=None, type=_S, help="fallback (b0,b1) of log(a0) ~ log(q)")),
]


def parse_args(parser=None):
    """Attach the ``bean run`` arguments to ``parser`` (a new one if None) and
    return it - the reference's function of the same name does not parse either."""
    if parser is None:
        parser = argparse.ArgumentParser(description="Run model on data.")
    parser.add_argument("selection", type=str, choices=["sorting", "survival"],
                        help="'sorting' (cells sorted on a continuous phenotype) or 'survival' (proliferation)")
    parser.add_argument("library_design", type=str, choices=["variant", "tiling"],
                        help="'variant' (one target variant per gRNA) or 'tiling' (all reporter alleles)")
    parser.add_argument("bdata_path", type=str, help="path of a ReporterScreen (.h5ad)")
    groups = {}
    for group, flags, kw in _FLAGS:
        if group not in groups:
            groups[group] = parser.add_argument_group(group)
        groups[group].add_argument(*flags, **kw)
    return parser

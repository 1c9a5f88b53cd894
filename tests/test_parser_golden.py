"""The `bean run` flag contract equals the reference's (golden dump of its parser)."""
import importlib.util
import json
import os

import bean_amd  # noqa: F401
from bean_amd.model import parser as run_parser

HERE = os.path.join(os.path.dirname(__file__), "golden")
spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "make_parser_golden.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)


def test_flag_table_matches_reference():
    want = json.load(open(os.path.join(HERE, "run_flags.json")))
    got = mk.dump(run_parser.parse_args())
    assert [g["dest"] for g in got] == [w["dest"] for w in want]  # same order too
    for g, w in zip(got, want):
        assert g == w, (g, w)


def test_parsing_examples():
    p = run_parser.parse_args()
    a = p.parse_args(["sorting", "variant", "x.h5ad", "--scale-by-acc", "--acc-bw-path", "a.bw", "-o", "out",
                      "--fit-negctrl", "--n-iter", "10", "--repguide-mask", "None", "-uq", "hi"])
    assert (a.selection, a.library_design, a.bdata_path) == ("sorting", "variant", "x.h5ad")
    assert a.scale_by_acc and a.fit_negctrl and a.n_iter == 10 and a.outdir == "out"
    assert a.repguide_mask is None and a.sorting_bin_upper_quantile_col == "hi"
    assert a.sample_mask_col == "mask" and a.control_condition == "bulk" and a.guide_lfc_pseudocount == 5

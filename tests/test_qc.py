"""`bean qc` masks: thresholds and masking logic of the reference's notebook
(bean/notebooks/sample_quality_report.ipynb cells 38-47, bean/qc/guide_qc.py:5-46) on constructed
screens and on the reference's mini-screen file; the metric definitions themselves restate
perturb-tools and are unpinned (see bean_amd/qc/sample_qc.py)."""
import os

import numpy as np
import pandas as pd
import pytest

import bean_amd  # noqa: F401
from bean_amd.framework import h5ad_io, read_h5ad
from bean_amd.framework.ReporterScreen import ReporterScreen
from bean_amd.qc import qc_masks
from bean_amd.qc.sample_qc import outlier_guides_and_mask

GOLD = os.path.join(os.path.dirname(__file__), "golden")
VAR = os.path.join(GOLD, "var_mini_screen.h5ad")


def _screen(n_guides=3000, seed=0, bad_sample=None, outlier=None):
    rng = np.random.default_rng(seed)
    rows = [dict(name=f"{rep}_{c}", replicate=rep, condition=c) for rep in ("r1", "r2", "r3") for c in ("top", "bot", "bulk")]
    samples = pd.DataFrame(rows).set_index("name")
    base = rng.gamma(2.0, 150.0, n_guides)                       # guide abundance shared by all samples
    eff = np.zeros(n_guides)
    eff[:60] = rng.normal(1.0, 0.3, 60)                          # positive controls: enriched in "top"
    X = np.empty((n_guides, len(samples)))
    for j, (_, row) in enumerate(samples.iterrows()):
        shift = {"top": eff, "bot": -eff, "bulk": 0 * eff}[row["condition"]]
        X[:, j] = rng.poisson(base * np.exp(shift))
    if bad_sample is not None:                                   # a sample unrelated to the library
        X[:, bad_sample] = rng.poisson(rng.gamma(2.0, 150.0, n_guides))
    if outlier is not None:                                      # one jackpot guide in one sample
        g, j = outlier
        X[g, j] = X[:, j].sum() * 0.05
    guides = pd.DataFrame({"target_group": ["PosCtrl"] * 60 + ["Variant"] * (n_guides - 60),
                           "target": [f"t{i // 4}" for i in range(n_guides)]}, index=[f"g{i}" for i in range(n_guides)])
    edits = np.floor(X * 0.8 * 0.4)
    return ReporterScreen(X.astype(np.float32), guides, samples, layers={"X_bcmatch": np.floor(X * 0.8), "edits": edits},
                          uns={"tiling": False})


def test_good_screen_keeps_everything():
    out = qc_masks(_screen())
    assert out.samples["mask"].tolist() == [1] * 9
    assert (out.samples["median_corr_X"] > 0.7).all() and (out.samples["median_lfc_corr.top_bot"] > 0.5).all()
    assert out.uns["repguide_mask"].shape == (3000, 3) and (out.uns["repguide_mask"].values == 1).all()
    assert abs(np.nanmedian(out.guides["edit_rate"]) - 0.4) < 0.01
    np.testing.assert_allclose(out.samples["mean_editing_rate"], 0.4, atol=0.01)


def test_uncorrelated_sample_is_masked_and_low_editing_rate_too():
    out = qc_masks(_screen(bad_sample=4))
    mask = out.samples["mask"].tolist()
    assert mask[4] == 0 and sum(mask) == 8
    assert out.samples["median_corr_X"].iloc[4] < 0.3
    scr = _screen()
    scr.layers["edits"][:, 7] = 0.0  # mean editing rate 0 < edit_rate_thres
    assert qc_masks(scr).samples["mask"].tolist()[7] == 0
    # without positive-control annotation every guide enters the LFC correlation and NaN no longer fails
    out = qc_masks(_screen(), posctrl_col="")
    assert out.samples["mask"].sum() == 9


def test_outlier_guide_masks_its_replicate():
    scr = _screen(outlier=(123, 5))  # sample 5 = r2_bulk
    outliers, mask = outlier_guides_and_mask(scr, "condition", "replicate")
    assert outliers["name"].tolist() == ["g123"] and outliers["sample"].tolist() == ["r2_bulk"]
    assert mask.loc["g123"].tolist() == [1, 0, 1] and int((mask.values == 0).sum()) == 1
    out = qc_masks(scr)
    assert out.uns["repguide_mask"].loc["g123", "r2"] == 0 and out.n_obs == 3000  # one sample: the guide stays


@pytest.mark.skipif(not os.path.exists(h5ad_io.HELPER_PYTHON), reason="no h5py interpreter")
def test_qc_cli_writes_a_screen_bean_run_accepts(tmp_path):
    from bean_amd.cli.execute import main as bean_main
    from bean_amd.model import parser as run_parser
    from bean_amd.model.run import check_args

    out = str(tmp_path / "var_masked.h5ad")
    assert bean_main(["qc", VAR, "-o", out, "-r", str(tmp_path / "report"), "--count-correlation-thres", "0.5"]) == 0
    a, b = read_h5ad(VAR), read_h5ad(out)
    assert np.array_equal(a.X, b.X) and set(b.layers) == set(a.layers)
    assert {"mask", "median_corr_X", "median_lfc_corr.top_bot"} <= set(b.samples.columns)
    assert b.uns["repguide_mask"].shape == (30, 2) and "edit_rate" in b.guides.columns
    assert b.uns["allele_counts"].shape == a.uns["allele_counts"].shape and b.uns["tiling"] is False
    assert os.path.exists(tmp_path / "report.samples.csv")
    # the defaults of `bean run` (--sample-mask-col mask, --repguide-mask repguide_mask) now find their inputs
    args = run_parser.parse_args().parse_args(["sorting", "variant", out])
    args, b2 = check_args(args, b)
    assert args.sample_mask_col == "mask" and args.repguide_mask == "repguide_mask"

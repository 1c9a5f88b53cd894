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
                          uns={"tiling": False, "target_base_changes": "A>G"})


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


# ---------------------------------------------------------------- reference fixtures with missing samples
VAR_MISSING = os.path.join(GOLD, "var_mini_screen_missing.h5ad")
TILING_MISSING = os.path.join(GOLD, "tiling_mini_screen_missing.h5ad")
TILING = os.path.join(GOLD, "tiling_mini_screen.h5ad")


@pytest.fixture()
def _needs_h5ad():
    try:
        import h5py  # noqa: F401
    except ImportError:
        assert os.path.exists(h5ad_io.HELPER_PYTHON), "no h5py and no helper interpreter: cannot read .h5ad"


@pytest.mark.parametrize("path", [VAR, TILING])
def test_edit_tables_reproduce_what_the_reference_stored(_needs_h5ad, path):
    """The mini-screen files carry ``uns["edit_counts"]`` and ``layers["edits"]`` as the reference's own
    pipeline derived them from ``uns["allele_counts"]``: ``get_edit_from_allele`` / ``get_edit_mat_from_uns``
    (notebook cell 27; window [2, 7), the `bean qc` default) must reproduce both exactly."""
    b = read_h5ad(path)
    if "target_base_change" in b.uns and "target_base_changes" not in b.uns:
        b.uns["target_base_changes"] = b.uns["target_base_change"]
    cols = b.samples.index.tolist()
    stored = b.uns["edit_counts"]
    want = stored.assign(edit=stored["edit"].map(str)).groupby(["guide", "edit"])[cols].sum()
    got = b.get_edit_from_allele(return_result=True).groupby(["guide", "edit"])[cols].sum()
    assert want.index.sort_values().equals(got.index.sort_values())
    assert float((want - got.loc[want.index]).abs().values.max()) == 0.0
    old = b.layers["edits"].copy()
    b.get_edit_from_allele()
    tcol = "target_pos" if "target_pos" in b.guides.columns else "target_start"
    returned = b.get_edit_mat_from_uns(rel_pos_start=2, rel_pos_end=7, target_pos_col=tcol)
    assert np.array_equal(returned, old) and np.array_equal(b.layers["edits"], old)
    if b.tiling:  # the window matters in tiling screens: a wider one counts more edits
        b.get_edit_mat_from_uns(rel_pos_start=0, rel_pos_end=32)
        assert b.layers["edits"].sum() > old.sum()


def test_fill_in_missing_samples_adds_masked_dummies(_needs_h5ad):
    from bean_amd.qc import fill_in_missing_samples

    b = read_h5ad(VAR_MISSING)
    n0 = b.n_vars
    pairs = b.samples[["replicate", "condition"]].astype(str).value_counts()
    n_missing = b.samples["replicate"].nunique() * b.samples["condition"].nunique() - len(pairs)
    assert n_missing > 0
    out = fill_in_missing_samples(b, "condition", "replicate")
    assert out.n_vars == n0 + n_missing
    new = [s for s in out.samples.index if s not in b.samples.index]
    assert all(float(out.X[:, out.samples.index.get_loc(s)].sum()) == 0.0 for s in new)
    counts = out.samples[["replicate", "condition"]].astype(str).value_counts()
    assert (counts == 1).all() and len(counts) == out.samples["replicate"].nunique() * out.samples["condition"].nunique()
    for k in ("allele_counts", "edit_counts"):
        assert all(s in out.uns[k].columns for s in new) and float(out.uns[k][new].sum().sum()) == 0.0
    assert set(out.layers) == {"X_bcmatch"}  # what the reference's concat keeps; edits are re-derived by qc
    # idempotent
    assert fill_in_missing_samples(out, "condition", "replicate").n_vars == out.n_vars


def test_cli_qc_on_screens_with_missing_samples(_needs_h5ad, tmp_path):
    """The reference's tests/test_qc.py:42-63 (`bean qc *_missing.h5ad --count-correlation-thres 0.6 -b`):
    clean exit, and here also: the written screen holds every (replicate, condition), the dummies are
    masked, and `bean run`'s data class accepts it."""
    from bean_amd.cli.execute import main as bean_main
    from bean_amd.preprocessing.screen_data import DATACLASS_DICT

    for path, extra in ((VAR_MISSING, []), (TILING_MISSING, ["--posctrl-col="])):
        out = str(tmp_path / (os.path.basename(path) + ".masked.h5ad"))
        assert bean_main(["qc", path, "-o", out, "-r", str(tmp_path / "rep"), "--count-correlation-thres", "0.6", "-b",
                          *extra]) == 0
        b = read_h5ad(out)
        counts = b.samples[["replicate", "condition"]].astype(str).value_counts()
        assert (counts == 1).all() and len(counts) == b.samples["replicate"].nunique() * b.samples["condition"].nunique()
        zero = np.asarray(b.X).sum(axis=0) == 0
        assert zero.any() and (b.samples["mask"].to_numpy()[zero] == 0).all()
        assert "edits" in b.layers and b.layers["edits"].shape == b.X.shape
    # -i keeps the screen as it is; an explicit window changes the tiling edit counts
    out_i = str(tmp_path / "ignore.h5ad")
    assert bean_main(["qc", TILING_MISSING, "-o", out_i, "-r", str(tmp_path / "rep"), "--count-correlation-thres", "0.6",
                      "--posctrl-col=", "-i"]) == 0
    assert read_h5ad(out_i).n_vars == read_h5ad(TILING_MISSING).n_vars
    out_w = str(tmp_path / "window.h5ad")
    assert bean_main(["qc", TILING_MISSING, "-o", out_w, "-r", str(tmp_path / "rep"), "--count-correlation-thres", "0.6",
                      "--posctrl-col=", "-i", "--edit-start-pos", "0", "--edit-end-pos", "20"]) == 0
    assert read_h5ad(out_w).layers["edits"].sum() > read_h5ad(out_i).layers["edits"].sum()
    out_d = str(tmp_path / "stored.h5ad")
    assert bean_main(["qc", TILING_MISSING, "-o", out_d, "-r", str(tmp_path / "rep"), "--count-correlation-thres", "0.6",
                      "--posctrl-col=", "-i", "--dont-recalculate-edits", "--edit-start-pos", "0", "--edit-end-pos", "20"]) == 0
    assert np.array_equal(read_h5ad(out_d).layers["edits"], read_h5ad(TILING_MISSING).layers["edits"])


def test_dummy_sample_id_with_several_replicate_columns():
    """The reference turns a multi-column replicate label into a LIST before it formats the dummy sample's
    id (bean/qc/utils.py:139, 96): "['r2', 'x']_bot", not a tuple's repr."""
    from bean_amd.qc.sample_qc import fill_in_missing_samples

    scr = _screen(n_guides=200)
    scr.samples["batch"] = ["x"] * 6 + ["y"] * 3
    keep = [i for i, n in enumerate(scr.samples.index) if n != "r2_bot"]
    scr = scr[:, scr.samples.index[keep]]
    out = fill_in_missing_samples(scr, "condition", ["replicate", "batch"])
    new = [n for n in out.samples.index if n not in scr.samples.index]
    assert new == ["['r2', 'x']_bot"]
    row = out.samples.loc[new[0]]
    assert row["replicate"] == "r2" and row["batch"] == "x" and row["condition"] == "bot"
    assert float(np.asarray(out.X)[:, list(out.samples.index).index(new[0])].sum()) == 0.0

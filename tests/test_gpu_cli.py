"""End-to-end `bean run` on the MI355X, mirroring the reference's black-box
tests for the variant library design (tests/test_run.py:5-80,164-213 there:
`--n-iter 10`, pass = clean exit) and additionally checking the output tables.
BASELINE config 1 (LDLvar mini-screen plumbing)."""
import os

import numpy as np
import pandas as pd
import pytest

import bean_amd  # noqa: F401
from bean_amd.cli.execute import main as bean_main
from bean_amd.framework import h5ad_io

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _h5ad_reader_present():
    """These tests must not vanish on a box that cannot read .h5ad: fail, do not skip."""
    try:
        import h5py  # noqa: F401
    except ImportError:
        assert os.path.exists(h5ad_io.HELPER_PYTHON), (
            f"no h5py in this interpreter and no helper interpreter at {h5ad_io.HELPER_PYTHON} "
            "(set BEAN_H5PY_PYTHON): `bean run` cannot read .h5ad screens here")
GOLD = os.path.join(os.path.dirname(__file__), "golden")
VAR = os.path.join(GOLD, "var_mini_screen.h5ad")
SURV = os.path.join(GOLD, "survival_var_mini_screen.h5ad")
TILING = os.path.join(GOLD, "tiling_mini_screen.h5ad")


def _run(tmp_path, *argv):
    out = str(tmp_path)
    assert bean_main(["run", *argv, "-o", out, "--sample-mask-col", ""]) == 0
    (d,) = [os.path.join(out, p) for p in os.listdir(out) if p.startswith("bean_run_result.")]
    return d


@pytest.mark.parametrize("extra,label", [
    ([], "MixtureNormal"),
    (["--fit-negctrl"], "MixtureNormal"),
    (["--uniform-edit"], "Normal"),
    (["--uniform-edit", "--fit-negctrl"], "Normal"),
    (["--dont-fit-noise"], "_MixtureNormal"),
])
def test_sorting_variant_runs(tmp_path, extra, label):
    d = _run(tmp_path, "sorting", "variant", VAR, "--n-iter", "10", *extra)
    el = pd.read_csv(f"{d}/bean_element_result.{label}.csv")
    sg = pd.read_csv(f"{d}/bean_sgRNA_result.{label}.csv")
    assert len(el) == 6 and len(sg) == 30
    assert {"target", "n_guides", "mu", "mu_sd", "mu_z", "sd", "CI[0.025", "0.975]"} <= set(el.columns)
    assert np.isfinite(el[["mu", "mu_sd", "mu_z", "sd"]].values).all() and (el["mu_sd"] > 0).all()
    assert {"rep5.top_bot.lfc", "rep6.top_bot.lfc"} <= set(sg.columns)
    if "--fit-negctrl" in extra:
        assert {"mu_scaled", "mu_z_scaled", "novl_scaled"} <= set(el.columns)
    if "--uniform-edit" not in extra:
        assert "edit_rate" in sg.columns and "edit_rate_mean" in el.columns
    assert os.path.exists(f"{d}/bean_run.log")


@pytest.mark.parametrize("extra,label", [([], "MixtureNormal+Acc"), (["--fit-negctrl"], "MixtureNormal+Acc")])
def test_sorting_variant_with_accessibility_track(tmp_path, extra, label):
    """tests/test_run.py:7,46 of the reference: `--scale-by-acc --acc-bw-path <bigWig>`."""
    d = _run(tmp_path, "sorting", "variant", VAR, "--n-iter", "10", "--scale-by-acc", "--acc-bw-path",
             os.path.join(GOLD, "accessibility_signal_chr6.bw"), "--repguide-mask", "None", *extra)
    sg = pd.read_csv(f"{d}/bean_sgRNA_result.{label}.csv")
    assert {"accessibility", "scaled_edit_eff"} <= set(sg.columns)
    assert np.isfinite(sg["accessibility"]).all() and sg["accessibility"].nunique() > 1


def test_sorting_tiling_with_accessibility_track(tmp_path):
    """tests/test_run.py:85,124: tiling with `--scale-by-acc --acc-bw-path`."""
    d = _run(tmp_path, "sorting", "tiling", TILING, "--n-iter", "10", "--repguide-mask", "None",
             "--allele-df-key", "allele_counts", "--control-guide-tag", "None", "--scale-by-acc",
             "--acc-bw-path", os.path.join(GOLD, "accessibility_signal.bw"))
    el = pd.read_csv(f"{d}/bean_element_result.MultiMixtureNormal+Acc.csv")
    assert len(el) > 20 and np.isfinite(el[["mu", "mu_sd", "mu_z", "sd"]].values).all()


def test_sorting_variant_with_accessibility_column(tmp_path):
    # the fixture has no accessibility track: add a column and write nothing back - the
    # CLI reads the file, so patch the reader's result through a small wrapper file
    import pickle
    import subprocess
    import sys

    from bean_amd.framework import read_h5ad
    from bean_amd.cli import run as cli_run

    s = read_h5ad(VAR)
    s.guides["acc"] = np.linspace(1.0, 40.0, len(s.guides))
    orig = cli_run.read_h5ad
    cli_run.read_h5ad = lambda path: s.copy()
    try:
        d = _run(tmp_path, "sorting", "variant", VAR, "--n-iter", "10", "--scale-by-acc", "--acc-col", "acc")
    finally:
        cli_run.read_h5ad = orig
    sg = pd.read_csv(f"{d}/bean_sgRNA_result.MixtureNormal+Acc.csv")
    assert {"accessibility", "scaled_edit_eff"} <= set(sg.columns)
    assert ((sg["scaled_edit_eff"] >= 1e-3) & (sg["scaled_edit_eff"] <= 1 - 1e-3)).all()


@pytest.mark.parametrize("extra", [[], ["--fit-negctrl"], ["--uniform-edit"], ["--uniform-edit", "--fit-negctrl"]])
def test_survival_variant_runs(tmp_path, extra):
    d = _run(tmp_path, "survival", "variant", SURV, "--n-iter", "10", "--control-condition=D7", *extra)
    label = "Normal" if "--uniform-edit" in extra else "MixtureNormal"
    el = pd.read_csv(f"{d}/bean_element_result.{label}.csv")
    # this data file lists target var_1 under two target groups, so the target table has one row more
    # than there are fitted targets (13): the reference's column-wise concat pads the last row with NaN
    fitted = el.dropna(subset=["mu"])
    assert len(el) == 14 and len(fitted) == 13 and "sd" not in el.columns
    assert np.isfinite(fitted[["mu", "mu_sd", "mu_z"]].values).all() and (fitted["mu_sd"] > 0).all()


@pytest.mark.parametrize("extra,label", [
    ([], "MultiMixtureNormal"),
    (["--uniform-edit"], "MultiMixtureNormal"),  # tiling ignores --uniform-edit (bean/model/run.py:404-419)
    (["--fit-negctrl", "--negctrl-col", "strand", "--negctrl-col-value", "neg", "--control-guide-tag", "neg"],
     "MultiMixtureNormal"),
    (["--uniform-edit", "--fit-negctrl", "--negctrl-col", "strand", "--negctrl-col-value", "neg",
      "--control-guide-tag", "neg"], "MultiMixtureNormal"),
])
def test_sorting_tiling_runs(tmp_path, extra, label):
    """The reference's `bean run sorting tiling` invocations (tests/test_run.py:85-150, 218) on its
    tiling mini-screen with the unfiltered allele table (`--allele-df-key allele_counts`): up to 230
    alleles per guide there, all kept (allele-parallel kernels)."""
    argv = ["sorting", "tiling", TILING, "--n-iter", "10", "--repguide-mask", "None",
            "--allele-df-key", "allele_counts"]
    if "--control-guide-tag" not in extra:
        argv += ["--control-guide-tag", "None"]
    d = _run(tmp_path, *argv, *extra)
    el = pd.read_csv(f"{d}/bean_element_result.{label}.csv")
    sg = pd.read_csv(f"{d}/bean_sgRNA_result.{label}.csv")
    assert len(sg) == 30
    if label == "MultiMixtureNormal":
        assert {"edit", "group", "coding", "effective_edit_rate", "editing_guides", "n_guides", "n_coocc",
                "mu", "mu_sd", "mu_z", "sd"} <= set(el.columns)
        assert len(el) > 20 and np.isfinite(el[["mu", "mu_sd", "mu_z", "sd"]].values).all()
        assert (el["n_guides"] >= 1).all() and (el["effective_edit_rate"] >= 0).all()
        assert "variants" in sg.columns
    if "--fit-negctrl" in extra:
        assert {"mu_scaled", "mu_z_scaled"} <= set(el.columns)


SURV_TILING = os.path.join(os.path.dirname(__file__), "golden", "survival_tiling_mini_screen.h5ad")


@pytest.mark.parametrize("extra", [[], ["--fit-negctrl", "--negctrl-col", "strand", "--negctrl-col-value", "neg"]])
def test_survival_tiling_runs(tmp_path, extra):
    """`bean run survival tiling` on the reference's survival tiling mini-screen (the reference ships
    the data file but no test invocation for it)."""
    argv = ["survival", "tiling", SURV_TILING, "--n-iter", "10", "--repguide-mask", "None",
            "--allele-df-key", "allele_counts", "--control-guide-tag", "None", "--control-condition=D7"]
    d = _run(tmp_path, *argv, *extra)
    el = pd.read_csv(f"{d}/bean_element_result.MultiMixtureNormal.csv")
    sg = pd.read_csv(f"{d}/bean_sgRNA_result.MultiMixtureNormal.csv")
    assert len(sg) == 30 and len(el) > 20 and "sd" not in el.columns
    assert np.isfinite(el[["mu", "mu_sd", "mu_z"]].values).all() and (el["mu_sd"] > 0).all()


def test_longer_fit_moves_parameters_and_saves_raw(tmp_path):
    import pickle

    d = _run(tmp_path, "sorting", "variant", VAR, "--n-iter", "300", "--fit-negctrl", "--save-raw")
    raw = pickle.load(open(f"{d}/MixtureNormal.result.pkl", "rb"))
    assert {"negctrl", "loss", "params"} <= set(raw)
    loss = raw["loss"]
    assert len(loss) == 300 and loss[-1] < loss[0]
    assert set(raw["params"]) == {"mu_loc", "mu_scale", "sd_loc", "sd_scale", "alpha_pi"}
    assert raw["params"]["mu_loc"].shape == (6, 1) and raw["params"]["alpha_pi"].shape == (30, 2)
    # --load-existing re-writes the same tables from the pickle without fitting
    el = pd.read_csv(f"{d}/bean_element_result.MixtureNormal.csv")
    os.remove(f"{d}/bean_element_result.MixtureNormal.csv")
    d2 = _run(tmp_path, "sorting", "variant", VAR, "--n-iter", "300", "--fit-negctrl", "--load-existing")
    assert d2 == d
    pd.testing.assert_frame_equal(pd.read_csv(f"{d}/bean_element_result.MixtureNormal.csv"), el)


def test_build_prior_then_run_with_prior_params(tmp_path):
    """`bean build-prior` (bean/cli/build_prior.py): the first batch's posterior of the shared variants
    becomes the prior of the second batch, which then runs with `--prior-params`."""
    import pickle

    out1 = str(tmp_path / "b1")
    d1 = _run(tmp_path / "b1", "sorting", "variant", VAR, "--n-iter", "40", "--save-raw")
    raw = f"{d1}/MixtureNormal.result.pkl"
    saved = pickle.load(open(raw, "rb"))
    assert "data" in saved and saved["data"].n_targets == 6
    prior_path = str(tmp_path / "prior.pkl")
    # the data file has no `mask` column: an empty --sample-mask-col (as `_run` passes) switches it off
    cmd = f"bean run sorting variant {VAR} -o {out1} --n-iter 10 --sample-mask-col "
    assert bean_main(["build-prior", cmd, cmd, raw, prior_path]) == 0
    prior = pickle.load(open(prior_path, "rb"))
    assert set(prior) == {"mu_loc", "mu_scale", "sd_loc", "sd_scale"}
    for k in prior:  # every variant is shared here: the prior is the first run's posterior
        assert tuple(prior[k].shape) == (6, 1)
        np.testing.assert_allclose(prior[k].numpy().ravel(), np.asarray(saved["params"][k]).ravel(), rtol=1e-6)
    d2 = _run(tmp_path / "b2", "sorting", "variant", VAR, "--n-iter", "10", "--prior-params", prior_path)
    el = pd.read_csv(f"{d2}/bean_element_result.MixtureNormal.csv")
    assert len(el) == 6 and np.isfinite(el[["mu", "mu_sd", "mu_z", "sd"]].values).all()


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_rank_cli_writes_the_single_process_tables(tmp_path):
    """`torchrun --nproc-per-node 2 bin/bean run ...` (rehearsed on one GPU over gloo): the guides
    are sharded over the ranks, rank 0 writes the tables; for the variant sorting family the result
    is the single-process result bit for bit (also with per-target `--prior-params`, which every rank
    cuts to its own targets), for tiling up to the regrouping of float64 sums."""
    import pickle
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BEAN_DIST_BACKEND="gloo", BEAN_DIST_SINGLE_DEVICE="1")
    g = torch.Generator().manual_seed(3)
    prior = {"mu_loc": torch.randn((6, 1), generator=g) * 0.3, "mu_scale": torch.rand((6, 1), generator=g) + 0.5,
             "sd_loc": torch.randn((6, 1), generator=g) * 0.1, "sd_scale": torch.rand((6, 1), generator=g) * 0.05 + 0.01}
    prior_path = str(tmp_path / "prior.pkl")
    with open(prior_path, "wb") as fh:
        pickle.dump(prior, fh)
    for tag, design, path, extra, label in (
        ("variant", "variant", VAR, [], "MixtureNormal"),
        ("prior", "variant", VAR, ["--prior-params", prior_path], "MixtureNormal"),
        ("tiling", "tiling", TILING, ["--allele-df-key", "allele_counts", "--control-guide-tag", "None"],
         "MultiMixtureNormal"),
    ):
        argv = ["sorting", design, path, "--n-iter", "30", "--repguide-mask", "None", *extra]
        d1 = _run(tmp_path / f"single_{tag}", *argv)
        out2 = str(tmp_path / f"two_{tag}")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bin", "bean"),
               "run", *argv, "-o", out2, "--sample-mask-col", ""]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        (d2,) = [os.path.join(out2, p) for p in os.listdir(out2) if p.startswith("bean_run_result.")]
        a = pd.read_csv(f"{d1}/bean_element_result.{label}.csv")
        b = pd.read_csv(f"{d2}/bean_element_result.{label}.csv")
        assert list(a.columns) == list(b.columns) and len(a) == len(b)
        num = [c for c in ("mu", "mu_sd", "mu_z", "sd") if c in a.columns]
        if design == "variant":
            pd.testing.assert_frame_equal(a, b)
        else:
            np.testing.assert_allclose(a[num].values, b[num].values, rtol=1e-5, atol=1e-7)


def test_qc_then_run_with_default_mask_columns(tmp_path):
    """`bean qc` writes samples["mask"] / uns["repguide_mask"]; `bean run` then works
    with its DEFAULT --sample-mask-col / --repguide-mask (no `--sample-mask-col ""` escape)."""
    masked = str(tmp_path / "var_masked.h5ad")
    assert bean_main(["qc", VAR, "-o", masked, "-r", str(tmp_path / "qc"), "--count-correlation-thres", "0.5"]) == 0
    out = str(tmp_path / "run")
    assert bean_main(["run", "sorting", "variant", masked, "-o", out, "--n-iter", "20"]) == 0
    (d,) = [os.path.join(out, p) for p in os.listdir(out) if p.startswith("bean_run_result.")]
    el = pd.read_csv(f"{d}/bean_element_result.MixtureNormal.csv")
    assert len(el) == 6 and np.isfinite(el[["mu", "mu_sd", "mu_z", "sd"]].values).all()
    sg = pd.read_csv(f"{d}/bean_sgRNA_result.MixtureNormal.csv")
    assert "edit_rate" in sg.columns and len(sg) == 30

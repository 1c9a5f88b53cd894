"""Sample covariates of the sorting NormalModel (`uns["sample_covariates"]`): the data-class side
(bean/preprocessing/data_class.py:75-92, 972-979) on CPU, and the HIP path against the oracle
(bean/model/model.py:73-91, 771-783) on the GPU."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.framework.ReporterScreen import ReporterScreen
from bean_amd.preprocessing.screen_data import DATACLASS_DICT
from oracle import elbo, svi


def _screen_with_covariates(n_targets=40, seed=0):
    """3 replicates x 2 batches ("b0" / "b1") x (4 sort bins + bulk), guides sorted by target."""
    rng = np.random.default_rng(seed)
    conds = [("bot", 0.0, 0.2), ("low", 0.2, 0.4), ("high", 0.6, 0.8), ("top", 0.8, 1.0), ("bulk", 0.0, 1.0)]
    rows = []
    for rep in ("r1", "r2", "r3"):
        for batch in ("0", "1"):
            for cname, lo, hi in conds:
                rows.append(dict(name=f"{rep}.{batch}_{cname}", replicate=rep, batch=batch, condition=cname,
                                 lower_quantile=lo, upper_quantile=hi, mask=1))
    samples = pd.DataFrame(rows).set_index("name")
    G = n_targets * 3
    guides = pd.DataFrame({"target": np.repeat([f"t{i:03d}" for i in range(n_targets)], 3),
                           "target_group": "Variant"}, index=[f"g{i:04d}" for i in range(G)])
    X = rng.poisson(120, size=(G, len(samples))).astype(np.float32) + 12
    return ReporterScreen(X, guides, samples, layers={"X_bcmatch": np.floor(X * 0.8)},
                          uns={"sample_covariates": ["batch"], "tiling": False})


def test_data_class_builds_replicate_by_covariate_design():
    scr = _screen_with_covariates()
    data = DATACLASS_DICT["sorting"]["Normal"](scr, sample_mask_column="mask", control_condition="bulk",
                                               use_bcmatch=True)
    # a "replicate" is a (replicate, batch) combination: 3 x 2 = 6, ordered by the joined label
    assert data.n_reps == 6 and data.n_condits == 5 and data.n_guides == 120
    assert data.sample_covariates == ["batch"] and data.n_sample_covariates == 1
    assert data.rep_by_cov.shape == (6, 1)
    assert data.rep_by_cov.reshape(-1).tolist() == [0, 1, 0, 1, 0, 1]  # r1.0, r1.1, r2.0, ...
    # the oracle runs on it, and the covariate shifts the likelihood of the batch-1 replicates only
    params = elbo.init_params("Normal", data)
    assert set(params) == {"mu_loc", "mu_scale", "sd_loc", "sd_scale", "mu_cov_loc", "mu_cov_scale"}
    z = {"eps_mu": torch.zeros(data.n_targets, 1), "eps_sd": torch.zeros(data.n_targets, 1)}
    l0 = svi.loss_and_grads(elbo.normal_loss, data, params, noise=dict(z, eps_cov=torch.zeros(1)))[0]
    l1, g1, _ = svi.loss_and_grads(elbo.normal_loss, data, params, noise=dict(z, eps_cov=torch.tensor([0.7])))
    assert np.isfinite(l0) and np.isfinite(l1) and l1 != l0
    assert g1["mu_cov_loc"].abs().max() > 0


def test_without_the_key_nothing_changes():
    scr = _screen_with_covariates()
    del scr.uns["sample_covariates"]
    scr.samples["replicate"] = scr.samples["replicate"] + "." + scr.samples["batch"]
    data = DATACLASS_DICT["sorting"]["Normal"](scr, sample_mask_column="mask", control_condition="bulk")
    assert data.n_reps == 6 and getattr(data, "sample_covariates", None) is None


# ------------------------------------------------------------------ GPU
def _synthetic_with_covariates(n_guides, n_reps, n_cov, seed):
    from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

    data = make_sorting_variant_screen(n_guides, n_reps, seed=seed, mask_fraction=0.05)
    g = torch.Generator().manual_seed(seed)
    data.sample_covariates = [f"cov{i}" for i in range(n_cov)]
    data.n_sample_covariates = n_cov
    data.rep_by_cov = torch.randint(0, 2, (n_reps, n_cov), generator=g)
    data.rep_by_cov[0, 0], data.rep_by_cov[-1, 0] = 1, 0
    return data


@pytest.mark.gpu
@pytest.mark.parametrize("n_guides,n_reps,n_cov,kw", [(1500, 4, 1, {}), (700, 3, 3, dict(use_bcmatch=False)),
                                                      (65, 2, 2, {})])
def test_normal_with_covariates_matches_oracle(n_guides, n_reps, n_cov, kw):
    from bean_amd import engine

    DEV = "cuda:0"
    data = _synthetic_with_covariates(n_guides, n_reps, n_cov, seed=17)
    torch.manual_seed(3)
    eng = engine.HipSVI("Normal", data.to(DEV), dump_noise=True, num_steps=50, **kw)
    assert set(eng.unconstrained) == {"mu_loc", "mu_scale", "sd_loc", "sd_scale", "mu_cov_loc", "mu_cov_scale"}
    for v in eng.unconstrained.values():
        v.add_(0.3 * torch.randn_like(v))
    loss, grads = eng.elbo_grad(step=2, seed=9)
    draws = {k: v.cpu() for k, v in eng.drawn_noise().items()}
    assert "eps_cov" in draws and draws["eps_cov"].numel() == n_cov
    for mode, tl, tg in (("f64", 1e-9, 5e-7), ("ref", 1e-6, 2e-5)):
        params = {k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()}
        d = data
        if mode == "f64":
            params = {k: v.double() for k, v in params.items()}
            d = elbo.as_float64(data)
        params = {k: v.requires_grad_(True) for k, v in params.items()}
        ref_loss, ref_grads, _ = svi.loss_and_grads(elbo.normal_loss, d, params, noise=draws, **kw)
        assert abs(loss - ref_loss) <= tl * abs(ref_loss), (mode, loss, ref_loss)
        for k, g in grads.items():
            ref = ref_grads[k].double().reshape(-1)
            err = (g.cpu().double().reshape(-1) - ref).abs().max().item()
            assert err <= tg * (ref.abs().max().item() + 1e-30), (mode, k, err, ref.abs().max().item())
    eng.set_noise(draws)
    loss_b, grads_b = eng.elbo_grad(step=2, seed=9)
    assert abs(loss_b - loss) <= 1e-12 * abs(loss)
    for k in grads:
        assert torch.equal(grads[k], grads_b[k])
    eng.close()


@pytest.mark.gpu
def test_covariate_trajectory_fused_loop_and_result_table(tmp_path):
    from functools import partial

    from bean_amd import engine
    from bean_amd.model import model as m
    from bean_amd.model.readwrite import write_result_table
    from bean_amd.model.run import run_inference

    DEV = "cuda:0"
    data = _synthetic_with_covariates(600, 4, 2, seed=23)
    n = 15
    eng = engine.HipSVI("Normal", data.to(DEV), dump_noise=True, num_steps=2000)
    params = elbo.init_params("Normal", data)
    optim = svi.ClippedAdam(params, lr=0.01, lrd=0.1 ** (1 / 2000))
    for t in range(n):
        loss, _ = eng.elbo_grad(step=t, seed=5, loss_index=t)
        draws = {k: v.cpu() for k, v in eng.drawn_noise().items()}
        eng.adam(t + 1)
        ref = svi.svi_step(elbo.normal_loss, data, params, optim, noise=draws)
        assert abs(loss - ref) <= 2e-6 * abs(ref), (t, loss, ref)
    torch.cuda.synchronize()
    for k, v in eng.unconstrained.items():
        ref = params[k].detach()
        err = (v.cpu() - ref).abs().max().item()
        assert err <= 1e-4 * max(1.0, ref.abs().max().item()), (k, err)
    fused = engine.HipSVI("Normal", data.to(DEV), num_steps=2000)
    fused.run(n, seed=5, graph_chunk=4)
    for k in eng.unconstrained:
        assert torch.equal(eng.unconstrained[k], fused.unconstrained[k]), k
    eng.close()
    fused.close()
    # the run_inference interface and the covariate columns of the element table (readwrite.py:87-99)
    store, out = run_inference(partial(m.NormalModel), m.NormalGuide, data, num_steps=200, verbose=False)
    assert {"mu_cov_loc", "mu_cov_scale"} <= set(store.keys()) and store["mu_cov_loc"].shape == (2,)
    assert out["loss"][-1] < out["loss"][0]
    target_info = pd.DataFrame(index=pd.Index([f"t{i}" for i in range(data.n_targets)], name="target"))
    guide_info = pd.DataFrame(index=pd.Index([f"g{i}" for i in range(data.n_guides)], name="name"))
    write_result_table(target_info, guide_info, store, model_label="Normal", prefix=str(tmp_path) + "/",
                       sample_covariates=data.sample_covariates, sd_is_fitted=True,
                       adjust_confidence_by_negative_control=False)
    el = pd.read_csv(tmp_path / "bean_element_result.Normal.csv")
    assert {"mu_cov0", "mu_sd_cov0", "mu_z_cov0", "mu_cov1"} <= set(el.columns)


@pytest.mark.gpu
def test_other_families_refuse_sample_covariates():
    from bean_amd import engine
    from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

    data = make_sorting_variant_screen(200, 2, seed=1)
    data.sample_covariates, data.n_sample_covariates = ["batch"], 1
    data.rep_by_cov = torch.tensor([[0], [1]])
    with pytest.raises(ValueError, match="sample_covariates"):
        engine.HipSVI("MixtureNormal", data.to("cuda:0"), num_steps=5)


@pytest.mark.gpu
def test_control_normal_ignores_sample_covariates():
    """``--fit-negctrl`` on a screen with sample covariates: the reference's ControlNormalModel / Guide have
    no mu_cov site (model.py:168-252, 861-875) and fit the regrouped (replicate, covariate) replicates;
    the negative-control subset keeps the attribute (``ScreenTensors.__getitem__``) and must not be
    refused.  The fit equals the one of the same screen without the attribute."""
    import numpy as np

    from bean_amd import engine

    data = _synthetic_with_covariates(900, 4, 1, seed=29)
    neg = data[data.negctrl_guide_idx]
    assert getattr(neg, "sample_covariates", None) is not None
    a = engine.HipSVI("ControlNormal", neg.to("cuda:0"), num_steps=100)
    a.run(30, seed=3)
    plain = data[data.negctrl_guide_idx]
    plain.sample_covariates = None
    b = engine.HipSVI("ControlNormal", plain.to("cuda:0"), num_steps=100)
    b.run(30, seed=3)
    assert set(a.constrained()) == {"mu_loc", "mu_scale", "sd_loc", "sd_scale"}
    for k, v in a.constrained().items():
        assert torch.equal(v, b.constrained()[k]), k
    assert a.losses() == b.losses()
    a.close()
    b.close()


@pytest.mark.gpu
def test_guide_shards_share_mu_cov():
    """Sample covariates in a guide-sharded fit: mu_cov is shared by every guide, so the replicates'
    gradient sums are exchanged per step (BEAN_BUF_XCHG_COV) and every shard applies the same update;
    per-target parameters stay shard-local.  Shards as engines in one process, the all-reduce done by
    hand between the phases, as in tests/test_gpu_sharded_exchange.py."""
    import numpy as np

    from bean_amd import engine, parallel

    DEV = "cuda:0"
    data = _synthetic_with_covariates(1200, 4, 2, seed=31)
    n = 25
    whole = engine.HipSVI("Normal", data.to(DEV), num_steps=200)
    whole.run(n, seed=9)
    ref, ref_loss = whole.constrained(), np.array(whole.losses())
    shards = parallel.plan_shards(data.target_lengths.numpy(), 3)
    engines = [engine.HipSVI("Normal", parallel.shard_screen(data, sh).to(DEV), num_steps=200, guide_offset=sh[0],
                             target_offset=sh[2], n_guides_total=data.n_guides, loss_owner=(k == 0))
               for k, sh in enumerate(shards)]
    for e in engines:
        assert set(e.exchange_buffers()) == {"cov"}
        e.phase("begin", 9, 0, n)
    for i in range(n):
        for e in engines:
            e.phase("sums")
            e.phase("guide")
        torch.cuda.synchronize()
        tot = torch.stack([e.exchange_buffers()["cov"] for e in engines]).sum(0)
        for e in engines:
            e.exchange_buffers()["cov"].copy_(tot)
        torch.cuda.synchronize()
        for e in engines:
            e.phase("update", 1 if i == n - 1 else 0)
    torch.cuda.synchronize()
    for e in engines:
        e.steps_done = n
    for name in ("mu_cov_loc", "mu_cov_scale"):
        for e in engines:
            assert torch.equal(e.constrained()[name], engines[0].constrained()[name]), name
        err = (engines[0].constrained()[name].double() - ref[name].double()).abs().max().item()
        assert err <= 2e-5, (name, err)
    for name in ("mu_loc", "mu_scale", "sd_loc", "sd_scale"):
        got = torch.cat([e.constrained()[name] for e in engines])
        err = (got.double() - ref[name].double()).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref[name].abs().max().item()), (name, err)
    np.testing.assert_allclose(sum(np.array(e.losses()) for e in engines), ref_loss, rtol=1e-6)
    for e in engines + [whole]:
        e.close()


@pytest.mark.gpu
def test_cli_uniform_edit_fit_negctrl_on_a_screen_with_sample_covariates(tmp_path):
    """`bean run sorting variant --uniform-edit --fit-negctrl` on an .h5ad with uns["sample_covariates"]: the
    negative-control fit (ControlNormal, no mu_cov site) runs on the regrouped replicates, the main fit
    models the covariate, and the element table carries the covariate and the scaled columns."""
    from bean_amd.cli.execute import main as bean_main
    from bean_amd.framework import h5ad_io

    scr = _screen_with_covariates(n_targets=60, seed=3)
    scr.guides["target_group"] = np.where(np.arange(len(scr.guides)) < 45, "NegCtrl", "Variant")
    # a replicate x guide mask must name the regrouped (replicate, covariate) replicates, here as in the
    # reference (data_class.py:165-177 asserts its columns against samples["_rc"].unique())
    rc = [f"{rep}.{b}" for rep in ("r1", "r2", "r3") for b in ("0", "1")]
    scr.uns["repguide_mask"] = pd.DataFrame(1, index=scr.guides.index, columns=rc)
    path = str(tmp_path / "cov_screen.h5ad")
    h5ad_io.write_screen(scr, path)
    out = str(tmp_path / "out")
    assert bean_main(["run", "sorting", "variant", path, "--uniform-edit", "--fit-negctrl", "--n-iter", "20",
                      "-o", out]) == 0
    (d,) = [os.path.join(out, q) for q in os.listdir(out) if q.startswith("bean_run_result.")]
    el = pd.read_csv(f"{d}/bean_element_result.Normal.csv")
    assert len(el) == 60
    assert {"mu", "mu_sd", "mu_z", "mu_scaled", "mu_z_scaled", "mu_batch"} <= set(el.columns), list(el.columns)
    assert np.isfinite(el[["mu", "mu_sd", "mu_scaled"]].values).all()

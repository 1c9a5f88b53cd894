"""The one-launch SVI step (k_step_wave2, csrc/bean_step_v2.hpp; opt-in, BEAN_HIP_STEP=fused) against the
default two-launch path (k_param + k_guide_wave2): same draws, same arithmetic, same summation order, so
the fitted parameters must be bit-identical; the loss history agrees to rounding (its per-target
terms are summed in another order).  -m gpu."""
import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fit(monkeypatch, mode, family, data, steps, eng_kw, chunks=None):
    from bean_amd import engine

    if mode == "pair":  # the product library
        monkeypatch.setenv("BEAN_HIP_STEP", "pair")  # (large screens default to k_svi_async)
        eng = engine.HipSVI(family, data.to(DEV), num_steps=steps, **eng_kw)
    else:  # the opt-in kernel lives in the A/B library (libbean_hip_ab.so)
        monkeypatch.setenv("BEAN_HIP_STEP", "fused")
        eng = engine.HipSVI(family, data.to(DEV), num_steps=steps, lib_variant="ab", **eng_kw)
    assert data.n_targets >= 64  # fewer targets: k_param's one-block-per-target mode, two launches
    assert eng.dominant_kernel == ("k_guide_wave2" if mode == "pair" else "k_step_wave2")
    for n in (chunks or [steps]):
        eng.run(n, seed=5)
    torch.cuda.synchronize()
    out = {k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()}
    loss = np.array(eng.losses())
    eng.close()
    return out, loss


def _same(monkeypatch, family, data, steps, eng_kw=None, chunks=None):
    a, la = _fit(monkeypatch, "pair", family, data, steps, eng_kw or {}, chunks)
    b, lb = _fit(monkeypatch, "fused", family, data, steps, eng_kw or {}, chunks)
    assert np.all(np.isfinite(la)) and len(la) == steps == len(lb)
    for k in a:
        assert torch.equal(a[k], b[k]), (k, (a[k] - b[k]).abs().max().item())
    assert np.max(np.abs(la - lb) / np.abs(la)) < 1e-12


@pytest.mark.parametrize("n_guides,n_reps,gpt", [(64, 1, 1), (200, 2, 3), (333, 1, 5), (461, 3, 7), (130, 5, 1),
                                                 (1300, 9, 5), (6400, 3, 64), (3000, 4, 33), (4097, 2, 5)])
def test_fused_step_is_bitwise_the_pair_path(monkeypatch, n_guides, n_reps, gpt):
    data = make_sorting_variant_screen(n_guides, n_reps, seed=300 + n_guides, guides_per_target=gpt,
                                       mask_fraction=0.05 if (n_guides > 100 and n_reps > 1) else 0.0)
    _same(monkeypatch, "MixtureNormal", data, 37)


@pytest.mark.parametrize("steps,chunks", [(1, None), (2, None), (3, None), (150, None), (131, [1, 2, 64, 64])])
def test_fused_step_counts_and_graph_replay(monkeypatch, steps, chunks):
    data = make_sorting_variant_screen(2500, 3, seed=77, guides_per_target=5)
    _same(monkeypatch, "MixtureNormal", data, steps, chunks=chunks)


def test_fused_step_families(monkeypatch):
    data = make_sorting_variant_screen(1800, 3, seed=78, with_accessibility=True, mask_fraction=0.05)
    _same(monkeypatch, "MixtureNormal", data, 40, dict(scale_by_accessibility=True))
    _same(monkeypatch, "MixtureNormal", data, 40, dict(scale_by_accessibility=True, fit_noise=False))
    _same(monkeypatch, "Normal", data, 40)
    _same(monkeypatch, "Normal", data, 40, dict(use_bcmatch=False))
    T = data.n_targets
    g = torch.Generator().manual_seed(0)
    prior = {
        "mu_loc": torch.randn((T, 1), generator=g, dtype=torch.float64) * 0.2,
        "mu_scale": torch.rand((T, 1), generator=g, dtype=torch.float64) + 0.5,
        "sd_loc": torch.randn((T, 1), generator=g, dtype=torch.float64) * 0.1,
        "sd_scale": torch.rand((T, 1), generator=g, dtype=torch.float64) * 0.05 + 0.01,
    }
    _same(monkeypatch, "MixtureNormal", data, 40, dict(prior_params=prior))


def test_fused_step_metric_shape(monkeypatch):
    data = make_sorting_variant_screen(50000, 5, seed=79)
    _same(monkeypatch, "MixtureNormal", data, 130)


def test_targets_longer_than_a_tile_take_the_pair_path(monkeypatch):
    from bean_amd import engine

    monkeypatch.setenv("BEAN_HIP_STEP", "fused")
    data = make_sorting_variant_screen(6500, 2, seed=80, guides_per_target=65)
    eng = engine.HipSVI("MixtureNormal", data.to(DEV), num_steps=10, lib_variant="ab")
    assert eng.dominant_kernel == "k_guide_wave2"
    eng.run(5)
    assert np.all(np.isfinite(eng.losses()))
    eng.close()


def test_product_library_refuses_the_ab_switches(monkeypatch):
    """libbean_hip.so holds the default kernels only: a switch that selects a superseded form is an error
    that names the A/B library, not a silent fallback."""
    from bean_amd import engine

    data = make_sorting_variant_screen(1000, 2, seed=1)
    for key, val in (("BEAN_HIP_STEP", "fused"), ("BEAN_HIP_STEP", "tile"), ("BEAN_HIP_GUIDE", "wave1")):
        monkeypatch.setenv(key, val)
        with pytest.raises(RuntimeError, match="libbean_hip_ab.so"):
            engine.HipSVI("MixtureNormal", data.to(DEV), num_steps=10)
        monkeypatch.delenv(key)

"""k_svi_tile (opt-in, BEAN_HIP_STEP=tile: one launch per call, a workgroup per tile of targets,
csrc/bean_tile_svi.hpp) against the default two launches per step ({k_param, k_guide_wave2}), which the oracle tests pin.
Same draws and the same per-pair arithmetic (guide_pair_math is one function).  Until round 4 the summation
orders were the same too and the parameters BIT-IDENTICAL; since then k_guide_wave2 adds a target's guides up
inside its waves (bean_guide_v2.hpp: target_part_sums) while this kernel keeps the per-guide rows and the old
order, so the two paths differ by roundings of float64 sums: float32 parameters and moments agree to a few ulp,
the loss history to 1e-9.  -m gpu."""
import os

import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(family, data, pair, **kw):
    from bean_amd import engine

    old = os.environ.get("BEAN_HIP_STEP")
    if pair:
        os.environ["BEAN_HIP_STEP"] = "pair"
    else:
        os.environ["BEAN_HIP_STEP"] = "tile"
    try:  # the tile kernel lives in the A/B library; the reference path is the product library
        eng = engine.HipSVI(family, data.to(DEV), num_steps=2000, **({} if pair else {"lib_variant": "ab"}), **kw)
    finally:
        if old is None:
            os.environ.pop("BEAN_HIP_STEP", None)
        else:
            os.environ["BEAN_HIP_STEP"] = old
    return eng


def _compare(family, data, chunks=(1, 7, 30), **kw):
    a = _engine(family, data, pair=False, **kw)
    b = _engine(family, data, pair=True, **kw)
    assert a.dominant_kernel == "k_svi_tile" and b.dominant_kernel == "k_guide_wave2"
    for n in chunks:  # several calls: each begins with a PREP-only launch and ends without one
        a.run(n, seed=13)
        b.run(n, seed=13)
    torch.cuda.synchronize()
    for k in a.unconstrained:
        torch.testing.assert_close(a.unconstrained[k], b.unconstrained[k], rtol=2e-5, atol=2e-6, msg=k)
        torch.testing.assert_close(a._m[k], b._m[k], rtol=2e-4, atol=1e-5, msg=k)
        torch.testing.assert_close(a._v[k], b._v[k], rtol=2e-4, atol=1e-7, msg=k)
    la, lb = np.array(a.losses()), np.array(b.losses())
    assert np.isfinite(la).all() and la.shape == lb.shape
    np.testing.assert_allclose(la, lb, rtol=1e-9)
    a.close()
    b.close()


@pytest.mark.parametrize("n_guides,n_reps,gpt,kw", [
    (3000, 5, 5, {}),                                   # the metric layout: 10 targets / 50 guides x 5 replicates per tile
    (1000, 3, 4, dict(scale_by_accessibility=True)),    # +Acc, three replicates (85 guides per tile)
    (777, 2, 3, dict(scale_by_accessibility=True, fit_noise=False)),
    (500, 7, 6, {}),                                    # 36 guides per tile, ragged last tile
    (1500, 2, 5, dict(use_bcmatch=True)),                 # two replicates: 125 guides per tile
    (2100, 12, 5, {}),                                  # twelve replicates: 21 guides per tile
])
def test_mixture_normal_tile_path_equals_two_launch_path(n_guides, n_reps, gpt, kw):
    acc = kw.get("scale_by_accessibility", False)
    data = make_sorting_variant_screen(n_guides, n_reps, seed=100 + n_guides, guides_per_target=gpt, mask_fraction=0.05,
                                       with_accessibility=acc)
    _compare("MixtureNormal", data, **kw)


def test_normal_family_and_priors():
    data = make_sorting_variant_screen(1500, 4, seed=5, guides_per_target=5)
    _compare("Normal", data)
    T = data.n_targets
    g = torch.Generator().manual_seed(3)
    prior = {"mu_loc": torch.randn(T, 1, generator=g) * 0.1, "mu_scale": torch.rand(T, 1, generator=g) + 0.5,
             "sd_loc": torch.randn(T, 1, generator=g) * 0.05, "sd_scale": torch.rand(T, 1, generator=g) * 0.1 + 0.01}
    _compare("MixtureNormal", data, prior_params=prior)


def test_ragged_target_lengths_fall_into_whole_target_tiles():
    """Targets of 1 ... 9 guides: tiles hold whole targets, at most 256 / R guides."""
    data = make_sorting_variant_screen(900, 3, seed=21, guides_per_target=5)
    rng = np.random.default_rng(0)
    lens = []  # re-cut the same guides into ragged targets
    left = data.n_guides
    while left > 0:
        k = int(min(left, rng.integers(1, 10)))
        lens.append(k)
        left -= k
    data.target_lengths = torch.tensor(lens, dtype=torch.int64)
    data.n_targets = len(lens)
    data.validate()
    _compare("MixtureNormal", data)


def test_a_target_longer_than_a_tile_keeps_the_two_launch_path():
    from bean_amd import engine

    data = make_sorting_variant_screen(600, 5, seed=2, guides_per_target=60)  # 60 guides > 256 / 5
    os.environ["BEAN_HIP_STEP"] = "tile"
    try:
        eng = engine.HipSVI("MixtureNormal", data.to(DEV), num_steps=50, lib_variant="ab")
    finally:
        os.environ.pop("BEAN_HIP_STEP", None)
    assert eng.dominant_kernel == "k_guide_wave2"
    eng.run(5)
    assert np.isfinite(eng.losses()).all()
    eng.close()

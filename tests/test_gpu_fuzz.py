"""Random geometries of the variant sorting path: screens whose targets have random lengths (1 ... 300 guides, so
that a target's partial sums come from one, two or several 64-guide tiles and both layouts of
DevArgs::tsum are hit), random replicate counts and random target-aligned shard cuts.  For every case:
ELBO and gradients against the oracle (float64 mode, 1e-9 / 5e-7), and the shards - each with its global
offsets, none starting on a tile boundary if the dice allow - reproduce the whole-screen fit bit for bit.  -m gpu."""
import os

import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
from oracle import elbo, svi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCALE = int(os.environ.get("BEAN_FUZZ_SCALE", "1"))  # more cases of every kind (a hunt, not the suite)


def _random_screen(seed):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(65, 2600))
    R = int(rng.integers(1, 7))
    lmax = int(rng.choice([2, 9, 40, 64, 65, 130, 300]))
    data = make_sorting_variant_screen(G, R, seed=1000 + seed, guides_per_target=1,
                                       mask_fraction=0.05 if R > 1 else 0.0, with_accessibility=bool(seed % 2))
    lengths = []
    left = G
    while left > 0:
        n = int(min(left, rng.integers(1, lmax + 1)))
        lengths.append(n)
        left -= n
    data.target_lengths = torch.tensor(lengths, dtype=torch.int64)
    data.n_targets = len(lengths)
    return data, rng, lmax


@pytest.mark.parametrize("seed", range(14 * SCALE))
def test_random_target_layouts(seed):
    from bean_amd import engine, parallel

    data, rng, lmax = _random_screen(seed)
    acc = data.guide_accessibility is not None
    kw = dict(scale_by_accessibility=True) if acc else {}
    family = "MixtureNormal" if seed % 3 else "Normal"
    if family == "Normal":
        kw = {}
    # ---- ELBO + gradients against the oracle
    torch.manual_seed(seed)
    eng = engine.HipSVI(family, data.to(DEV), dump_noise=True, num_steps=40, **kw)
    for v in eng.unconstrained.values():
        v.add_(0.3 * torch.randn_like(v))
    loss, grads = eng.elbo_grad(step=1, seed=3)
    noise = {k: v.cpu() for k, v in eng.drawn_noise().items()}
    params = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in eng.unconstrained.items()}
    ref_loss, ref_grads, _ = svi.loss_and_grads(elbo.LOSSES[family], elbo.as_float64(data), params, noise=noise, **kw)
    assert abs(loss - ref_loss) <= 1e-9 * abs(ref_loss), (seed, loss, ref_loss)
    for k, g in grads.items():
        ref = ref_grads[k].double().reshape(-1)
        err = (g.cpu().double().reshape(-1) - ref).abs().max().item()
        assert err <= 5e-7 * (ref.abs().max().item() + 1e-30), (seed, k, err, lmax)
    eng.close()
    # ---- shards reproduce the whole-screen fit bitwise
    steps = 25
    whole = engine.HipSVI(family, data.to(DEV), num_steps=steps, **kw)
    whole.run(steps, seed=9)
    ref_p = whole.constrained()
    ref_l = np.array(whole.losses())
    whole.close()
    n_sh = int(min(data.n_targets, rng.integers(2, 6)))
    shards = parallel.plan_shards(data.target_lengths.numpy(), n_sh)
    parts, losses = [], np.zeros(steps)
    for sh in shards:
        e = engine.HipSVI(family, parallel.shard_screen(data, sh).to(DEV), guide_offset=sh[0], target_offset=sh[2],
                          n_guides_total=data.n_guides, num_steps=steps, **kw)
        e.run(steps, seed=9, resume=True)
        parts.append(e.constrained())
        losses += np.array(e.losses())
        e.close()
    for k in ref_p:
        got = torch.cat([p[k] for p in parts], dim=0)
        assert torch.equal(got, ref_p[k]), (seed, k, lmax, [s[0] for s in shards])
    np.testing.assert_allclose(losses, ref_l, rtol=1e-12)


# ---------------------------------------------------------------- the other families, random shapes
def _parity_module():
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("_gpu_parity_helpers", os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_random_survival_shapes(seed):
    """Survival variant families (the guide kernel parks its cold values in LDS since round 4): random guide and
    replicate counts, 2 ... 9 timepoints, guides per target 1 ... 40, with / without accessibility."""
    from bean_amd import engine
    from bean_amd.preprocessing.synthetic import make_survival_variant_screen

    P = _parity_module()
    rng = np.random.default_rng(100 + seed)
    nt = int(rng.integers(2, 10))
    times = tuple(float(t) for t in np.cumsum(rng.integers(1, 5, nt)) - 1)
    acc = bool(seed % 2)
    family = "Normal" if seed % 4 == 3 else "MixtureNormal"
    R = int(rng.integers(1, 6))
    data = make_survival_variant_screen(int(rng.integers(65, 1500)), R, times=times,
                                        control_index=int(rng.integers(0, nt)), guides_per_target=int(rng.integers(1, 41)),
                                        seed=200 + seed, with_accessibility=acc and family == "MixtureNormal",
                                        mask_fraction=0.05 if R > 1 else 0.0)  # (one replicate: a masked sample would empty a bin)
    kw = dict(scale_by_accessibility=True) if (acc and family == "MixtureNormal") else {}
    P._compare_survival(engine, family, data, kw)


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_random_tiling_shapes(seed):
    """Tiling families, sorting and survival: 2 ... 8 alleles per guide in the register-resident kernels (the
    survival builds form the control-count term per allele since round 4), 17 ... 70 in the allele-parallel ones
    (allele-contiguous tables since round 4), with / without accessibility."""
    from bean_amd import engine
    from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen, make_survival_tiling_screen

    P = _parity_module()
    rng = np.random.default_rng(300 + seed)
    wide = seed % 4 == 3
    A = int(rng.integers(17, 71)) if wide else int(rng.integers(2, 9))
    G, R = int(rng.integers(65, 260 if wide else 700)), int(rng.integers(1, 5))
    acc = bool(seed % 2)
    kw = dict(scale_by_accessibility=True) if acc else {}
    gen = dict(n_max_alleles=A, with_accessibility=acc, mask_fraction=0.05 if R > 1 else 0.0)
    if seed % 3 == 0:
        nt = int(rng.integers(3, 7))
        data = make_survival_tiling_screen(G, R, times=tuple(float(3 * i) for i in range(nt)),
                                           control_index=int(rng.integers(0, nt)), seed=400 + seed, **gen)
        cmp = P._compare_survival_tiling
    else:
        data = make_sorting_tiling_screen(G, R, seed=400 + seed, **gen)
        cmp = P._compare_tiling
    if seed % 2 == 0 or seed % 8 == 7:
        # the order run_inference hands the guides over in (most alleles first): the waves' slot loops then stop at
        # different slots from wave to wave
        from bean_amd import parallel

        data = parallel.order_by_alleles(data)[0]
    cmp(engine, data, kw)

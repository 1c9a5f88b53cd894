"""All the steps of a call in one launch, tile-asynchronous (k_svi_async, csrc/bean_async_v2.hpp; BEAN_HIP_STEP=async /
=pair) against the two launches per step (k_guide_wave2 + k_param): same draws, same per-pair arithmetic, same summation
orders - the fitted parameters must be bit-identical, for every window length, across resumed windows, for shards
with offsets, and whatever the number of resident waves; the loss history agrees to the 2^-40 granule of its fixed-point
parts (the prior / entropy terms of a tile's targets are rounded to that granule per finishing wave, per block of
k_param on the pair path) and is itself reproducible run to run.  -m gpu."""
import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fit(monkeypatch, mode, family, data, steps, eng_kw, chunks=None, resume=False, blocks=None, fin=None):
    from bean_amd import engine

    monkeypatch.setenv("BEAN_HIP_STEP", mode)
    for key, val in (("BEAN_HIP_ASYNC_BLOCKS", blocks), ("BEAN_HIP_ASYNC_FIN", fin)):
        if val is not None:
            monkeypatch.setenv(key, str(val))
        else:
            monkeypatch.delenv(key, raising=False)
    eng = engine.HipSVI(family, data.to(DEV), num_steps=steps, **eng_kw)
    assert eng.dominant_kernel == ("k_guide_wave2" if mode == "pair" else "k_svi_async")
    for n in (chunks or [steps]):
        eng.run(n, seed=5, resume=resume)
    torch.cuda.synchronize()
    out = {k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()}
    loss = np.array(eng.losses())
    eng.close()
    return out, loss


def _same(monkeypatch, family, data, steps, eng_kw=None, chunks=None, resume=False, blocks=None, fin=None):
    a, la = _fit(monkeypatch, "pair", family, data, steps, eng_kw or {}, chunks, resume)
    b, lb = _fit(monkeypatch, "async", family, data, steps, eng_kw or {}, chunks, resume, blocks, fin)
    assert np.all(np.isfinite(la)) and len(la) == steps == len(lb)
    for k in a:
        assert torch.equal(a[k], b[k]), (k, (a[k] - b[k]).abs().max().item())
    assert np.max(np.abs(la - lb) / np.abs(la)) < 1e-12


@pytest.mark.parametrize("n_guides,n_reps,gpt", [(64, 1, 1), (200, 2, 3), (333, 1, 5), (461, 3, 7), (130, 5, 1),
                                                 (1300, 9, 5), (6400, 3, 64), (3000, 4, 33), (4097, 2, 5)])
def test_async_is_bitwise_the_pair_path(monkeypatch, n_guides, n_reps, gpt):
    data = make_sorting_variant_screen(n_guides, n_reps, seed=300 + n_guides, guides_per_target=gpt,
                                       mask_fraction=0.05 if (n_guides > 100 and n_reps > 1) else 0.0)
    _same(monkeypatch, "MixtureNormal", data, 37)


@pytest.mark.parametrize("steps,chunks,resume", [(1, None, False), (2, None, False), (3, None, True), (150, None, False),
                                                 (131, [1, 2, 64, 64], True), (131, [1, 2, 64, 64], False),
                                                 (300, [100, 100, 100], True),
                                                 (8300, None, False)])  # (one call too long for finisher roles: no rings)
def test_async_step_counts_and_windows(monkeypatch, steps, chunks, resume):
    data = make_sorting_variant_screen(2500, 3, seed=77, guides_per_target=5)
    _same(monkeypatch, "MixtureNormal", data, steps, chunks=chunks, resume=resume)


def test_async_families(monkeypatch):
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


@pytest.mark.parametrize("blocks", [8, 64, 1000])
def test_async_does_not_depend_on_the_number_of_resident_waves(monkeypatch, blocks):
    """Fewer workgroups than one step has items: every wave works through several items per step and the polls
    really wait.  Same bits."""
    data = make_sorting_variant_screen(9000, 4, seed=81, guides_per_target=5, mask_fraction=0.03)
    _same(monkeypatch, "MixtureNormal", data, 60, blocks=blocks)


@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("blocks,fin", [(64, 0), (64, 8), (512, 64), (64, -1), (1000, -1)])
def test_async_finisher_roles(monkeypatch, blocks, fin, split):
    """Finishes by the last arriver (fin = 0), by dedicated finisher waves (the default where a SIMD has room for one
    more wave), and - roles on, no finisher resident (fin = -1) - by the waves that wait for them, which take the oldest
    finish nobody has taken: the launch makes progress whatever is resident.  Same bits."""
    monkeypatch.setenv("BEAN_HIP_ASYNC_SPLIT", split)  # a tile's finish as two ring entries (targets, guides) or as one
    data = make_sorting_variant_screen(9000, 4, seed=84, guides_per_target=5, mask_fraction=0.03)
    _same(monkeypatch, "MixtureNormal", data, 50, blocks=blocks, fin=fin)


def test_async_metric_shape_long(monkeypatch):
    """The metric shape over a whole fit of the reference's default length: the hand-overs under full load (every wave slot of the chip in use,
    finishing waves of one step beside the guide waves of the next, rewritten tables read across XCDs)."""
    data = make_sorting_variant_screen(50000, 5, seed=79)
    _same(monkeypatch, "MixtureNormal", data, 2000, chunks=[100] * 20, resume=True)  # a whole `bean run` fit: 2 000 steps


def test_async_config4_shard_under_full_load(monkeypatch):
    """One rank's shard of BASELINE config 4 (62 500 of 500 000 guides, an offset that is no multiple of 64) over three
    report windows: item waves with ~2.4 items per step, finisher roles with the finish as ONE ring entry - the grid this
    shape gets by default - against the pair path."""
    from bean_amd import engine, parallel

    data = make_sorting_variant_screen(500000, 5, seed=85)
    shards = parallel.plan_shards(data.target_lengths.numpy(), 8)
    sh = shards[3]
    assert sh[0] % 64 != 0
    sub = parallel.shard_screen(data, sh).to(DEV)
    out = {}
    for mode in ("pair", "async"):
        monkeypatch.setenv("BEAN_HIP_STEP", mode)
        eng = engine.HipSVI("MixtureNormal", sub, guide_offset=sh[0], target_offset=sh[2], n_guides_total=data.n_guides,
                            num_steps=300)
        assert eng.dominant_kernel == ("k_guide_wave2" if mode == "pair" else "k_svi_async")
        for _ in range(3):
            eng.run(100, seed=11, resume=True)
        torch.cuda.synchronize()
        out[mode] = ({k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()}, np.array(eng.losses()))
        eng.close()
    for k in out["pair"][0]:
        assert torch.equal(out["pair"][0][k], out["async"][0][k]), k
    la, lb = out["pair"][1], out["async"][1]
    assert np.all(np.isfinite(la)) and np.max(np.abs(la - lb) / np.abs(la)) < 1e-12


def test_async_shard_with_an_offset(monkeypatch):
    """A shard that does not start at a multiple of 64 guides (tiles follow the global guide index)."""
    from bean_amd import engine, parallel

    data = make_sorting_variant_screen(20000, 3, seed=82, guides_per_target=5)
    shards = parallel.plan_shards(data.target_lengths.numpy(), 3)
    assert any(sh[0] % 64 for sh in shards)
    outs = {}
    for mode in ("pair", "async"):
        monkeypatch.setenv("BEAN_HIP_STEP", mode)
        res = []
        for sh in shards:
            eng = engine.HipSVI("MixtureNormal", parallel.shard_screen(data, sh).to(DEV), guide_offset=sh[0],
                                target_offset=sh[2], n_guides_total=data.n_guides, num_steps=50)
            assert eng.dominant_kernel == ("k_guide_wave2" if mode == "pair" else "k_svi_async")
            eng.run(50, seed=9)
            torch.cuda.synchronize()
            res.append(({k2: v.detach().cpu().clone() for k2, v in eng.unconstrained.items()}, np.array(eng.losses())))
            eng.close()
        outs[mode] = res
    for (pa, la), (pb, lb) in zip(outs["pair"], outs["async"]):
        for k in pa:
            assert torch.equal(pa[k], pb[k]), k
        assert np.max(np.abs(la - lb) / np.abs(la)) < 1e-12


def test_async_many_conditions(monkeypatch):
    """The 16-condition build's copy of the kernel: 12 and 40 sorting conditions (beyond 64 KB of LDS per wave)."""
    for n_bins, guides in ((11, 1500), (39, 700)):
        edges = np.linspace(0, 1, n_bins + 1)
        bins = tuple((float(edges[i]), float(edges[i + 1])) for i in range(n_bins))
        data = make_sorting_variant_screen(guides, 2, bins=bins, seed=86 + n_bins, guides_per_target=4)
        assert data.n_condits == n_bins + 1
        _same(monkeypatch, "MixtureNormal", data, 25)


def test_async_loss_history_is_reproducible(monkeypatch):
    data = make_sorting_variant_screen(12000, 3, seed=83, guides_per_target=5)
    _, l1 = _fit(monkeypatch, "async", "MixtureNormal", data, 80, {})
    _, l2 = _fit(monkeypatch, "async", "MixtureNormal", data, 80, {}, blocks=512, fin=0)
    assert np.array_equal(l1, l2)


def test_targets_longer_than_a_tile_take_the_pair_path(monkeypatch):
    from bean_amd import engine

    monkeypatch.setenv("BEAN_HIP_STEP", "async")
    data = make_sorting_variant_screen(6500, 2, seed=80, guides_per_target=65)
    eng = engine.HipSVI("MixtureNormal", data.to(DEV), num_steps=10)
    assert eng.dominant_kernel == "k_guide_wave2"
    eng.run(5)
    assert np.all(np.isfinite(eng.losses()))
    eng.close()


SCALE = int(__import__("os").environ.get("BEAN_FUZZ_SCALE", "1"))


@pytest.mark.parametrize("seed", range(10 * SCALE))
def test_async_random_geometries_shards_and_grids(monkeypatch, seed):
    """Random screens (targets of 1 ... 64 guides, 1 ... 6 replicates, +Acc or not, NormalModel every third seed), cut into
    random target-aligned shards; every shard steps through k_svi_async with a random number of item waves, finisher mode
    and finish form - and the shards together must reproduce the WHOLE screen's pair-path fit bit for bit."""
    from bean_amd import engine, parallel

    rng = np.random.default_rng(7000 + seed)
    G = int(rng.integers(200, 5000))
    R = int(rng.integers(1, 7))
    lmax = int(rng.choice([1, 2, 9, 40, 64]))
    data = make_sorting_variant_screen(G, R, seed=2000 + seed, guides_per_target=1,
                                       mask_fraction=0.05 if R > 1 else 0.0, with_accessibility=bool(seed % 2))
    lengths, left = [], G
    while left > 0:
        n = int(min(left, rng.integers(1, lmax + 1)))
        lengths.append(n)
        left -= n
    data.target_lengths = torch.tensor(lengths, dtype=torch.int64)
    data.n_targets = len(lengths)
    family = "Normal" if seed % 3 == 0 else "MixtureNormal"
    kw = dict(scale_by_accessibility=True) if (family == "MixtureNormal" and data.guide_accessibility is not None) else {}
    steps = 30
    monkeypatch.setenv("BEAN_HIP_STEP", "pair")
    whole = engine.HipSVI(family, data.to(DEV), num_steps=steps, **kw)
    assert whole.dominant_kernel == "k_guide_wave2"
    whole.run(steps, seed=9)
    ref_p, ref_l = whole.constrained(), np.array(whole.losses())
    whole.close()
    monkeypatch.setenv("BEAN_HIP_STEP", "async")
    shards = parallel.plan_shards(data.target_lengths.numpy(), int(min(data.n_targets, rng.integers(1, 5))))
    parts, losses = [], np.zeros(steps)
    for sh in shards:
        monkeypatch.setenv("BEAN_HIP_ASYNC_BLOCKS", str(int(rng.choice([8, 16, 64, 256, 2048]))))
        monkeypatch.setenv("BEAN_HIP_ASYNC_FIN", str(int(rng.choice([-1, 0, 8, 64, 1024]))))
        monkeypatch.setenv("BEAN_HIP_ASYNC_SPLIT", str(int(rng.integers(0, 2))))
        e = engine.HipSVI(family, parallel.shard_screen(data, sh).to(DEV), guide_offset=sh[0], target_offset=sh[2],
                          n_guides_total=data.n_guides, num_steps=steps, **kw)
        # (a shard with fewer than 64 targets runs k_param in its one-block-per-target mode: the pair path, also fine)
        assert e.dominant_kernel in ("k_svi_async", "k_guide_wave2")
        for n in (7, 23):
            e.run(n, seed=9, resume=True)
        parts.append(e.constrained())
        losses += np.array(e.losses())
        e.close()
    for k in ref_p:
        got = torch.cat([p[k] for p in parts], dim=0)
        assert torch.equal(got, ref_p[k]), (seed, k)
    assert np.max(np.abs(losses - ref_l) / np.abs(ref_l)) < 1e-12

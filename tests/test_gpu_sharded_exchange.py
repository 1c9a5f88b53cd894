"""Guide-sharded fits of the families in which something is shared across shards
(SURVEY.md section 8e): tiling per-edit parameters, ControlNormal scalars, the survival
MixtureNormal's Dirichlet-over-guides normalisers.  -m gpu.

One GPU, so the shards are separate engines in one process and the "all-reduce" sums their
exchange buffers between the phases of ``bean_hip_sharded_*`` - the same call sequence a rank runs
with ``torch.distributed.all_reduce`` in between (``HipSVI.run_exchanged``).  The sharded fit must
agree with the single-engine fit up to summation order (float64 sums regrouped by shard)."""
import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd import parallel
from bean_amd.preprocessing.synthetic import (make_sorting_tiling_screen, make_sorting_variant_screen,
                                              make_survival_variant_screen)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = 25


@pytest.fixture(scope="module")
def engine():
    from bean_amd import engine as eng

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return eng


def _all_reduce(engines, key):
    bufs = [e.exchange_buffers()[key] for e in engines if key in e.exchange_buffers()]
    if not bufs:
        return
    torch.cuda.synchronize()
    tot = torch.stack(bufs).sum(0)
    for b in bufs:
        b.copy_(tot)
    torch.cuda.synchronize()


def _run_interleaved(engines, n_steps, seed):
    for e in engines:
        e.exchange_buffers()
        e.phase("begin", seed, 0, n_steps)
    for i in range(n_steps):
        for e in engines:
            e.phase("sums")
        _all_reduce(engines, "gsum")
        for e in engines:
            e.phase("guide")
        _all_reduce(engines, "tgrad")
        _all_reduce(engines, "sq")
        for e in engines:
            e.phase("update", 1 if i == n_steps - 1 else 0)
    torch.cuda.synchronize()
    for e in engines:
        e.steps_done = n_steps


def _close(a, b, tol, what):
    err = (a.double() - b.double()).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), (what, err)


# ordered: every shard hands its guides over ordered by allele count, as run_inference's engine factory does
# (parallel.order_by_alleles) - the whole-screen fit in screen order is still what the shards reproduce
@pytest.mark.parametrize("ordered", [False, True])
def test_tiling_guide_shards_share_the_edit_parameters(engine, ordered):
    data = make_sorting_tiling_screen(420, 2, seed=31, n_max_alleles=5)
    whole = engine.HipSVI("MultiMixtureNormal", data.to(DEV), num_steps=200)
    whole.run(N, seed=9)
    ref, ref_loss = whole.constrained(), np.array(whole.losses())
    cuts = [0, 150, 290, 420]  # guides need no target alignment here: the edits are shared
    engines = []
    for k in range(3):
        sub, kw = data[np.arange(cuts[k], cuts[k + 1])], {}
        if ordered:
            sub, ids = parallel.order_by_alleles(sub, cuts[k])
            assert ids is not None
            kw = dict(guide_ids=ids)
        engines.append(engine.HipSVI("MultiMixtureNormal", sub.to(DEV), num_steps=200, guide_offset=cuts[k],
                                     n_guides_total=data.n_guides, loss_owner=(k == 0), **kw))
    _run_interleaved(engines, N, seed=9)
    for name in ("mu_loc", "mu_scale", "sd_loc", "sd_scale"):
        for e in engines:  # replicated parameters stay identical on every shard
            assert torch.equal(e.constrained()[name], engines[0].constrained()[name]), name
        _close(engines[0].constrained()[name], ref[name], 2e-5, name)
    _close(torch.cat([e.constrained()["alpha_pi"] for e in engines]), ref["alpha_pi"], 2e-5, "alpha_pi")
    loss = sum(np.array(e.losses()) for e in engines)
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-6)
    for e in engines + [whole]:
        e.close()


def test_control_normal_shards(engine):
    data = make_sorting_variant_screen(900, 3, seed=32)
    neg = data[data.negctrl_guide_idx]
    whole = engine.HipSVI("ControlNormal", neg.to(DEV), num_steps=200)
    whole.run(N, seed=9)
    ref, ref_loss = whole.constrained(), np.array(whole.losses())
    G = neg.n_guides
    cuts = [0, G // 3, G]
    engines = [engine.HipSVI("ControlNormal", neg[np.arange(cuts[k], cuts[k + 1])].to(DEV), num_steps=200,
                             guide_offset=cuts[k], n_guides_total=G, loss_owner=(k == 0)) for k in range(2)]
    _run_interleaved(engines, N, seed=9)
    for name in ref:
        assert torch.equal(engines[0].constrained()[name], engines[1].constrained()[name])
        _close(engines[0].constrained()[name], ref[name], 2e-5, name)
    np.testing.assert_allclose(sum(np.array(e.losses()) for e in engines), ref_loss, rtol=1e-6)
    for e in engines + [whole]:
        e.close()


def test_survival_mixture_shards_share_the_abundance_normalisers(engine):
    data = make_survival_variant_screen(700, 3, seed=33, frac_effect=0.4)
    whole = engine.HipSVI("MixtureNormal", data.to(DEV), num_steps=200)
    whole.run(N, seed=9)
    ref, ref_loss = whole.constrained(), np.array(whole.losses())
    shards = parallel.plan_shards(data.target_lengths.numpy(), 3)
    t0_totals = (data.X[:, 0, :].float() + 1).sum(-1)
    engines = [engine.HipSVI("MixtureNormal", parallel.shard_screen(data, sh).to(DEV), num_steps=200,
                             guide_offset=sh[0], target_offset=sh[2], n_guides_total=data.n_guides,
                             t0_totals=t0_totals) for sh in shards]
    _run_interleaved(engines, N, seed=9)
    for name in ref:
        _close(torch.cat([e.constrained()[name] for e in engines]), ref[name], 2e-5, name)
    np.testing.assert_allclose(sum(np.array(e.losses()) for e in engines), ref_loss, rtol=1e-6)
    for e in engines + [whole]:
        e.close()


def test_survival_normal_shards_share_draw_normalisers_and_projection(engine):
    """survival NormalModel: the Dirichlet-over-guides draw enters every guide's likelihood, so its
    normalisers AND the projection sums of its pathwise gradient are exchanged."""
    data = make_survival_variant_screen(600, 3, seed=34, frac_effect=0.4)
    whole = engine.HipSVI("Normal", data.to(DEV), num_steps=200)
    whole.run(N, seed=9)
    ref, ref_loss = whole.constrained(), np.array(whole.losses())
    shards = parallel.plan_shards(data.target_lengths.numpy(), 3)
    engines = [engine.HipSVI("Normal", parallel.shard_screen(data, sh).to(DEV), num_steps=200, guide_offset=sh[0],
                             target_offset=sh[2], n_guides_total=data.n_guides) for sh in shards]
    assert sum(len(e.data.negctrl_guide_idx) for e in engines) == len(data.negctrl_guide_idx)
    _run_interleaved(engines, N, seed=9)
    for name in ref:
        _close(torch.cat([e.constrained()[name] for e in engines]), ref[name], 2e-5, name)
    np.testing.assert_allclose(sum(np.array(e.losses()) for e in engines), ref_loss, rtol=1e-6)
    for e in engines + [whole]:
        e.close()

"""bean_hip_svi_resume (HipSVI.run(..., resume=True)): a fit stepped in windows skips the two preparing launches
of every window after the first.  Parameters, moments and the loss history are those of the plain loop, bit for
bit, whatever the window lengths; a call that does not continue the previous one (other first step, other
seed, a bind or an ELBO evaluation in between) takes the full head again.  -m gpu."""
import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing.synthetic import (make_sorting_tiling_screen, make_sorting_variant_screen,
                                              make_survival_variant_screen)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fit(family, data, windows, resume, kw=None, graph_chunk=50):
    from bean_amd import engine

    eng = engine.HipSVI(family, data.to(DEV), num_steps=sum(windows) + 8, **(kw or {}))
    for n in windows:
        eng.run(n, seed=9, resume=resume, graph_chunk=graph_chunk)
    torch.cuda.synchronize()
    out = ({k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()},
           {k: v.detach().cpu().clone() for k, v in eng._m.items()}, np.array(eng.losses()))
    eng.close()
    return out


def _same(a, b):
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), k
        assert torch.equal(a[1][k], b[1][k]), k
    assert np.array_equal(a[2], b[2])


@pytest.mark.parametrize("windows", [[20], [5, 20], [1, 1, 1, 3, 4, 7, 64, 37], [100, 100, 30]])
def test_resumed_windows_equal_the_plain_loop(windows):
    data = make_sorting_variant_screen(3000, 3, seed=41, mask_fraction=0.05)
    plain = _fit("MixtureNormal", data, [sum(windows)], False)
    assert np.isfinite(plain[2]).all() and len(plain[2]) == sum(windows)
    _same(_fit("MixtureNormal", data, windows, True), plain)
    _same(_fit("MixtureNormal", data, windows, False), plain)
    _same(_fit("MixtureNormal", data, windows, True, graph_chunk=0), plain)  # eager launches


@pytest.mark.parametrize("family,make,kw", [
    ("MixtureNormal", lambda: make_sorting_variant_screen(1500, 2, seed=42, with_accessibility=True), dict(scale_by_accessibility=True)),
    ("Normal", lambda: make_sorting_variant_screen(1200, 3, seed=43), {}),
    ("MultiMixtureNormal", lambda: make_sorting_tiling_screen(500, 3, seed=44, n_max_alleles=6), {}),
    ("MixtureNormal", lambda: make_survival_variant_screen(1500, 3, seed=45), {}),
    ("Normal", lambda: make_survival_variant_screen(900, 2, seed=46), {}),
])
def test_every_family_resumes(family, make, kw):
    data = make()
    _same(_fit(family, data, [6, 13, 30], True, kw), _fit(family, data, [49], False, kw))


def test_a_broken_chain_takes_the_full_head_again():
    from bean_amd import engine

    data = make_sorting_variant_screen(2000, 3, seed=47)
    ref = engine.HipSVI("MixtureNormal", data.to(DEV), num_steps=64)
    ref.run(10, seed=9)
    ref.run(10, seed=9)
    ref.run(12, seed=9, first_step=20)
    eng = engine.HipSVI("MixtureNormal", data.to(DEV), num_steps=64)
    eng.run(10, seed=9, resume=True)
    loss, _ = eng.elbo_grad(step=3, seed=9, loss_index=60)  # another entry point: the prepared step is gone
    assert np.isfinite(loss)
    eng.run(10, seed=9, resume=True, first_step=10)
    eng.run(12, seed=9, resume=True)                         # continues: resumed
    torch.cuda.synchronize()
    for k in ref.unconstrained:
        assert torch.equal(ref.unconstrained[k], eng.unconstrained[k]), k
    assert ref.losses()[:32] == eng.losses()[:32]
    # another seed from the same step on: both sides redo the draws
    ref.run(8, seed=10, first_step=32)
    eng.run(8, seed=10, resume=True, first_step=32)
    torch.cuda.synchronize()
    for k in ref.unconstrained:
        assert torch.equal(ref.unconstrained[k], eng.unconstrained[k]), k
    ref.close()
    eng.close()


def test_alternating_run_and_resume_with_two_seeds_never_replays_the_other_seeds_graphs():
    """The graph families share one recorded seed (the captured launches hold it by value): run(A), resume(B),
    run(B) used to replay run's graphs with A baked in.  Every call is compared with a FRESH engine stepped
    through the same entry point with the same seed from the same parameters."""
    from bean_amd import engine

    data = make_sorting_variant_screen(1500, 3, seed=48).to(DEV)

    def fresh_copy(src):
        e = engine.HipSVI("MixtureNormal", data, num_steps=200)
        for d_src, d_dst in ((src.unconstrained, e.unconstrained), (src._m, e._m), (src._v, e._v)):
            for k in d_src:
                d_dst[k].copy_(d_src[k])
        return e

    eng = engine.HipSVI("MixtureNormal", data, num_steps=200)
    first = 0
    for seed, resume in [(11, False), (12, True), (12, False), (11, True), (12, True), (11, False), (11, True)]:
        ref = fresh_copy(eng)
        ref.run(16, seed=seed, first_step=first, resume=resume)
        eng.run(16, seed=seed, first_step=first, resume=resume)
        torch.cuda.synchronize()
        for k in ref.unconstrained:
            assert torch.equal(ref.unconstrained[k], eng.unconstrained[k]), (seed, resume, k)
        assert ref.losses()[first:first + 16] == eng.losses()[first:first + 16], (seed, resume)
        ref.close()
        first += 16
    eng.close()


def test_a_torch_write_to_the_parameters_between_windows_breaks_the_chain():
    """run(resume=True) after the caller has written a parameter tensor: the draw and tables left on the device
    belong to the old values; the engine sees the tensor's version counter and steps through the plain loop."""
    from bean_amd import engine

    data = make_sorting_variant_screen(1500, 3, seed=49).to(DEV)
    a = engine.HipSVI("MixtureNormal", data, num_steps=64)
    b = engine.HipSVI("MixtureNormal", data, num_steps=64)
    a.run(10, seed=9, resume=True)
    b.run(10, seed=9)
    for e in (a, b):
        e.unconstrained["mu_loc"].mul_(0.5)  # warm start / clamp
    a.run(10, seed=9, resume=True)
    b.run(10, seed=9)
    a.run(10, seed=9, resume=True)
    b.run(10, seed=9)
    torch.cuda.synchronize()
    for k in a.unconstrained:
        assert torch.equal(a.unconstrained[k], b.unconstrained[k]), k
    assert a.losses() == b.losses()
    a.close()
    b.close()

"""The allele blocks of k_param<..., 3> (csrc/bean_kernels.hpp, round 5): k_allele's work - the Phi tables of every
allele slot of the step a launch prepares (reference: bean/model/model.py:618-622, utils.py:10-104) - as the tail
of k_param's grid instead of a launch of its own.  Same arithmetic, so every path must give the bits of the split
form (BEAN_HIP_ALLELE=split, read when an engine is created): pairs in graphs, eager launches, resumed windows,
one ELBO evaluation, accessibility, the 16-allele build.  -m gpu."""
import os

import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _mode:
    def __init__(self, split):
        self.split = split

    def __enter__(self):
        self.old = os.environ.get("BEAN_HIP_ALLELE")
        if self.split:
            os.environ["BEAN_HIP_ALLELE"] = "split"
        else:
            os.environ.pop("BEAN_HIP_ALLELE", None)

    def __exit__(self, *exc):
        if self.old is None:
            os.environ.pop("BEAN_HIP_ALLELE", None)
        else:
            os.environ["BEAN_HIP_ALLELE"] = self.old


def _fit(data, windows, split, resume=False, graph_chunk=50, kw=None):
    from bean_amd import engine

    with _mode(split):
        eng = engine.HipSVI("MultiMixtureNormal", data.to(DEV), num_steps=sum(windows) + 8, **(kw or {}))
    for n in windows:
        eng.run(n, seed=5, resume=resume, graph_chunk=graph_chunk)
    torch.cuda.synchronize()
    out = ({k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()},
           {k: v.detach().cpu().clone() for k, v in eng._m.items()}, np.array(eng.losses()))
    eng.close()
    return out


def _same(a, b):
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), k
        assert torch.equal(a[1][k], b[1][k]), k
    assert np.isfinite(a[2]).all()
    assert np.array_equal(a[2], b[2])


@pytest.mark.parametrize("windows,resume,graph_chunk", [
    ([70], False, 50),          # graphs of {k_param, guide} pairs
    ([23], False, 0),           # eager launches
    ([6, 13, 64, 30], True, 50),  # resumed windows: graphs of {guide, k_param} pairs that begin with a guide launch
    ([6, 13, 30], True, 0),
    ([5, 9], False, 50),        # windows that each take the full head again
])
def test_allele_blocks_give_the_bits_of_k_allele(windows, resume, graph_chunk):
    data = make_sorting_tiling_screen(6000, 3, seed=61, mask_fraction=0.03)
    _same(_fit(data, windows, False, resume, graph_chunk), _fit(data, windows, True, resume, graph_chunk))


def test_allele_blocks_with_accessibility_and_many_alleles():
    acc = make_sorting_tiling_screen(3000, 2, seed=62, with_accessibility=True)
    kw = dict(scale_by_accessibility=True)
    _same(_fit(acc, [40], False, kw=kw), _fit(acc, [40], True, kw=kw))
    wide = make_sorting_tiling_screen(1500, 2, seed=63, n_max_alleles=14, alleles_mean=6.0)  # the 16-allele build
    _same(_fit(wide, [30], False), _fit(wide, [30], True))


def test_one_elbo_evaluation_and_a_fit_behind_it():
    from bean_amd import engine

    data = make_sorting_tiling_screen(2500, 3, seed=64).to(DEV)
    res = []
    for split in (False, True):
        with _mode(split):
            eng = engine.HipSVI("MultiMixtureNormal", data, num_steps=64)
        loss0, g0 = eng.elbo_grad(step=3, seed=5, loss_index=60)
        eng.run(10, seed=5, resume=True)
        loss1, g1 = eng.elbo_grad(step=4, seed=5, loss_index=61)  # another entry point between two windows
        eng.run(10, seed=5, resume=True)
        torch.cuda.synchronize()
        res.append((loss0, loss1, {k: v.detach().cpu().clone() for k, v in g0.items()},
                    {k: v.detach().cpu().clone() for k, v in g1.items()},
                    {k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()}, list(eng.losses())))
        eng.close()
    a, b = res
    assert a[0] == b[0] and a[1] == b[1] and np.isfinite(a[0]) and np.isfinite(a[1])
    for i in (2, 3, 4):
        for k in a[i]:
            assert torch.equal(a[i][k], b[i][k]), (i, k)
    assert a[5] == b[5]

"""A fit whose loss turns non-finite halts at the end of that report window (the reference stops at the
failing step, bean/model/run.py:375-390), dumps the parameters to tmp_result.pkl and raises the reference's
ValueError - it does not run the remaining steps first.  -m gpu."""
import os
import pickle
from functools import partial

import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

pytestmark = pytest.mark.gpu


def test_non_finite_loss_halts_at_its_report_window(tmp_path, monkeypatch):
    from bean_amd import engine
    from bean_amd.model import model as m
    from bean_amd.model.run import run_inference

    data = make_sorting_variant_screen(2000, 3, seed=4)
    data.a0 = data.a0.clone()
    data.a0[17] = float("nan")  # one guide's Dirichlet-Multinomial concentration: the loss is NaN from step 0
    calls = []
    real_run = engine.HipSVI.run

    def counting_run(self, n, *a, **k):
        calls.append(n)
        return real_run(self, n, *a, **k)

    monkeypatch.setattr(engine.HipSVI, "run", counting_run)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(ValueError, match=r"(?s)Fitting halted.*non-finite loss at iteration 0"):
        run_inference(partial(m.MixtureNormalModel), partial(m.MixtureNormalGuide), data, num_steps=1000, verbose=False)
    assert calls == [100]  # one report window, not ten
    with open(tmp_path / "tmp_result.pkl", "rb") as fh:
        dump = pickle.load(fh)
    assert "mu_loc" in dump["param"]
    # the dump is the parameter store at the start of the failing window (here: the initial values), not what
    # a hundred NaN updates have left of it
    for k, v in dump["param"].items():
        assert torch.isfinite(v).all(), k

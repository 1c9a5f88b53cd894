"""k_param<..., KIND> (csrc/bean_kernels.hpp): the specialised builds state their launch conditions to the
compiler (__builtin_assume) and must change nothing.  The generic build is forced in a child process
(BEAN_HIP_PARAM_KIND=0 is read once per process) and the fitted parameters compared bit for bit.  -m gpu."""
import os
import pickle
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import pickle, sys
sys.path.insert(0, %(root)r)
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import (make_sorting_tiling_screen, make_sorting_variant_screen,
                                              make_survival_variant_screen)
out = {}
cases = {
    "variant": ("MixtureNormal", make_sorting_variant_screen(3000, 3, seed=11, mask_fraction=0.05), {}),
    "variant_acc": ("MixtureNormal", make_sorting_variant_screen(2000, 2, seed=12, with_accessibility=True),
                    dict(scale_by_accessibility=True)),
    "normal": ("Normal", make_sorting_variant_screen(2000, 3, seed=13), {}),
    "survival": ("MixtureNormal", make_survival_variant_screen(3000, 3, seed=14), {}),
    "survival_normal": ("Normal", make_survival_variant_screen(2000, 2, seed=15), {}),
    "tiling": ("MultiMixtureNormal", make_sorting_tiling_screen(1500, 2, seed=16), {}),
}
for name, (family, data, kw) in cases.items():
    eng = engine.HipSVI(family, data.to("cuda:0"), num_steps=40, **kw)
    eng.run(33, seed=9)
    torch.cuda.synchronize()
    out[name] = ({k: v.detach().cpu() for k, v in eng.unconstrained.items()}, eng.losses())
    eng.close()
pickle.dump(out, open(sys.argv[1], "wb"))
"""


def _run(tmp_path, tag, env_extra):
    path = str(tmp_path / f"{tag}.pkl")
    env = dict(os.environ, **env_extra)
    env.pop("BEAN_HIP_STEP", None)
    res = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), path], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    with open(path, "rb") as f:
        return pickle.load(f)


def test_specialised_k_param_builds_change_nothing(tmp_path):
    import torch

    a = _run(tmp_path, "auto", {})
    b = _run(tmp_path, "generic", {"BEAN_HIP_PARAM_KIND": "0"})
    assert a.keys() == b.keys()
    for name in a:
        pa, la = a[name]
        pb, lb = b[name]
        for k in pa:
            assert torch.equal(pa[k], pb[k]), (name, k)
        assert la == lb, name

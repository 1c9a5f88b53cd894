"""k_guide_tiling_rep's launch-level switches change no bit: which slice of the guide order a workgroup takes
(BEAN_HIP_TILING_MAP=0: workgroup b the b-th slice; default: XCD-contiguous runs, one per quarter of the order) and the
issue priority by progress (BEAN_HIP_TILING_PRIO=0: none).  Both are read once per process, so the other setting runs in
a child process and the fitted parameters are compared bit for bit (reference semantics of the kernel:
bean/model/model.py:550-751, 878-962).  -m gpu."""
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
from bean_amd import engine, parallel
from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen
out = {}
cases = {
    "ordered": (make_sorting_tiling_screen(9000, 5, seed=71, mask_fraction=0.02), {}),      # 177 workgroups -> a padded grid of 192
    "few": (make_sorting_tiling_screen(130, 3, seed=72), {}),                                 # 2 workgroups -> 32
    "acc": (make_sorting_tiling_screen(2500, 2, seed=73, with_accessibility=True), dict(scale_by_accessibility=True)),
    "wide16": (make_sorting_tiling_screen(1200, 2, seed=74, n_max_alleles=13, alleles_mean=6.0), {}),
}
for name, (data, kw) in cases.items():
    data, ids = parallel.order_by_alleles(data)
    eng = engine.HipSVI("MultiMixtureNormal", data.to("cuda:0"), num_steps=40, guide_ids=ids, **kw)
    eng.run(30, seed=9)
    torch.cuda.synchronize()
    out[name] = ({k: v.detach().cpu() for k, v in eng.unconstrained.items()}, eng.losses())
    eng.close()
pickle.dump(out, open(sys.argv[1], "wb"))
"""


def _run(tmp_path, tag, env_extra):
    path = str(tmp_path / f"{tag}.pkl")
    env = dict(os.environ, **env_extra)
    for k in ("BEAN_HIP_TILING_MAP", "BEAN_HIP_TILING_PRIO", "BEAN_HIP_TILING_W"):
        if k not in env_extra:
            env.pop(k, None)
    res = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), path], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    with open(path, "rb") as f:
        return pickle.load(f)


@pytest.mark.parametrize("waves", ["auto", "4"])
def test_slice_map_and_priority_change_nothing(tmp_path, waves):
    """waves: the library's choice (screens this small give every SIMD at most one single-wave workgroup: W = 1) and four
    waves per workgroup forced (what BASELINE config 3 runs with) - the same bits, too."""
    import numpy as np
    import torch

    w = {} if waves == "auto" else {"BEAN_HIP_TILING_W": waves}
    a = _run(tmp_path, "default" + waves, dict(w))
    b = _run(tmp_path, "plain" + waves, dict(w, BEAN_HIP_TILING_MAP="0", BEAN_HIP_TILING_PRIO="0"))
    if waves != "auto":
        ref = _run(tmp_path, "auto_ref", {})
        for name in a:
            for k in a[name][0]:
                assert torch.equal(a[name][0][k], ref[name][0][k]), (name, k, "W = 4 against the library's choice")
    assert a.keys() == b.keys()
    for name in a:
        pa, la = a[name]
        pb, lb = b[name]
        assert np.isfinite(la).all(), name
        for k in pa:
            assert torch.equal(pa[k], pb[k]), (name, k)
        assert la == lb, name

"""write_result_table against CSVs produced by the reference's own readwrite.py."""
import contextlib
import importlib.util
import io
import os

import numpy as np
import pandas as pd
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd.model import readwrite

HERE = os.path.join(os.path.dirname(__file__), "golden")
spec = importlib.util.spec_from_file_location("make_rw", os.path.join(HERE, "make_readwrite_golden.py"))


def _cases():
    # reuse the generator's case table / builders without importing the reference
    src = open(os.path.join(HERE, "make_readwrite_golden.py")).read()
    ns = {}
    head = src[src.index("CASES = {"):src.index("def main():")]
    exec("import numpy as np, pandas as pd, torch\n" + head, ns)
    return ns["CASES"], ns["build"]


CASES, build = _cases()
STORE = np.load(os.path.join(HERE, "readwrite_cases.npz"))


@pytest.mark.parametrize("i,name", list(enumerate(CASES)))
def test_tables_match_reference(tmp_path, i, name):
    target, guide, P, neg, kw = build(name, 100 + i)
    for k, v in P.items():  # the stored inputs are what the reference saw
        np.testing.assert_array_equal(v.numpy(), STORE[f"{name}__P__{k}"])
    prefix = str(tmp_path) + "/"
    with contextlib.redirect_stdout(io.StringIO()):
        readwrite.write_result_table(target.copy(), guide.copy(), P, "M", prefix=prefix, **kw)
    for kind in ("element", "sgRNA"):
        got = pd.read_csv(prefix + f"bean_{kind}_result.M.csv")
        want = pd.read_csv(os.path.join(HERE, f"readwrite_{name}_{kind}.csv"))
        assert list(got.columns) == list(want.columns), (kind, got.columns, want.columns)
        assert len(got) == len(want)
        for c in want.columns:
            if want[c].dtype.kind in "fc":
                np.testing.assert_allclose(got[c].values, want[c].values, rtol=1e-10, atol=1e-12, err_msg=f"{kind}.{c}")
            else:
                assert (got[c].values == want[c].values).all(), (kind, c)  # includes the row order


def test_return_result_and_bad_shape(tmp_path):
    target, guide, P, neg, kw = build("plain", 100)
    with contextlib.redirect_stdout(io.StringIO()):
        df = readwrite.write_result_table(target.copy(), guide.copy(), P, "M", prefix=str(tmp_path) + "/",
                                          return_result=True, **kw)
    assert {"mu", "mu_sd", "mu_z", "sd", "mu_adj", "mu_z_adj", "novl_adj", "CI[0.025", "0.975]"} <= set(df.columns)
    assert not os.path.exists(str(tmp_path) + "/bean_element_result.M.csv")
    P["mu_loc"] = torch.zeros((2, 2, 2))
    with pytest.raises(ValueError, match="invalid shape"):
        readwrite.write_result_table(target.copy(), guide.copy(), P, "M", prefix=str(tmp_path) + "/", **kw)

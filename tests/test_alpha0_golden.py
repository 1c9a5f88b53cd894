"""Preprocessing (a0 / pi_a0) against fixtures produced by the reference's own
get_alpha0.py / get_pi_alpha0.py (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

import bean_amd  # noqa: F401
from bean_amd.preprocessing import alpha0

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "alpha0_cases.npz"))
N = int(GOLD["n_cases"])


@pytest.mark.parametrize("i", range(N))
def test_fitted_alpha0_matches_reference(i):
    X, sf, mask = GOLD[f"c{i}_X"], GOLD[f"c{i}_sf"], GOLD[f"c{i}_mask"]
    a0, popt = alpha0.fitted_alpha0(X, sf, mask)
    np.testing.assert_allclose(popt, GOLD[f"c{i}_popt"], rtol=1e-9)
    np.testing.assert_allclose(a0, GOLD[f"c{i}_a0"], rtol=1e-9)
    a0s, _ = alpha0.fitted_alpha0(X, sf, mask, shrink=True)
    np.testing.assert_allclose(a0s, GOLD[f"c{i}_a0_shrunk"], rtol=1e-9)
    pred = alpha0.pred_alpha0(X * 0.8, sf, popt, mask)
    np.testing.assert_allclose(pred, GOLD[f"c{i}_pred"], rtol=1e-6)  # X*0.8 is float32 in the fixture


@pytest.mark.parametrize("i", range(N))
def test_fitted_pi_alpha0_matches_reference(i):
    ac, sfc = GOLD[f"c{i}_ac"], GOLD[f"c{i}_sfc"]
    pa0, popt = alpha0.fitted_pi_alpha0(ac, sfc)
    np.testing.assert_allclose(popt, GOLD[f"c{i}_pi_popt"], rtol=1e-9)
    np.testing.assert_allclose(pa0, GOLD[f"c{i}_pi_a0"], rtol=1e-9)
    pa0s, _ = alpha0.fitted_pi_alpha0(ac, sfc, shrink=True)
    np.testing.assert_allclose(pa0s, GOLD[f"c{i}_pi_a0_shrunk"], rtol=1e-9)
    raw, _ = alpha0.fitted_pi_alpha0(ac, sfc, fit=False)
    np.testing.assert_allclose(raw, GOLD[f"c{i}_pi_a0_raw"], rtol=1e-9, equal_nan=True)
    pred = alpha0.pred_pi_alpha0(ac, sfc, popt)
    np.testing.assert_allclose(pred, GOLD[f"c{i}_pi_pred"], rtol=1e-9)


def test_too_sparse_falls_back_to_published_trend():
    # case 2 has 3 guides: < 5 valid points => fallback (b0, b1) = (-1.510, 0.7861)
    _, popt = alpha0.fitted_alpha0(GOLD["c2_X"], GOLD["c2_sf"], GOLD["c2_mask"])
    assert popt == alpha0.FALLBACK_POPT
    _, ppopt = alpha0.fitted_pi_alpha0(GOLD["c2_ac"], GOLD["c2_sfc"])
    assert tuple(ppopt) == alpha0.FALLBACK_PI_POPT


def test_empty_bin_raises():
    X, sf = GOLD["c0_X"], GOLD["c0_sf"]
    mask = np.ones(X.shape[:2])
    mask[:, 1] = 0
    with pytest.raises(ValueError):
        alpha0.fitted_alpha0(X, sf, mask)

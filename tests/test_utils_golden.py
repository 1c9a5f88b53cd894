"""The oracle's restatements of ``get_std_normal_prob``, ``get_alpha`` and ``_scale_edited_pi`` against
outputs of the reference's OWN functions (``bean/model/utils.py:10-31, 34-76, 79-103``), evaluated
unchanged by ``tests/golden/make_utils_golden.py`` in the build container.  These three are the
pure-torch arithmetic of the ELBO; what stays unpinned after this file is Pyro's assembly of them
(trace / mask / ``DirichletMultinomial`` / ``ClippedAdam``: SURVEY.md Appendix A.6 items 1-7)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import elbo

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = np.load(os.path.join(HERE, "golden", "utils_cases.npz"))


def _t(name):
    return torch.from_numpy(CASES[name])


@pytest.mark.parametrize("i", [0, 1, 2])
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_std_normal_bin_prob_equals_reference_get_std_normal_prob(i, tag):
    uq, lq, mu, sd = _t(f"snp{i}_uq"), _t(f"snp{i}_lq"), _t(f"snp{i}_mu"), _t(f"snp{i}_sd")
    B, (G, A) = uq.numel(), mu.shape
    dt = torch.float32 if tag == "f32" else torch.float64
    kw = {}
    if f"snp{i}_mask" in CASES.files:
        kw["mask"] = _t(f"snp{i}_mask").unsqueeze(0).expand(B, -1, -1)
    got = elbo.std_normal_bin_prob(
        uq[:, None, None].expand(-1, G, A), lq[:, None, None].expand(-1, G, A),
        mu.to(dt).unsqueeze(0).expand(B, -1, -1), sd.to(dt).unsqueeze(0).expand(B, -1, -1), **kw)
    want = _t(f"snp{i}_{tag}_out")
    assert got.dtype == want.dtype == torch.float64
    assert torch.equal(got, want)  # same torch ops on the same values: bitwise
    # open edges (uq == 1.0 / lq == 0.0 by exact equality, utils.py:48-49) and the partition of unity
    if i == 0:
        np.testing.assert_allclose((want[:4].sum(0) + _gap(want, mu, sd, dt)).numpy(), 1.0, atol=1e-6)
        assert torch.all(want[4] == 1.0)  # the bulk pseudo-bin (0, 1)


def _gap(want, mu, sd, dt):
    """Mass of the quantile gap (0.4, 0.6) that case 0's four sort bins leave out."""
    z = torch.distributions.Normal(0.0, 1.0).icdf(torch.tensor([0.4, 0.6], dtype=torch.float64))
    d = torch.distributions.Normal(mu.to(dt), sd.to(dt))
    return d.cdf(z[1]) - d.cdf(z[0])


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_dirmult_concentration_equals_reference_get_alpha(i):
    p, sf, sm, a0 = _t(f"ga{i}_p"), _t(f"ga{i}_sf"), _t(f"ga{i}_mask"), _t(f"ga{i}_a0")
    want = _t(f"ga{i}_out")
    for mask in (sm, sm.double()):  # the reference holds an int mask, this build's ScreenTensors a float one
        got = elbo.dirmult_concentration(p, sf, mask, a0)
        assert got.dtype == want.dtype and torch.equal(got, want)
    assert float(want.min()) >= 1e-5 * (1 - 1e-12)  # floored at epsilon where the sample is masked
    if i in (1, 2):
        R, B = sm.shape
        assert torch.all(want[R - 1, :, 0] == 1e-5) and torch.all(want[0, :, B - 1] == 1e-5)


@pytest.mark.parametrize("i", [0, 1, 2])
def test_accessibility_scaling_equals_reference_scale_edited_pi(i):
    """The scaling half of ``scale_pi_by_accessibility``: the oracle's function continues with the
    noise half (a Pyro site in the reference), so the scaled edited columns are recovered from its
    record of the intermediate."""
    pi_e, acc, want = _t(f"sep{i}_pi"), _t(f"sep{i}_acc"), _t(f"sep{i}_out")
    got = pi_e * torch.exp(torch.tensor(elbo.ACC_B)) * torch.pow(acc, elbo.ACC_A).unsqueeze(-1)
    assert got.dtype == want.dtype and torch.equal(got, want)
    # and through the oracle's own function: with zero noise the logit / sigmoid round trip returns the
    # clamped scaled values
    pi = torch.cat([1 - pi_e.sum(-1, keepdim=True), pi_e], -1)
    full = elbo.scale_pi_by_accessibility(pi, acc, torch.zeros(acc.numel(), dtype=pi.dtype))
    ctrl = 1 - want.sum(-1)
    scaled = want / torch.cat([ctrl.unsqueeze(-1), want], -1).sum(-1).clamp(min=1.0)[..., None]
    np.testing.assert_allclose(full[..., 1:].numpy(), scaled.clamp(1e-3, 1 - 1e-3).numpy(), rtol=5e-6 if pi.dtype == torch.float32 else 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("i", [0, 1, 2])
def test_device_phi_reproduces_reference_bin_probabilities(i):
    """The device Phi (``norm_cdf``, the function k_param's bin-edge lanes call) on the reference's
    inputs: Phi(u_hi) - Phi(u_lo) equals the reference's float64 output to 1e-15."""
    from bean_amd import engine

    uq, lq = _t(f"snp{i}_uq"), _t(f"snp{i}_lq")
    mu, sd = _t(f"snp{i}_mu").double(), _t(f"snp{i}_sd").double()
    want = _t(f"snp{i}_f64_out")
    z_hi, z_lo = engine._quantile_edges(uq, lq)
    if f"snp{i}_mask" in CASES.files:
        mask = _t(f"snp{i}_mask")
        sd = sd + (~mask).long() * 100
    u_hi = (z_hi[:, None, None] - mu[None]) * sd.reciprocal()[None]
    u_lo = (z_lo[:, None, None] - mu[None]) * sd.reciprocal()[None]
    c_hi, _ = engine.test_special(3, u_hi.reshape(-1))
    c_lo, _ = engine.test_special(3, u_lo.reshape(-1))
    got = (c_hi - c_lo).reshape(want.shape).cpu()
    if f"snp{i}_mask" in CASES.files:
        got = torch.where(mask.unsqueeze(0), got, torch.zeros_like(got))
    assert float((got - want).abs().max()) < 1e-15 * math.sqrt(2) + 2e-16

"""Tiling screens reach the engine with their guides ordered by allele count (parallel.order_by_alleles): the
draws are those of the screen order (the random streams are keyed by the screen index, BEAN_BUF_GUIDE_IDS), the
fit is the same fit, and what comes back has the screen's guide order.  -m gpu."""
import numpy as np
import pytest
import torch

import bean_amd  # noqa: F401
from bean_amd import engine, parallel
from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# (12 alleles per guide: the 16-allele build; 24: the 32-allele build; 40: the allele-parallel kernels)
@pytest.mark.parametrize("acc,offset,n_al", [(False, 0, 8), (True, 0, 8), (False, 7000, 8), (False, 0, 12), (True, 300, 24),
                                             (False, 100, 40)])
def test_ordered_engine_draws_and_fits_what_the_screen_order_does(acc, offset, n_al):
    data = make_sorting_tiling_screen(700, 3, seed=5, with_accessibility=acc, n_max_alleles=n_al,
                                      alleles_mean=3.0 if n_al == 8 else n_al / 2.5)
    ordered, ids = parallel.order_by_alleles(data, offset)
    assert ids is not None and sorted(ids.tolist()) == list(range(offset, offset + 700))
    n_al = ordered.allele_mask.sum(1)
    assert bool((n_al[:-1] >= n_al[1:]).all())
    kw = dict(num_steps=40, dump_noise=True, scale_by_accessibility=acc, guide_offset=offset,
              n_guides_total=offset + 700)
    a = engine.HipSVI("MultiMixtureNormal", data.to(DEV), **kw)
    b = engine.HipSVI("MultiMixtureNormal", ordered.to(DEV), guide_ids=ids, **kw)
    la, ga = a.elbo_grad(step=3, seed=17)
    lb, gb = b.elbo_grad(step=3, seed=17)
    da, db = a.drawn_noise(), b.drawn_noise()
    for k in ga:  # per-guide gradients come back in screen order too
        np.testing.assert_allclose(gb[k].cpu().numpy(), ga[k].cpu().numpy(), rtol=1e-4, atol=1e-6 * float(ga[k].abs().max()),
                                   err_msg=k)
    # ... and injected draws are taken in screen order: the ordered engine on the screen-order engine's draws
    b.set_noise({k: v for k, v in da.items()})
    lb2, _ = b.elbo_grad(step=3, seed=17)
    assert abs(lb2 - la) <= 1e-11 * abs(la)
    b.set_noise(None)
    # the very draws, guide for guide: what crosses the engine is in SCREEN order (draws, gradients, parameters)
    assert db["pi"].shape == (3, 1, 700, data.n_max_alleles)  # (R, 1, G, A) as the reference shapes it
    assert torch.equal(db["pi"], da["pi"])
    if acc:
        assert torch.equal(db["eps_noise"], da["eps_noise"])
    assert torch.equal(db["eps_mu"], da["eps_mu"])
    assert abs(la - lb) <= 1e-11 * abs(la)  # per-edit sums run in another order
    a.run(40, seed=17, first_step=0)
    b.run(40, seed=17, first_step=0)
    ha, hb = np.asarray(a.losses()), np.asarray(b.losses())
    np.testing.assert_allclose(hb, ha, rtol=1e-9)
    pa, pb = a.constrained(), b.constrained()
    assert set(pa) == set(pb)
    for k in pa:
        # constrained() of the ordered engine is in SCREEN order
        np.testing.assert_allclose(pb[k].cpu().numpy(), pa[k].cpu().numpy(), rtol=2e-4, atol=1e-7, err_msg=k)
    # masked alleles keep their initial value: an exact check of the order that came back
    masked = ~data.allele_mask
    assert torch.equal(pb["alpha_pi"].cpu()[masked], pa["alpha_pi"].cpu()[masked])
    a.close()
    b.close()


def test_guide_ids_are_validated():
    data = make_sorting_tiling_screen(100, 2, seed=1)
    with pytest.raises(ValueError, match="permutation"):
        engine.HipSVI("MultiMixtureNormal", data.to(DEV), guide_ids=torch.arange(1, 101))
    from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

    with pytest.raises(ValueError, match="tiling"):
        engine.HipSVI("MixtureNormal", make_sorting_variant_screen(100, 2, seed=1).to(DEV), guide_ids=torch.arange(100))


def test_run_inference_orders_tiling_guides_and_returns_screen_order(monkeypatch):
    from functools import partial

    from bean_amd.model import model as m
    from bean_amd.model.run import run_inference

    data = make_sorting_tiling_screen(400, 2, seed=9)
    seen = []
    real = engine.HipSVI.__init__

    def spy(self, *a, **k):
        seen.append(k.get("guide_ids"))
        return real(self, *a, **k)

    monkeypatch.setattr(engine.HipSVI, "__init__", spy)
    _, res = run_inference(partial(m.MultiMixtureNormalModel), partial(m.MultiMixtureNormalGuide), data,
                           num_steps=60, verbose=False)
    monkeypatch.setenv("BEAN_HIP_ORDER_GUIDES", "0")
    _, ref = run_inference(partial(m.MultiMixtureNormalModel), partial(m.MultiMixtureNormalGuide), data,
                           num_steps=60, verbose=False)
    assert seen[0] is not None and seen[1] is None
    np.testing.assert_allclose(res["loss"], ref["loss"], rtol=1e-9)
    for k in ref["params"]:
        np.testing.assert_allclose(res["params"][k].numpy(), ref["params"][k].numpy(), rtol=5e-4, atol=1e-7, err_msg=k)
    masked = ~data.allele_mask
    assert torch.equal(res["params"]["alpha_pi"][masked], ref["params"]["alpha_pi"][masked])

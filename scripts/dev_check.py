"""Ad-hoc GPU check used during development (not part of the test suite)."""
import sys, time, copy
import numpy as np, torch, scipy.special as sp
sys.path.insert(0, ".")
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
from oracle import elbo, svi

torch.manual_seed(0)
dev = "cuda:0"
# ---- special functions
rng = np.random.default_rng(0)
a = np.exp(rng.uniform(np.log(1e-5), np.log(1e4), 200000))
x = np.floor(np.exp(rng.uniform(0, np.log(5e4), 200000))) * (rng.random(200000) > 0.1)
d, dp = engine.test_special(0, a, x)
ref_d = sp.gammaln(a + x) - sp.gammaln(a)
ref_dp = sp.digamma(a + x) - sp.digamma(a)
err_d = np.abs(d.cpu().numpy() - ref_d) / np.maximum(1, np.abs(ref_d))
err_dp = np.abs(dp.cpu().numpy() - ref_dp) / np.maximum(1e-300, np.abs(ref_dp) + 1e-12)
print("lgdiff max rel err", err_d.max(), "digdiff max rel err", err_dp.max())
lg, dg = engine.test_special(1, a)
print("lgamma err", (np.abs(lg.cpu().numpy() - sp.gammaln(a)) / np.maximum(1, np.abs(sp.gammaln(a)))).max(),
      "digamma err", (np.abs(dg.cpu().numpy() - sp.digamma(a)) / np.maximum(1, np.abs(sp.digamma(a)))).max())
# dirichlet grad vs torch
al = np.exp(rng.uniform(np.log(1e-3), np.log(500), 100000)); be = np.exp(rng.uniform(np.log(1e-3), np.log(500), 100000))
xx = rng.beta(al, be).clip(1e-12, 1 - 1e-12)
g, _ = engine.test_special(2, al, xx, al + be)
tg = torch._dirichlet_grad(torch.tensor(xx), torch.tensor(al), torch.tensor(al + be)).numpy()
rel = np.abs(g.cpu().numpy() - tg) / np.maximum(1e-300, np.abs(tg))
print("dirichlet_grad max rel err vs torch", np.nanmax(rel), "nan", np.isnan(g.cpu().numpy()).sum(), np.isnan(tg).sum())
# sampler moments
n = 400000
a1 = np.full(n, 0.7); b1 = np.full(n, 12.3)
seed = np.zeros(n); seed[:1] = np.frombuffer(np.uint64(1234).tobytes(), dtype=np.float64)
p0, p1 = engine.test_special(4, a1, seed, b1)
p0 = p0.cpu().numpy()
print("beta sample mean/var", p0.mean(), p0.var(), "expect", 0.7 / 13.0, 0.7 * 12.3 / (13.0**2 * 14.0))

# ---- ELBO + grads
for fam, kw in [("MixtureNormal", {}), ("Normal", {}), ("ControlNormal", {}), ("MixtureNormal", dict(scale_by_accessibility=True))]:
    acc = kw.get("scale_by_accessibility", False)
    data = make_sorting_variant_screen(3000, 3, with_accessibility=acc, mask_fraction=0.05)
    if fam == "ControlNormal":
        data = data[np.arange(0, 500)]
    eng = engine.HipSVI(fam, data.to(dev), dump_noise=True, num_steps=100, **kw)
    # perturb params so that grads are generic
    for k, v in eng.unconstrained.items():
        v.add_(0.3 * torch.randn_like(v))
    loss, grads = eng.elbo_grad(step=3, seed=7)
    noise = {k: v.cpu() for k, v in eng.drawn_noise().items()}
    for mode in ("f64", "ref"):
        params = {k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()}
        d2 = data
        if mode == "f64":
            params = {k: v.double() for k, v in params.items()}
        params = {k: v.requires_grad_(True) for k, v in params.items()}
        okw = dict(kw)
        l2, g2, rec = svi.loss_and_grads(elbo.LOSSES[fam], d2, params, noise=noise, **okw)
        print(fam, kw, mode, "loss", loss, l2, "rel", abs(loss - l2) / abs(l2))
        for k in grads:
            gg = grads[k].cpu().double().reshape(-1); oo = g2[k].double().reshape(-1)
            den = oo.abs().max().item() + 1e-30
            print("   grad", k, "max abs err / max", ((gg - oo).abs().max().item()) / den, "max", den)
    # injected-noise path must reproduce the same numbers
    eng.set_noise({k: v for k, v in noise.items()})
    loss_b, grads_b = eng.elbo_grad(step=3, seed=7)
    print("   injected-noise loss diff", abs(loss_b - loss), max((grads_b[k] - grads[k]).abs().max().item() for k in grads))
    eng.set_noise(None)
    # fused run
    t = time.time(); eng.run(100, seed=11, graph_chunk=10); ls = eng.losses(); print("   run 100 steps", time.time() - t, ls[0], ls[-1])
    eng.close()

import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_survival_variant_screen
from oracle import survival, svi, elbo
dev = "cuda:0"
torch.manual_seed(0)
cases = [("ControlNormal", {}, dict(n_guides=3000, n_reps=3)),
         ("MixtureNormal", {}, dict(n_guides=700, n_reps=3, mask_fraction=0.05)),
         ("MixtureNormal", dict(scale_by_accessibility=True), dict(n_guides=400, n_reps=2, with_accessibility=True))]
for fam, kw, gen in cases:
    data = make_survival_variant_screen(seed=4, **gen)
    if fam == "ControlNormal":
        data = data[data.negctrl_guide_idx]
    eng = engine.HipSVI(fam, data.to(dev), dump_noise=True, num_steps=100, **kw)
    for k, v in eng.unconstrained.items():
        v.add_(0.3 * torch.randn_like(v))
    loss, grads = eng.elbo_grad(step=3, seed=7)
    noise = {k: v.cpu() for k, v in eng.drawn_noise().items()}
    for mode in ("f64", "ref"):
        params = {k: v.detach().cpu().clone() for k, v in eng.unconstrained.items()}
        d2 = data
        if mode == "f64":
            params = {k: v.double() for k, v in params.items()}
            d2 = elbo.as_float64(data)
        params = {k: v.requires_grad_(True) for k, v in params.items()}
        l2, g2, rec = svi.loss_and_grads(survival.LOSSES[fam], d2, params, noise=noise, **kw)
        print(fam, kw, mode, "loss", loss, l2, "rel", abs(loss - l2) / abs(l2))
        for k in grads:
            gg = grads[k].cpu().double().reshape(-1); oo = g2[k].double().reshape(-1)
            den = oo.abs().max().item() + 1e-30
            print("   grad", k, "max abs err / max", ((gg - oo).abs().max().item()) / den, "max", den)
    t = time.time(); eng.run(100, seed=11, graph_chunk=10); ls = eng.losses(); print("   run 100 steps", time.time() - t, ls[0], ls[-1])
    eng.close()

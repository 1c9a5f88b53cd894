"""2000-step fits of every family at moderate sizes: losses stay finite and improve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

cases = [
    ("sorting Normal", "Normal", syn.make_sorting_variant_screen(5000, 3, seed=1), {}),
    ("sorting MixtureNormal+Acc", "MixtureNormal", syn.make_sorting_variant_screen(5000, 3, seed=2, with_accessibility=True), dict(scale_by_accessibility=True, fit_noise=True)),
    ("sorting tiling+Acc", "MultiMixtureNormal", syn.make_sorting_tiling_screen(3000, 3, seed=3, with_accessibility=True), dict(scale_by_accessibility=True)),
    ("sorting tiling A=14", "MultiMixtureNormal", syn.make_sorting_tiling_screen(1000, 2, seed=4, n_max_alleles=14), {}),
    ("survival Normal", "Normal", syn.make_survival_variant_screen(3000, 3, seed=5), {}),
    ("survival MixtureNormal+Acc", "MixtureNormal", syn.make_survival_variant_screen(5000, 3, seed=6, with_accessibility=True), dict(scale_by_accessibility=True, fit_noise=True)),
    ("survival tiling", "MultiMixtureNormal", syn.make_survival_tiling_screen(2000, 3, seed=7), {}),
]
bad = 0
for name, fam, data, kw in cases:
    eng = engine.HipSVI(fam, data.to("cuda:0"), num_steps=2000, **kw)
    eng.run(2000)
    torch.cuda.synchronize()
    l = np.array(eng.losses())
    ok = np.isfinite(l).all() and l[-100:].mean() < l[:100].mean()
    pars = eng.constrained()
    fin = all(torch.isfinite(v).all().item() for v in pars.values())
    print(f"{name:30s} loss {l[0]:.5g} -> {l[-1]:.5g}  finite={bool(np.isfinite(l).all())} improved={bool(ok)} params_finite={fin}")
    bad += not (ok and fin)
    eng.close()
sys.exit(bad)

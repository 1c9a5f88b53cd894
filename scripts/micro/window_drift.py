"""Per-window step time of a long fit: does the step slow down as the run goes on (clocks under sustained load)?
Diagnostic.    python scripts/micro/window_drift.py [tiling|metric|survival] [windows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bean_amd  # noqa: F401,E402
from bean_amd import engine, parallel  # noqa: E402
from bean_amd.preprocessing import synthetic as syn  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "tiling"
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 30
kw = {}
if cfg == "tiling":
    data, ids = parallel.order_by_alleles(syn.make_sorting_tiling_screen(50_000, 5, seed=20240503))
    fam, kw = "MultiMixtureNormal", dict(guide_ids=ids)
elif cfg == "survival":
    data, fam = syn.make_survival_variant_screen(100_000, 3, seed=20240506), "MixtureNormal"
else:
    data, fam = syn.make_sorting_variant_screen(50_000, 5, seed=20240502), "MixtureNormal"
eng = engine.HipSVI(fam, data.to("cuda:0"), num_steps=100 * nwin + 200, **kw)
eng.run(100, resume=True)
torch.cuda.synchronize()
out = []
t00 = time.perf_counter()
for w in range(nwin):
    t = time.perf_counter()
    eng.run(100, resume=True)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t) * 1e4, 1))
print(cfg, "us/step per 100-step window (synchronised after each):", out)
# and without a synchronisation between the windows
torch.cuda.synchronize()
t = time.perf_counter()
for w in range(nwin):
    eng.run(100, resume=True) if eng.steps_done + 100 <= 100 * nwin + 200 else None
torch.cuda.synchronize()

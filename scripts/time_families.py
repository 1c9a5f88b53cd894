import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
data = make_sorting_variant_screen(50000, 5, seed=20240502).to("cuda:0")
for fam in ("Normal", "MixtureNormal"):
    eng = engine.HipSVI(fam, data, num_steps=3000)
    eng.run(100); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(1000); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(fam, "us/step", dt / 1000 * 1e6)
    eng.close()

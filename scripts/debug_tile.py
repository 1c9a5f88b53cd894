import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
fam = sys.argv[1] if len(sys.argv) > 1 else "MixtureNormal"
data = make_sorting_variant_screen(int(os.environ.get("DBG_G", "3000")), 5, seed=3, guides_per_target=5).to("cuda:0")
def mk(pair):
    if pair: os.environ["BEAN_HIP_STEP"] = "pair"
    else: os.environ.pop("BEAN_HIP_STEP", None)
    return engine.HipSVI(fam, data, num_steps=2000)
for chunks in ((1,), (2,), (1, 1), (3, 2)):
    a, b = mk(False), mk(True)
    for n in chunks:
        a.run(n, seed=13); b.run(n, seed=13)
    torch.cuda.synchronize()
    print("chunks", chunks, a.dominant_kernel, b.dominant_kernel)
    for k in a.unconstrained:
        d = (a.unconstrained[k].double() - b.unconstrained[k].double()).abs()
        print("  ", k, "max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(), "first idx", (d.reshape(-1) > 0).nonzero()[:5].reshape(-1).tolist())
    print("   loss", a.losses(), b.losses())
    a.close(); b.close()

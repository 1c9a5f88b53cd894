import os, sys
sys.path.insert(0, "/root/repo")
os.environ["BEAN_HIP_STEP"] = "fused"
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
acc = len(sys.argv) > 1 and sys.argv[1] == "acc"
data = make_sorting_variant_screen(1800, 3, seed=78, with_accessibility=True, mask_fraction=0.05)
kw = dict(scale_by_accessibility=True) if acc else {}
eng = engine.HipSVI("MixtureNormal", data.to("cuda:0"), num_steps=40, lib_variant="ab", **kw)
print("kernel", eng.dominant_kernel, flush=True)
eng.run(40, seed=5)
torch.cuda.synchronize()
print("ok", eng.losses()[-1], flush=True)

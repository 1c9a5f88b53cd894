"""Two-rank rehearsal of the sharded run_inference on ONE GPU (gloo, both ranks on cuda:0):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/rehearse_sharded.py
Fits a tiling screen and a survival screen sharded over the ranks and compares every rank's result
with the single-engine fit."""
import os
import sys
from functools import partial

sys.path.insert(0, ".")
import numpy as np
import torch
import torch.distributed as dist

import bean_amd  # noqa: F401
from bean_amd import engine
from bean_amd.model import model as sm
from bean_amd.model import survival_model as vm
from bean_amd.model.run import run_inference
from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen, make_survival_variant_screen

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
STEPS = 120
cases = [
    ("tiling", make_sorting_tiling_screen(500, 3, seed=41, n_max_alleles=6), "MultiMixtureNormal",
     partial(sm.MultiMixtureNormalModel), partial(sm.MultiMixtureNormalGuide)),
    ("survival", make_survival_variant_screen(800, 3, seed=42, frac_effect=0.4), "MixtureNormal",
     partial(vm.MixtureNormalModel), partial(vm.MixtureNormalGuide)),
    ("survival-normal", make_survival_variant_screen(600, 3, seed=43, frac_effect=0.4), "Normal",
     partial(vm.NormalModel), vm.NormalGuide),
]
for name, data, family, model, guide in cases:
    store, out = run_inference(model, guide, data, num_steps=STEPS, verbose=False)
    ref = engine.HipSVI(family, data.to("cuda:0"), num_steps=STEPS)
    ref.run(STEPS)
    torch.cuda.synchronize()
    want, want_loss = ref.constrained(), np.array(ref.losses())
    worst = 0.0
    for k, v in want.items():
        got = out["params"][k].reshape(v.shape)
        worst = max(worst, float((got - v.cpu()).abs().max() / max(1.0, float(v.abs().max()))))
    lerr = float(np.max(np.abs(np.array(out["loss"]) - want_loss) / np.abs(want_loss)))
    print(f"rank {rank} {name}: max param err {worst:.2e}, max loss rel err {lerr:.2e}, "
          f"loss {out['loss'][0]:.6g} -> {out['loss'][-1]:.6g}", flush=True)
    assert worst < 1e-4 and lerr < 1e-6, (name, worst, lerr)
    ref.close()
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank} ok")

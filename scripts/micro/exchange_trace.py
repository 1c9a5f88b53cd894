"""Kernel timeline of the exchange families' stepping paths, one rank under `nccl` (run under rocprofv3):

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/xtrace -- python3 $REPO/scripts/micro/exchange_trace.py

XCASE=tiling/8 (default) | survival/8 ...; XMODE=python | native | plain.  60 steps after 20 of warm-up; the trace's
queue ids tell whether the collective runs on the engine's stream or on one of the process group's own.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29593")

import torch
import torch.distributed as dist

import bean_amd  # noqa: F401
from bean_amd import engine, parallel
from bean_amd.preprocessing import synthetic as syn

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
fam_name, frac = os.environ.get("XCASE", "tiling/8").split("/")
frac = int(frac)
mode = os.environ.get("XMODE", "python")
kw = {}
if fam_name == "tiling":
    data = syn.make_sorting_tiling_screen(50000 // frac, 5, seed=20240503)
    data, ids = parallel.order_by_alleles(data)
    if ids is not None:
        kw["guide_ids"] = ids
    family = "MultiMixtureNormal"
else:
    data = syn.make_survival_variant_screen(100000 // frac, 3, seed=20240506)
    kw["t0_totals"] = (data.X[:, 0, :].float() + 1).sum(-1)
    family = "MixtureNormal"
eng = engine.HipSVI(family, data.to(dev), num_steps=200, device=dev, **kw)
eng.exchange_buffers()
if mode == "native":
    assert eng.init_native_comm()
for n in (20, 60):
    if mode == "plain":
        eng.run(n, seed=101)
    else:
        eng.run_exchanged(n, dist.all_reduce)
    torch.cuda.synchronize()
eng.close()
dist.destroy_process_group()

"""Where the Python-driven exchange loop (HipSVI.run_exchanged without a native communicator) spends its HOST time.

    XCASE=tiling/8 python scripts/micro/exchange_host_profile.py   ->  gpurun_out/exchange_host_<case>.json

One rank under `nccl`.  The loop of HipSVI.run_exchanged is restated here with a perf_counter around every call:
the three library phases (ctypes) and the dist.all_reduce in between, 400 steps, the first 40 dropped; per call kind
min / median / p99 / max and the wall time per step, once with the device allowed to run ahead (no sync) and once
with a device synchronisation per step (host cost and device cost no longer overlap: their sum).  The distribution
tells a steady host-bound loop (median call times add up to the step) from stalls (a few calls 100 x the median).
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29591")

import numpy as np
import torch
import torch.distributed as dist

import bean_amd  # noqa: F401
from bean_amd import engine, parallel
from bean_amd.preprocessing import synthetic as syn

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
case = os.environ.get("XCASE", "tiling/8")
fam_name, frac = case.split("/")
frac = int(frac)
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
data = data.to(dev)
N, DROP = 400, 40


def loop(sync_each):
    eng = engine.HipSVI(family, data, num_steps=N + 8, device=dev, **kw)
    x = eng.exchange_buffers()
    lib, h, sp = eng.lib, eng._h, eng._sptr()
    rec = {k: [] for k in ("sums", "ar_gsum", "guide", "ar_tgrad", "update", "step")}
    pc = time.perf_counter
    with eng._on_stream(), torch.cuda.stream(eng.stream):
        lib.bean_hip_sharded_begin(h, 101, 0, N, sp)
        torch.cuda.synchronize()
        for i in range(N):
            t0 = pc()
            lib.bean_hip_sharded_sums(h, sp)
            t1 = pc()
            if "gsum" in x:
                dist.all_reduce(x["gsum"])
            t2 = pc()
            lib.bean_hip_sharded_guide(h, sp)
            t3 = pc()
            if "tgrad" in x:
                dist.all_reduce(x["tgrad"])
            t4 = pc()
            lib.bean_hip_sharded_update(h, 1 if i == N - 1 else 0, sp)
            if sync_each:
                torch.cuda.synchronize()
            t5 = pc()
            for k, v in zip(("sums", "ar_gsum", "guide", "ar_tgrad", "update", "step"),
                            (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0)):
                rec[k].append(v * 1e6)
        torch.cuda.synchronize()
    eng.close()
    out = {}
    for k, v in rec.items():
        a = np.asarray(v[DROP:])
        out[k] = {"min": round(float(a.min()), 2), "median": round(float(np.median(a)), 2),
                  "p99": round(float(np.percentile(a, 99)), 2), "max": round(float(a.max()), 2),
                  "mean": round(float(a.mean()), 2), "n_over_5x_median": int((a > 5 * np.median(a)).sum())}
    first = {k: [round(t, 1) for t in v[:6]] for k, v in rec.items()}
    return out, first


res = {"case": case, "guides": int(data.n_guides)}
# wall time per step of the product call itself, three times in this process
eng = engine.HipSVI(family, data, num_steps=3 * 300 + 64, device=dev, **kw)
eng.exchange_buffers()
eng.run_exchanged(20, dist.all_reduce)
torch.cuda.synchronize()
walls = []
for _ in range(3):
    t = time.perf_counter()
    eng.run_exchanged(300, dist.all_reduce)
    torch.cuda.synchronize()
    walls.append(round((time.perf_counter() - t) / 300 * 1e6, 2))
eng.close()
res["run_exchanged_us_per_step_3x300"] = walls
res["free_running"], res["free_running_first_steps"] = loop(False)
res["synchronised_each_step"], _ = loop(True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/exchange_host_{fam_name}_{frac}.json", "w"), indent=1)
print(json.dumps(res, indent=1))
dist.destroy_process_group()

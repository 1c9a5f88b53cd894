"""Step time of the tile-asynchronous stepper (BEAN_HIP_STEP=async) against the two launches per step (=pair), same box,
one process per (mode, size): python scripts/time_async.py [guides ...]"""
import json
import os
import subprocess
import sys
import time

CHILD = r"""
import json, os, sys, time, torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
G = int(sys.argv[1]); acc = sys.argv[2] == "1"
data = make_sorting_variant_screen(G, 5, seed=7, with_accessibility=acc)
eng = engine.HipSVI("MixtureNormal", data.to("cuda:0"), num_steps=2100, **({"scale_by_accessibility": True} if acc else {}))
eng.run(100, seed=3, resume=True)
torch.cuda.synchronize()
out = []
for w in range(int(os.environ.get("WINDOWS", "8"))):
    t0 = time.perf_counter()
    eng.run(100, seed=3, resume=True)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) * 1e4)
L = eng.losses()
print(json.dumps({"kernel": eng.dominant_kernel, "us_per_step": [round(x, 2) for x in out], "loss_last": L[len(out) * 100 + 99]}))
eng.close()
"""

sizes = [int(x) for x in sys.argv[1:]] or [50000, 62500]
res = {}
for G in sizes:
    for acc in (os.environ.get("ACC", "0"),):
        for mode in ["pair", "async"] + [f"async@{b}" for b in os.environ.get("ASYNC_BLOCKS", "").split(",") if b]:
            env = dict(os.environ, BEAN_HIP_STEP=mode.split("@")[0])
            if "@" in mode:
                env["BEAN_HIP_ASYNC_BLOCKS"] = mode.split("@")[1]
            p = subprocess.run([sys.executable, "-c", CHILD, str(G), acc], env=env, capture_output=True, text=True, timeout=600)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]
            print(G, mode, line, flush=True)
            res[f"{G}_{mode}"] = line
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/time_async.json", "w"), indent=1)

"""Tiling step time with the guides' roles of k_param moved into the allele-table launch (k_allele_guides, the default)
against one k_param + k_allele (BEAN_HIP_TILING_SPLIT=0); one process per case: python scripts/time_tiling_split.py [G ...]"""
import json
import os
import subprocess
import sys

CHILD = r"""
import json, os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bean_amd
from bean_amd import engine, parallel
from bean_amd.preprocessing import synthetic as syn
G = int(sys.argv[1]); acc = sys.argv[2] == "1"
data = syn.make_sorting_tiling_screen(G, 5, seed=20240503, with_accessibility=acc)
data, ids = parallel.order_by_alleles(data)
kw = {"guide_ids": ids} if ids is not None else {}
if acc: kw["scale_by_accessibility"] = True
eng = engine.HipSVI("MultiMixtureNormal", data.to("cuda:0"), num_steps=1000, **kw)
eng.run(100, seed=3, resume=True); torch.cuda.synchronize()
out = []
for w in range(6):
    t0 = time.perf_counter(); eng.run(100, seed=3, resume=True); torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) * 1e4, 2))
print(json.dumps({"us_per_step": out, "loss_last": eng.losses()[699]}))
eng.close()
"""
res = {}
for G in [int(a) for a in sys.argv[1:]] or [50000, 6250]:
    for acc in ("0", "1"):
        for split in ("1", "0"):
            env = dict(os.environ, BEAN_HIP_TILING_SPLIT=split)
            p = subprocess.run([sys.executable, "-c", CHILD, str(G), acc], env=env, capture_output=True, text=True, timeout=600)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]
            print(G, "acc" if acc == "1" else "plain", "split" if split == "1" else "one k_param + k_allele", line, flush=True)
            res[f"{G}_{acc}_{split}"] = line
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/time_tiling_split.json", "w"), indent=1)

"""Spread of k_svi_async's launches: many report windows of one fit, HIP events around every window, no host
synchronisation in between - median / p99 / max us per step and the number of windows beyond 1.3 x the median (a
persistent launch keeps whatever placement the dispatcher gave it for its whole duration: a bad one would show as a
slow WINDOW).   python scripts/micro/async_window_spread.py G n_windows window_steps"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bean_amd  # noqa: F401
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

G, n_win, win = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = syn.make_sorting_variant_screen(G, 5, seed=5).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", d, num_steps=n_win * win + 200)
eng.run(100, resume=True)
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
evs[0].record(eng.stream)
for w in range(n_win):
    eng.run(win, resume=True)
    evs[w + 1].record(eng.stream)
torch.cuda.synchronize()
t = np.array([evs[i].elapsed_time(evs[i + 1]) / win * 1e3 for i in range(n_win)])
out = {"guides": G, "windows": n_win, "steps_per_window": win, "kernel": eng.dominant_kernel,
       "us_per_step_median": round(float(np.median(t)), 2), "p99": round(float(np.percentile(t, 99)), 2),
       "max": round(float(t.max()), 2), "windows_beyond_1.3x_median": int((t > 1.3 * np.median(t)).sum())}
print(json.dumps(out), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/async_window_spread.jsonl", "a") as fh:
    fh.write(json.dumps(out) + "\n")
eng.close()

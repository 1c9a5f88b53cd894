"""Where the host time of a 20-step call goes: enqueue (HipSVI.run returns) vs the wait for the device."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn
data = syn.make_sorting_variant_screen(50000, 5, seed=syn.BASE_SEED + 1).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=5000)
for resume in (False, True, False, True):
    eng.run(5, resume=resume, graph_chunk=64); torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(60):
        torch.cuda.synchronize()
        t = time.perf_counter(); eng.run(20, resume=resume, graph_chunk=64); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append(t1 - t); tot.append(t2 - t)
    enq.sort(); tot.sort()
    print(json.dumps({"resume": resume, "enqueue_us_median": enq[30] * 1e6, "total_us_median": tot[30] * 1e6,
                      "us_per_step": tot[30] / 20 * 1e6}))

"""Per-step time of the variant sorting MixtureNormal fit over screen sizes, tile path vs two-launch path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

def fit(data, steps, pair):
    if pair: os.environ["BEAN_HIP_STEP"] = "pair"
    else: os.environ.pop("BEAN_HIP_STEP", None)
    eng = engine.HipSVI("MixtureNormal", data, num_steps=steps + 200)
    eng.run(100); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
    k = eng.dominant_kernel
    eng.close()
    return round(dt / steps * 1e6, 2), k

for G in [int(a) for a in sys.argv[1:]] or [5000, 25000, 50000, 62500, 100000, 500000]:
    data = syn.make_sorting_variant_screen(G, 5, seed=20240600 + G // 1000).to("cuda:0")
    steps = 400 if G <= 100000 else 100
    a, ka = fit(data, steps, False)
    b, kb = fit(data, steps, True)
    print(json.dumps({"guides": G, ka: a, kb: b, "ratio": round(b / a, 3)}), flush=True)

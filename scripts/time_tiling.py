"""Tiling step / guide-kernel timings over screen sizes and kernel variants (env switches are read by
the library at engine creation): python scripts/time_tiling.py G [G ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

def fit(data, steps=300):
    eng = engine.HipSVI("MultiMixtureNormal", data, num_steps=steps + 200)
    eng.run(50); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
    out = {"us_per_step": round(dt / steps * 1e6, 2), "loss_last": eng.losses()[-1]}
    eng.close()
    for mode, key in ((1, "kernel_us"), (2, "k_param_us")):
        prof = engine.HipSVI("MultiMixtureNormal", data, num_steps=100); prof.set_profile(mode); prof.run(30, graph_chunk=0)
        torch.cuda.synchronize(); ms, n = prof.get_profile(); out[key] = round(ms * 1e3, 2)
        out["kernel"] = prof.dominant_kernel
        prof.close()
    return out

variants = [("rep W=auto", {}), ("rep W=1", {"BEAN_HIP_TILING_W": "1"}), ("rep W=2", {"BEAN_HIP_TILING_W": "2"}),
            ("rep W=4", {"BEAN_HIP_TILING_W": "4"}), ("wave + k_sum_trow", {"BEAN_HIP_TILING": "wave"})]
for G in [int(a) for a in sys.argv[1:]] or [50000]:
    data = syn.make_sorting_tiling_screen(G, 5, seed=20240503).to("cuda:0")
    for name, env in variants:
        for k in ("BEAN_HIP_TILING_W", "BEAN_HIP_TILING"):
            os.environ.pop(k, None)
        os.environ.update(env)
        print(G, name, json.dumps(fit(data)), flush=True)

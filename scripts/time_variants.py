"""Time k_guide / whole step for one library variant (BEAN_HIP_LIB env)."""
import os, sys, time, json
sys.path.insert(0, ".")
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
data = make_sorting_variant_screen(50000, 5, seed=20240502).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=3000)
eng.run(100); torch.cuda.synchronize()
t = time.perf_counter(); eng.run(1000); torch.cuda.synchronize(); dt = time.perf_counter() - t
prof = engine.HipSVI("MixtureNormal", data, num_steps=100, dump_noise=True); prof.set_profile(True); prof.run(50, graph_chunk=0); torch.cuda.synchronize()
ms, n = prof.get_profile()
out = {"lib": os.environ.get("BEAN_HIP_LIB"), "us_per_step": dt / 1000 * 1e6, "k_guide_us": ms * 1e3, "loss_end": eng.losses()[-1]}
if os.environ.get("INJECT_PI"):
    noise = prof.drawn_noise(); prof.set_noise({"pi": noise["pi"]}); prof.run(50, graph_chunk=0, first_step=0); torch.cuda.synchronize()
    ms2, _ = prof.get_profile(); out["k_guide_us_no_sampling"] = ms2 * 1e3
print(json.dumps(out))

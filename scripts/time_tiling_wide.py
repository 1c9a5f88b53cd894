"""Wide tiling path (17 ... 256 alleles per guide: csrc/bean_tiling_wide.hpp) on an unfiltered-style allele
table: python scripts/time_tiling_wide.py [guides] [alleles_mean] -> gpurun_out/tiling_wide.json"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

G = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
mean = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
data = syn.make_sorting_tiling_screen(G, 5, n_max_alleles=232, seed=20240509, alleles_mean=mean)
n_alleles = int(data.allele_mask.sum()) - G
data = data.to("cuda:0")
steps = int(os.environ.get("WIDE_STEPS", "100"))
eng = engine.HipSVI("MultiMixtureNormal", data, num_steps=steps + 100)
eng.run(20); torch.cuda.synchronize()
t = time.perf_counter(); eng.run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
out = {"guides": G, "n_max_alleles": 232, "edited_alleles": n_alleles, "alleles_per_guide_mean": n_alleles / G, "n_edits": int(data.n_edits),
       "us_per_step": round(dt / steps * 1e6, 1), "kernel": eng.dominant_kernel, "loss_first": eng.losses()[0], "loss_last": eng.losses()[-1]}
eng.close()
if not os.environ.get("WIDE_NO_PROFILE"):
    for mode, key in ((1, "kernel_us"), (2, "k_param_us")):
        prof = engine.HipSVI("MultiMixtureNormal", data, num_steps=100); prof.set_profile(mode); prof.run(20, graph_chunk=0)
        torch.cuda.synchronize(); ms, n = prof.get_profile(); out[key] = round(ms * 1e3, 1)
        prof.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/tiling_wide.json", "w"), indent=1)
print(json.dumps(out, indent=1))

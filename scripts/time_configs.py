"""Time one SVI step of BASELINE.json's other configurations (not bench lines): tiling sorting
(config 3), one rank's shard of the 500k-guide variant screen (config 4), survival (config 5).
Writes gpurun_out/configs.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

def fit(family, data, steps, **kw):
    data = data.to("cuda:0")
    eng = engine.HipSVI(family, data, num_steps=steps + 200, **kw)
    eng.run(50); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
    prof = engine.HipSVI(family, data, num_steps=100, **kw); prof.set_profile(True); prof.run(30, graph_chunk=0); torch.cuda.synchronize()
    ms, n = prof.get_profile()
    out = {"us_per_step": dt / steps * 1e6, "dominant_kernel": prof.dominant_kernel, "kernel_us": ms * 1e3,
           "algorithmic_bytes": prof.step_bytes, "GBps": prof.step_bytes / (ms * 1e-3) / 1e9, "loss_first": eng.losses()[0], "loss_last": eng.losses()[-1]}
    eng.close(); prof.close()
    return out

res = {}
which = sys.argv[1:] or ["tiling", "variant500k_shard", "survival"]
if "tiling" in which:
    d = syn.make_sorting_tiling_screen(50000, 5, seed=20240503)
    res["config3 tiling sorting 50k guides x ~200k alleles x (4 bins+bulk) x 5 reps, MultiMixtureNormal"] = dict(
        fit("MultiMixtureNormal", d, 300), n_edits=int(d.n_edits), n_alleles=int(d.allele_mask.sum()) - 50000)
if "variant500k_shard" in which:
    d = syn.make_sorting_variant_screen(62500, 5, seed=20240504)
    res["config4 one of 8 shards (62.5k guides) of the 500k-guide variant sorting screen, MixtureNormal"] = fit("MixtureNormal", d, 1000)
    d = syn.make_sorting_variant_screen(500000, 5, seed=20240505)
    res["config4 whole 500k-guide variant sorting screen on ONE GPU, MixtureNormal"] = fit("MixtureNormal", d, 200)
if "survival" in which:
    d = syn.make_survival_variant_screen(100000, 3, seed=20240506)
    res["config5 survival 100k guides x 6 timepoints x 3 reps, MixtureNormal"] = fit("MixtureNormal", d, 500)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/configs.json", "w"), indent=1)
print(json.dumps(res, indent=1))

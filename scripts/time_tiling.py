import sys, time, json
sys.path.insert(0, ".")
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_tiling_screen
t=time.time(); data = make_sorting_tiling_screen(50000, 5, seed=20240503); print("gen", time.time()-t, "E", data.n_edits, "nnz", data.a2e_idx.numel(), "alleles", int(data.allele_mask.sum()) - 50000)
data = data.to("cuda:0")
eng = engine.HipSVI("MultiMixtureNormal", data, num_steps=3000)
eng.run(50); torch.cuda.synchronize()
t = time.perf_counter(); eng.run(500); torch.cuda.synchronize(); dt = time.perf_counter() - t
prof = engine.HipSVI("MultiMixtureNormal", data, num_steps=100); prof.set_profile(True); prof.run(30, graph_chunk=0); torch.cuda.synchronize()
ms, n = prof.get_profile()
print(json.dumps({"us_per_step": dt / 500 * 1e6, "k_guide_tiling_us": ms * 1e3, "bytes": prof.step_bytes, "loss": eng.losses()[-1]}))

import os, sys, time, json
sys.path.insert(0, "/root/repo")
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn
data = syn.make_survival_variant_screen(100000, 3, seed=20240506).to("cuda:0")
def kp(inject):
    prof = engine.HipSVI("MixtureNormal", data, num_steps=100)
    if inject:
        x0 = torch.full((3, 100000), 1e-5, dtype=torch.float64)
        prof.set_noise({"initial_abundance": x0})
    prof.set_profile(2); prof.run(30, graph_chunk=0); torch.cuda.synchronize(); ms, n = prof.get_profile(); prof.close()
    return round(ms * 1e3, 2)
print(json.dumps({"k_param_us_with_sampler": kp(False), "k_param_us_draws_injected": kp(True)}))

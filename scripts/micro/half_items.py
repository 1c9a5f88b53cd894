"""Would finer work items help the guide kernel?  A wave with ONE likelihood (use_bcmatch=False) is roughly half
a wave's work; twice as many of them (100k guides) is the same likelihood work in items of half the size (the
pi-site work is then done twice: a pessimistic stand-in for a lane-pair form that splits it)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

def kernel_us(G, bc, family="MixtureNormal"):
    d = syn.make_sorting_variant_screen(G, 5, seed=20240600 + G // 1000).to("cuda:0")
    out = {}
    for mode, key in ((1, "guide_us"), (2, "k_param_us")):
        prof = engine.HipSVI(family, d, num_steps=100, use_bcmatch=bc); prof.set_profile(mode); prof.run(40, graph_chunk=0)
        torch.cuda.synchronize(); ms, n = prof.get_profile(); out[key] = round(ms * 1e3, 2); prof.close()
    eng = engine.HipSVI(family, d, num_steps=700, use_bcmatch=bc)
    eng.run(100); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(500); torch.cuda.synchronize(); out["step_us"] = round((time.perf_counter() - t) / 500 * 1e6, 2)
    eng.close()
    return out

res = {}
for G, bc in ((50000, True), (50000, False), (100000, False), (62500, True), (125000, False), (25000, True), (25000, False)):
    res[f"{G} guides, use_bcmatch={bc}"] = kernel_us(G, bc)
    print(G, bc, res[f"{G} guides, use_bcmatch={bc}"], flush=True)

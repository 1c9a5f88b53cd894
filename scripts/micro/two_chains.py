"""Experiment: does one GPU finish a 50k-guide fit sooner as K independent target-aligned shards whose
launch chains (k_param -> guide kernel -> k_param ...) run concurrently on K streams?  (k_param is
latency-bound and the guide kernel has a ramp and a tail: another shard's kernels can fill both.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bean_amd
from bean_amd import engine, parallel
from bean_amd.preprocessing import synthetic as syn

G = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
KS = [int(k) for k in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4]
steps = 1024
d = syn.make_sorting_variant_screen(G, 5, seed=syn.BASE_SEED + 1)
res = {}
for K in KS:
    shards = parallel.plan_shards(d.target_lengths.numpy(), K)
    engs, streams = [], []
    for sh in shards:
        sc = parallel.shard_screen(d, sh).to("cuda:0")
        engs.append(engine.HipSVI("MixtureNormal", sc, num_steps=2 * steps + 200, guide_offset=sh[0],
                                  target_offset=sh[2], n_guides_total=d.n_guides))
        streams.append(torch.cuda.Stream())
    def go(n):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.run(n)
    go(128); torch.cuda.synchronize()
    t = time.perf_counter(); go(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
    loss = sum(e.losses()[-1] for e in engs)
    res[K] = {"us_per_step": dt / steps * 1e6, "loss_last": loss}
    print(K, res[K], flush=True)
    for e in engs: e.close()
os.makedirs("gpurun_out", exist_ok=True)


"""Cost model of k_guide_wave2's pieces in isolation (diagnostic): time the device special functions
and the sampler through bean_hip_test_special on as many elements as one launch of the guide kernel
evaluates at the metric shape (250k (replicate, guide) pairs), with realistic arguments."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bean_amd
from bean_amd import engine

def timed(op, a, x=None, b=None, reps=20):
    dev = "cuda:0"
    a = torch.as_tensor(a, dtype=torch.float64, device=dev)
    x = None if x is None else torch.as_tensor(x, dtype=torch.float64, device=dev)
    b = None if b is None else torch.as_tensor(b, dtype=torch.float64, device=dev)
    engine.test_special(op, a, x, b)  # warm
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); engine.test_special(op, a, x, b); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))

rng = np.random.default_rng(0)
N = 250_000
out = {}
# lgamma/digamma differences: 10 per (rep, guide) with alpha_b ~ 3 (sort bins) / 16 (bulk), counts ~ 100-600; 2 totals
al = np.concatenate([rng.gamma(4.0, 0.8, 8 * N), rng.gamma(16.0, 1.0, 2 * N)])
xx = np.floor(np.concatenate([rng.gamma(2.0, 60.0, 8 * N), rng.gamma(4.0, 150.0, 2 * N)]))
out["lgamma_diff bins x10 per pair (2.5M calls)"] = timed(0, al, xx)
out["lgamma_diff totals x2 per pair (0.5M calls)"] = timed(0, rng.gamma(30.0, 1.0, 2 * N), np.floor(rng.gamma(4.0, 300.0, 2 * N)))
out["lgamma_diff, alpha >= 6 only (2.5M calls)"] = timed(0, 6.0 + rng.gamma(4.0, 2.0, 10 * N), xx)
# sampler: one pair of gammas per (rep, guide), concentrations like pi_a0 / 2
seed = np.zeros(N); seed[:1] = np.frombuffer(np.uint64(99).tobytes(), dtype=np.float64)
c0 = rng.gamma(3.0, 0.4, N) + 0.05
out["gamma pair sampler (250k pairs, conc ~1.2)"] = timed(4, c0, seed, c0[::-1].copy())
out["gamma pair sampler (250k pairs, conc >= 1.5)"] = timed(4, c0 + 1.5, seed, c0[::-1].copy() + 1.5)
# implicit-reparameterisation gradient: two per pair
a = np.concatenate([c0, c0[::-1]]); tot = np.concatenate([c0 + c0[::-1]] * 2)
x = rng.beta(a, tot - a).clip(1e-9, 1 - 1e-9)
out["dirichlet_grad_one x2 per pair (0.5M calls)"] = timed(2, a, x, tot)
out["lgamma_digamma single (0.5M calls)"] = timed(1, a)
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/special_cost.json", "w"), indent=1)

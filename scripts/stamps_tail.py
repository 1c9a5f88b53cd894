"""Diagnostic (-DBEAN_STAMP=3 build only): cycle stamps of the tail of k_step_wave2 (one record per tile)."""
import sys, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import bean_amd
from bean_amd import engine, _lib
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
G = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
data = make_sorting_variant_screen(G, 5, seed=20240502).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=100)
eng.run(20, graph_chunk=0); torch.cuda.synchronize()
n = (G + 63) // 64
buf = np.zeros(n * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(n, 8).astype(np.int64)
s = s[(s[:, 0] > 0) & (s[:, 5] > s[:, 0])]
names = {1: "ownership (counters, offsets)", 2: "phases A + B (row sums, priors, Adam, draw)", 3: "phase C (Phi tables)",
         4: "guide part (alpha_pi, tables of lgamma / digamma)", 5: "loss parts + atomics"}
print("tiles", len(s))
for k in range(1, 6):
    d = s[:, k] - s[:, k - 1]
    print(f"  {k} {names[k]:52s} median {np.median(d):8.0f}  p10 {np.percentile(d,10):8.0f} p90 {np.percentile(d,90):8.0f}")
tot = s[:, 5] - s[:, 0]
print("tail median", np.median(tot), "p90", np.percentile(tot, 90))

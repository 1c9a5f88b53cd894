"""Diagnostic (-DBEAN_STAMP=1 build only): per-segment cycle stamps of k_guide_wave2 at the metric shape.
   BEAN_HIP_LIB=build/variants/libbean_hip_stamp.so python scripts/stamps.py"""
import os, sys, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import bean_amd
from bean_amd import engine, _lib
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
G = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
data = make_sorting_variant_screen(G, 5, seed=20240502).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=100)
eng.run(20, graph_chunk=0); torch.cuda.synchronize()
n_waves = 5 * (((G + 63) // 64 + 7) // 8 * 8)
buf = np.zeros(n_waves * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(n_waves, 8).astype(np.int64)
s = s[(s[:, 0] != 0) & (s[:, 7] > s[:, 0])]  # padded wave ids of the XCD-aware grid exit at once
names = {1: "loads + LDS staging + pi draw (last sampler round)", 2: "accessibility/weights + pass 1", 3: "lik X: loop 1 (lgamma differences)",
         4: "lik X: total + loop 2", 5: "(between likelihoods)", 6: "lik X_bcmatch (both loops)", 7: "pi terms + implicit gradients + rows + loss"}
print("waves", len(s), " median cycles per segment (stamp k - stamp k-1; s_memtime ticks):")
prev = 0
for k in range(1, 8):
    ok = (s[:, k] > 0) & (s[:, prev] > 0)
    d = s[ok, k] - s[ok, prev]
    print(f"  {k} {names[k]:52s} median {np.median(d):8.0f}  p10 {np.percentile(d,10):8.0f} p90 {np.percentile(d,90):8.0f}")
    prev = k
tot = s[:, 7] - s[:, 0]
print("wave lifetime median", np.median(tot), "p10", np.percentile(tot, 10), "p90", np.percentile(tot, 90))

import os, sys, ctypes
sys.path.insert(0, ".")
import numpy as np, torch
import bean_amd
from bean_amd import engine, _lib
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
data = make_sorting_variant_screen(50000, 5, seed=20240502).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=100)
eng.run(20, graph_chunk=0); torch.cuda.synchronize()
WAVE = os.environ.get("BEAN_HIP_GUIDE") != "split"
n_waves = (1 if WAVE else 2) * 5 * (((50000 + 63) // 64 + 7) // 8 * 8)
buf = np.zeros(n_waves * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(n_waves, 8).astype(np.int64)
s = s[s[:, 0] != 0]  # padded wave ids of the XCD-aware grid exit at once
d = np.diff(s, axis=1)
if WAVE:
    names = ["setup loads", "pi draw + pass 1", "lik X loop 1 (lgamma)", "lik X total + loop 2", "(gap)", "lik X_bcmatch", "pi terms + dirichlet grads + rows"]
else:
  names = ["prologue loads+LDS store", "barrier", "pass1 (+pi loads)", "pass2", "d0+final math", "exchange+writes", "block_sum"]
print("median cycles per segment:")
for i, n in enumerate(names):
    print(f"  {n:28s} median {np.median(d[:, i]):9.0f}  mean {d[:, i].mean():9.0f}")
print("total median", np.median(s[:, 7] - s[:, 0]), "kernel span cycles", s[:, 7].max() - s[:, 0].min())
t0 = s[:, 0].min()
for nm, col in (("start", 0), ("end", 7)):
    v = s[:, col] - t0
    print(nm, "percentiles 0/10/50/90/100:", [int(np.percentile(v, q)) for q in (0, 10, 50, 90, 100)])

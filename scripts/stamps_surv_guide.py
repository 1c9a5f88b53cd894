"""Diagnostic (-DBEAN_STAMP=5 build, BEAN_HIP_LIB=<that build>): start and end of every wave of
k_guide_survival_wave on the 100 MHz real-time clock (BASELINE config 5), i.e. how much of the launch is a
full machine and how much is the drain of its last waves."""
import ctypes
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import bean_amd  # noqa: F401
from bean_amd import _lib, engine
from bean_amd.preprocessing import synthetic as syn

G, R = 100000, 3
data = syn.make_survival_variant_screen(G, R, seed=20240506).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=100)
eng.run(20, graph_chunk=0)
torch.cuda.synchronize()
nw = ((G + 63) // 64 + 7) // 8 * 8 * R
buf = np.zeros(nw * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(nw, 8).astype(np.int64)
s = s[(s[:, 5] > 0) & (s[:, 6] > 0)]  # slots 5 / 6: start / end of a wave
t0 = s[:, 5].min()
st, en = (s[:, 5] - t0) / 100.0, (s[:, 6] - t0) / 100.0
print(f"{len(s)} waves; kernel span {en.max():.1f} us")
for name, v in (("start", st), ("end", en), ("duration", en - st)):
    q = np.percentile(v, [0, 10, 50, 90, 99, 100])
    print(f"  {name:9s} min {q[0]:6.1f} p10 {q[1]:6.1f} median {q[2]:6.1f} p90 {q[3]:6.1f} p99 {q[4]:6.1f} max {q[5]:6.1f}")
grid = np.arange(0, en.max() + 2, 2.0)
res = [(int(((st <= t) & (en > t)).sum())) for t in grid]
print("  resident waves every 2 us:", res)
late = st > 5.0
print(f"  waves started after 5 us: {int(late.sum())}; their duration median {np.median((en - st)[late]):.1f} us; "
      f"first-round duration median {np.median((en - st)[~late]):.1f} us")

"""Diagnostic (-DBEAN_STAMP=6 build only, BEAN_HIP_LIB=<that build>): where the waves of k_guide_tiling_rep spend their
time at BASELINE config 3 - eight stamps per wave on the 100 MHz real-time clock: 0 start, 1 first loads +
concentrations, 2 draw, 3 forward mix, 4 likelihoods, 5 backward loop + rows, 6 Multinomial + Dirichlet terms +
implicit gradients, 7 end.  Waves are grouped by the quarter of the allele-count order their workgroup's slice lies in.

    BEAN_HIP_LIB=build/variants/libbean_hip_stamp6.so python scripts/stamps_tiling_rep.py [guides]
"""
import ctypes
import os
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import bean_amd  # noqa: F401
from bean_amd import _lib, engine, parallel
from bean_amd.preprocessing import synthetic as syn

G, R = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 5
data = syn.make_sorting_tiling_screen(G, R, seed=20240503)
data, ids = parallel.order_by_alleles(data)
eng = engine.HipSVI("MultiMixtureNormal", data.to("cuda:0"), num_steps=100, guide_ids=ids)
eng.run(20, graph_chunk=0)
torch.cuda.synchronize()
# waves per workgroup as the library chooses them (tiling_rep_waves): 1 where every SIMD gets at most one single-wave
# workgroup, else 4 at R = 5 (STAMP_W overrides, with BEAN_HIP_TILING_W)
W = int(os.environ.get("STAMP_W", "1" if (G + (64 // R) - 1) // (64 // R) <= 1024 else "4"))
Gw = 64 * W // R
n_wg = (G + Gw - 1) // Gw
mode = int(os.environ.get("BEAN_HIP_TILING_MAP", "1"))
n = (n_wg + 31) // 32 * 32 if mode else n_wg
buf = np.zeros(n * W * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(n, W, 8).astype(np.int64)
b = np.arange(n)
if mode:
    x, j, run = b & 7, b >> 3, n >> 5
    q = j // run
    sl = q * (n >> 2) + x * run + (j - q * run)
else:
    sl = b
live = sl * Gw < G
t0 = s[live][:, :, 0].min()
names = ["first loads", "draw", "forward mix", "likelihoods", "backward + rows", "Dirichlet + gradients", "tail"]
n_al = data.allele_mask.sum(1).numpy()
print(f"{int(live.sum())} workgroups x {W} waves, map mode {mode}; us (real-time clock, 10 ns)")
quarter = np.minimum(sl * 4 // n_wg, 3)
for qq in range(4):
    m = live & (quarter == qq)
    rows = s[m].reshape(-1, 8)
    d = np.diff(rows, axis=1) / 100.0
    g0 = int(sl[m].min() * Gw)
    g1 = min(int((sl[m].max() + 1) * Gw), G)
    print(f"quarter {qq}: {int(m.sum())} workgroups, alleles per guide {n_al[g0:g1].mean():.2f}; start {np.median(rows[:, 0] - t0) / 100:6.2f}  "
          f"end median {np.median(rows[:, 7] - t0) / 100:6.2f} max {(rows[:, 7] - t0).max() / 100:6.2f}")
    print("   " + "  ".join(f"{nm} {np.median(d[:, i]):5.2f}" for i, nm in enumerate(names)))
print(f"launch: {(s[live][:, :, 7].max() - t0) / 100:.2f} us")
eng.close()

"""Diagnostic (BEAN_STAMP build only): per-phase cycle stamps of k_param at the metric shape."""
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
T = int(data.n_targets)
ntb = (T + 15) // 16
ngb = (G + 255) // 256
n_waves = (ntb + ngb) * 4
buf = np.zeros(n_waves * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(n_waves, 8).astype(np.int64)
def show(name, rows, cols):
    rows = rows[rows[:, 0] > 0]
    print(name, "waves", len(rows))
    for c in cols:
        ok = rows[:, c] > 0
        v = rows[ok, c] - rows[ok, 0]
        if len(v): print(f"   stamp {c} - stamp 0: min {v.min():7d} p10 {int(np.percentile(v,10)):7d} median {int(np.median(v)):7d} p90 {int(np.percentile(v,90)):7d} max {v.max():7d}")
tw = s[: ntb * 4]
show("target blocks, wave 0 (owner lanes)", tw[0::4], [1, 2, 3, 4, 7])
show("target blocks, waves 1-3", np.concatenate([tw[1::4], tw[2::4], tw[3::4]]), [1, 3, 4, 7])
show("guide blocks", s[ntb * 4:], [1, 2, 4, 7])

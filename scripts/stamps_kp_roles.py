"""Diagnostic (-DBEAN_STAMP=5 build only, BEAN_HIP_LIB=<that build>): where k_param's block roles sit on the
100 MHz real-time clock (default: BASELINE config 5, 100k guides x 3 replicates, with the survival q0 site;
`metric [n_guides]`: the sorting metric shape).
One record per block (role order: targets, alpha_pi blocks, q0 blocks): 0 start, 1 q0 updated,
3 draws done, 4 counted in, 7 end."""
import ctypes
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import bean_amd  # noqa: F401
from bean_amd import _lib, engine
from bean_amd.preprocessing import synthetic as syn

CONFIG = sys.argv[1] if len(sys.argv) > 1 else "survival"
if CONFIG == "survival":
    G, R = 100000, 3
    data = syn.make_survival_variant_screen(G, R, seed=20240506).to("cuda:0")
else:  # the metric shape: no q0 blocks, 16 lanes per target
    G, R = int(sys.argv[2]) if len(sys.argv) > 2 else 50000, 5
    data = syn.make_sorting_variant_screen(G, R, seed=20240502).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=100)
eng.run(20, graph_chunk=0)
torch.cuda.synchronize()
T = int(data.n_targets)
gpb = 256
lpt = 4 if CONFIG == "survival" else 16  # kLanesPerTargetNarrow / kLanesPerTarget
ntb, ngb = (T * lpt + 255) // 256, (G + 255) // 256
ndb = (G + gpb - 1) // gpb if CONFIG == "survival" else 0
nrec = ntb + ngb + ndb
buf = np.zeros(nrec * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(nrec, 8).astype(np.int64)
t0 = s[:, 0][s[:, 0] > 0].min()


def show(name, rows, cols):
    print(f"{name}: {len(rows)} blocks   (us after the first block's start)")
    for c in cols:
        v = (rows[rows[:, c] > 0, c] - t0) / 100.0
        if len(v):
            print(f"   stamp {c}: min {v.min():6.2f} p10 {np.percentile(v, 10):6.2f} median {np.median(v):6.2f} "
                  f"p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}")


show("target blocks", s[:ntb], [0, 7])
show("alpha_pi blocks", s[ntb:ntb + ngb], [0, 7])
if ndb:
    show("q0 blocks", s[ntb + ngb:], [0, 1, 3, 4, 7])

"""Diagnostic (-DBEAN_STAMP=5 build only, BEAN_HIP_LIB=<that build>): where the block roles of the tiling k_param -
edit blocks, guide blocks, allele blocks (or k_allele's blocks, BEAN_HIP_ALLELE=split) - sit on the 100 MHz real-time
clock at BASELINE config 3.  One record per block: 0 start, 1 (allele blocks) the edit blocks have all counted in,
4 (edit blocks) counted in, 7 end.

    BEAN_HIP_LIB=build/variants/libbean_hip_stamp5.so [BEAN_HIP_ALLELE=split] python scripts/stamps_kp_tiling.py [guides]
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
E_ = int(data.n_targets)
A = int(data.allele_mask.shape[1])
n_live = int(((data.allele_mask[:, 1:] != 0)).sum())
ntb, ngb = (E_ * 4 + 255) // 256, (G * 8 + 255) // 256
nab = (n_live + 255) // 256
nrec = ntb + ngb + nab + 8
buf = np.zeros(nrec * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(nrec, 8).astype(np.int64)
t0 = s[:ntb + ngb, 0][s[:ntb + ngb, 0] > 0].min()
print(f"mode: {os.environ.get('BEAN_HIP_ALLELE', 'allele blocks')}; {ntb} edit blocks, {ngb} guide blocks, ~{nab} allele blocks")


def show(name, rows, cols):
    print(f"{name}: {len(rows)} blocks   (us after the first block's start)")
    for c in cols:
        v = (rows[rows[:, c] > 0, c] - t0) / 100.0
        if len(v):
            print(f"   stamp {c}: n {len(v):5d} min {v.min():6.2f} p10 {np.percentile(v, 10):6.2f} median {np.median(v):6.2f} "
                  f"p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}")


show("edit blocks", s[:ntb], [0, 4, 7])
show("guide blocks", s[ntb:ntb + ngb], [0, 7])
al = s[ntb + ngb:]
al = al[al[:, 0] > 0]
show("allele blocks", al, [0, 1, 7])
d = (al[:, 7] - al[:, 0]) / 100.0
print(f"   allele block duration: median {np.median(d):.2f} max {d.max():.2f} us;  after the poll matched: "
      f"{np.median((al[:, 7] - np.maximum(al[:, 1], al[:, 0])) / 100.0):.2f}")
eng.close()

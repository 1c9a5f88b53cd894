"""Item timeline of k_svi_async from a -DBEAN_ASYNC_STAMP build (BEAN_HIP_LIB=<that build>):
    python scripts/async_timeline.py [guides] [blocks]
Per item of four consecutive steps (real-time clock, 10 ns): pulled, dependencies seen, guide work + arrival done,
(finishing wave) finish published.  Prints where a wave's time goes."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bean_amd  # noqa: F401
from bean_amd import engine
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen

G = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
if len(sys.argv) > 2:
    os.environ["BEAN_HIP_ASYNC_BLOCKS"] = sys.argv[2]
os.environ["BEAN_HIP_STEP"] = "async"
R = 5
data = make_sorting_variant_screen(G, R, seed=7)
eng = engine.HipSVI("MixtureNormal", data.to("cuda:0"), num_steps=400)
eng.run(100, seed=3, resume=True)
eng.run(100, seed=3, resume=True)
torch.cuda.synchronize()
lib = eng.lib
fn = lib.bean_hip_async_stamps
fn.restype = ctypes.c_int64
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
n_tiles = (G + 63) // 64
all_items = (n_tiles + 7) // 8 * 8 * R
FIN_OFF = 1 << 22
buf = np.zeros(2 * FIN_OFF, dtype=np.uint64)
n = fn(eng._h, buf.ctypes.data, buf.size)
assert n == buf.size, (n, buf.size)
st = buf[:4 * all_items * 8].reshape(4, all_items, 8).astype(np.int64)
stf = buf[FIN_OFF:FIN_OFF + 4 * all_items * 8].reshape(4, all_items, 8).astype(np.int64)
ok = st[:, :, 0] > 0
t0 = st[:, :, 0][ok].min()
us = lambda x: (x - t0) / 100.0  # noqa: E731
out = {"guides": G, "blocks": os.environ.get("BEAN_HIP_ASYNC_BLOCKS", "default")}
for s in range(4):
    m = ok[s]
    pull, ready, done, fin = (st[s, :, k][m] for k in range(4))
    isfin = fin > 0
    q = lambda v: [round(float(np.percentile(v, p)), 2) for p in (10, 50, 90, 100)]  # noqa: E731
    out[f"step{s}"] = {
        "items": int(m.sum()), "finishers": int(isfin.sum()),
        "pulled_at_us_p10_50_90_max": q(us(pull)),
        "wait_for_deps_us": q((ready - pull) / 100.0),
        "guide_plus_arrival_us": q((done - ready) / 100.0),
        "finish_us": q((fin[isfin] - done[isfin]) / 100.0),
        "guide_math_us": q((st[s, :, 4][m] - ready) / 100.0),
        "arrival_us (stores drained + counter)": q((done - st[s, :, 4][m]) / 100.0),
        "finish_phases_us (call+boundary counters | sums, Adam, draw | Phi tables | guides | loss + drain + publish)": [
            q((stf[s, :, 0][m][isfin] - done[isfin]) / 100.0), q((stf[s, :, 1][m][isfin] - stf[s, :, 0][m][isfin]) / 100.0),
            q((stf[s, :, 2][m][isfin] - stf[s, :, 1][m][isfin]) / 100.0), q((stf[s, :, 3][m][isfin] - stf[s, :, 2][m][isfin]) / 100.0),
            q((fin[isfin] - stf[s, :, 3][m][isfin]) / 100.0)],
        "published_at_us": q(us(fin[isfin])),
    }
# per wave (block id): busy fraction between its first pull of step 0 and its last event of step 3
blk = st[:, :, 5]
span, busy = [], []
for b in np.unique(blk[ok]):
    sel = ok & (blk == b)
    p, r, d, f = (st[:, :, k][sel] for k in range(4))
    end = np.maximum(d, f)
    span.append(end.max() - p.min())
    busy.append(((d - r) + np.where(f > 0, f - d, 0)).sum())
span, busy = np.array(span), np.array(busy)
out["waves"] = len(span)
out["per_wave_span_us_mean"] = round(float(span.mean() / 100.0), 2)
out["per_wave_busy_fraction_mean"] = round(float((busy / span).mean()), 3)
out["four_steps_us"] = round(float((np.maximum(st[:, :, 2], st[:, :, 3])[ok].max() - t0) / 100.0), 2)
xcc = st[:, :, 7][ok] & 0xF
lab = blk[ok] & 7
out["xcc_of_group_label"] = {int(g): sorted(set(int(v) for v in xcc[lab == g])) for g in range(8)}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/async_timeline_{G}_{out['blocks']}.json", "w"), indent=1)
eng.close()

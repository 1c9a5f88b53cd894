"""Item and finish timeline of k_svi_async from a -DBEAN_ASYNC_STAMP build (BEAN_HIP_LIB=<that build>):
    python scripts/async_timeline.py [guides] [item blocks] [finisher blocks]
Per item of four consecutive steps (real-time clock, 10 ns): pulled, dependencies seen, guide math done, arrived; per tile:
when its last wave arrived, when the targets' / the guides' part of its finish started and was published, the phases of
the targets' part.  Prints where a wave's time goes."""
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
if len(sys.argv) > 3:
    os.environ["BEAN_HIP_ASYNC_FIN"] = sys.argv[3]
os.environ["BEAN_HIP_STEP"] = "async"
R = 5
data = make_sorting_variant_screen(G, R, seed=7)
eng = engine.HipSVI("MixtureNormal", data.to("cuda:0"), num_steps=400)
eng.run(100, seed=3, resume=True)
eng.run(100, seed=3, resume=True)
torch.cuda.synchronize()
fn = eng.lib.bean_hip_async_stamps
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
q = lambda v: [round(float(np.percentile(v, p)), 2) for p in (10, 50, 90, 100)] if len(v) else None  # noqa: E731
out = {"guides": G, "item_blocks": os.environ.get("BEAN_HIP_ASYNC_BLOCKS", "default"),
       "finisher_blocks": os.environ.get("BEAN_HIP_ASYNC_FIN", "default"), "percentiles": "p10, p50, p90, max; us"}
for s in range(4):
    m = ok[s]
    pull, ready, arrived, math = (st[s, :, k][m] for k in (0, 1, 2, 4))
    # per tile: rows of replicate 0 carry the finish stamps; the tile is complete when its last wave has arrived
    tiles = np.arange(n_tiles)
    rows0 = tiles * R
    complete = np.max(st[s, :n_tiles * R, 2].reshape(n_tiles, R), axis=1)
    f = stf[s, rows0, :]
    have_t, have_g = f[:, 4] > 0, f[:, 6] > 0
    out[f"step{s}"] = {
        "items": int(m.sum()),
        "pulled_at": q((pull - t0) / 100.0),
        "wait_for_dependencies": q((ready - pull) / 100.0),
        "guide_math": q((math - ready) / 100.0),
        "store_drain_and_arrival": q((arrived - math) / 100.0),
        "tile_complete_to_targets_part_started": q((f[have_t, 4] - complete[have_t]) / 100.0),
        "targets_part (start to published)": q((f[have_t, 5] - f[have_t, 4]) / 100.0),
        "targets_part_phases (ownership | sums, Adam, draw | Phi tables)": [
            q((f[have_t, 0] - f[have_t, 4]) / 100.0), q((f[have_t, 1] - f[have_t, 0]) / 100.0), q((f[have_t, 2] - f[have_t, 1]) / 100.0)],
        "tile_complete_to_guides_part_started": q((f[have_g, 6] - complete[have_g]) / 100.0),
        "guides_part (start to published)": q((f[have_g, 7] - f[have_g, 6]) / 100.0),
        "tile_complete_to_last_part_published": q((np.maximum(f[:, 5], f[:, 7])[have_t] - complete[have_t]) / 100.0),
    }
blk = st[:, :, 5]
span, busy = [], []
for b in np.unique(blk[ok]):
    sel = ok & (blk == b)
    p, r, d = (st[:, :, k][sel] for k in (0, 1, 2))
    span.append(d.max() - p.min())
    busy.append((d - r).sum())
span, busy = np.array(span), np.array(busy)
out["item_waves_seen"] = len(span)
out["item_wave_busy_fraction_mean (guide work only)"] = round(float((busy / span).mean()), 3)
last = np.maximum(np.maximum(st[:, :, 2], stf[:, :, 5]), stf[:, :, 7])
out["four_steps_us"] = round(float((last[ok].max() - t0) / 100.0), 2)
xcc = st[:, :, 7][ok] & 0xF
lab = blk[ok] & 7
out["xcc_of_group_label"] = {int(g): sorted(set(int(v) for v in xcc[lab == g])) for g in range(8)}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/async_timeline_{G}_{out['item_blocks']}_{out['finisher_blocks']}.json", "w"), indent=1)
eng.close()

"""Diagnostic (-DBEAN_STAMP=4 build only): the clock the chip holds inside k_guide_wave2 =
delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, 'DVFS give-back' item 6), after ~2 s of
back-to-back steps."""
import sys, ctypes, time
sys.path.insert(0, ".")
import numpy as np, torch
import bean_amd
from bean_amd import engine, _lib
from bean_amd.preprocessing.synthetic import make_sorting_variant_screen
G = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
data = make_sorting_variant_screen(G, 5, seed=20240502).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=60000, loss_capacity=60000)
t = time.time()
while time.time() - t < 2.0:
    eng.run(2000); torch.cuda.synchronize()
eng.run(50, graph_chunk=0); torch.cuda.synchronize()
n = 5 * (((G + 63) // 64 + 7) // 8 * 8)
buf = np.zeros(n * 8, dtype=np.uint64)
lib = _lib.load()
lib.bean_hip_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
assert lib.bean_hip_debug_stamps(eng._h, buf.ctypes.data, buf.size) == 0
s = buf.reshape(n, 8).astype(np.int64)
s = s[(s[:, 0] > 0) & (s[:, 2] > s[:, 0]) & (s[:, 3] > s[:, 1])]
clk = (s[:, 2] - s[:, 0]) / (s[:, 3] - s[:, 1]) * 100.0  # MHz
life_us = (s[:, 3] - s[:, 1]) / 100.0
print("waves", len(s), "in-kernel clock MHz: median %.0f p10 %.0f p90 %.0f" % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
print("wave lifetime us: median %.1f p10 %.1f p90 %.1f max %.1f" % (np.median(life_us), np.percentile(life_us, 10), np.percentile(life_us, 90), life_us.max()))
print("launch span us (first start to last end, real-time clock): %.1f" % ((s[:, 3].max() - s[:, 1].min()) / 100.0))
start = (s[:, 1] - s[:, 1].min()) / 100.0
print("wave start offsets us: median %.1f p90 %.1f max %.1f" % (np.median(start), np.percentile(start, 90), start.max()))

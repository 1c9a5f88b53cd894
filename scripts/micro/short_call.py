"""Per-call cost of a 20-step run (what the driver's bench line times): median over repeats, with the default
host wait policy and with hipDeviceScheduleSpin."""
import ctypes, json, os, sys, time
sys.path.insert(0, "/root/repo")
spin = "spin" in sys.argv[1:]
resume = "resume" in sys.argv[1:]
chunk = 0 if 'eager' in sys.argv[1:] else 64
guides = int(os.environ.get('GUIDES', '50000'))
if spin:
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(1)))  # hipDeviceScheduleSpin
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn
data = syn.make_sorting_variant_screen(guides, 5, seed=syn.BASE_SEED + 1).to("cuda:0")
eng = engine.HipSVI("MixtureNormal", data, num_steps=5000)
eng.run(5, resume=resume, graph_chunk=chunk); torch.cuda.synchronize()
ts = []
for _ in range(60):
    torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(20, resume=resume, graph_chunk=chunk); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
ts = sorted(ts)
print(json.dumps({"spin": spin, "resume": resume, "graph_chunk": chunk, "guides": guides, "us_per_step_median": ts[len(ts) // 2] / 20 * 1e6, "min": ts[0] / 20 * 1e6,
                  "p90": ts[int(len(ts) * 0.9)] / 20 * 1e6}))

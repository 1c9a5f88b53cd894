"""Per-step cost of the exchange families' stepping path against the plain device loop.

    python scripts/time_exchange.py [tag]     ->  gpurun_out/exchange_1rank[_tag].json

One rank, `nccl` process group (every collective is a copy, but the call path - host submission, RCCL
launch, stream interplay - is the one an N-rank fit pays per step): tiling (BASELINE config 3) and
survival (config 5) at full size and at 1/8 of the guides (what one of 8 ranks holds), each timed as
  svi_run         bean_hip_svi_run, hipGraph replay, no exchange           (the floor)
  exchanged       HipSVI.run_exchanged driven from Python with torch.distributed.all_reduce
  native          bean_hip_svi_run_exchanged: the library enqueues kernels + ncclAllReduce itself
  native_graph    ... replaying hipGraphs of 32 exchanged steps, collectives captured
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")

import torch
import torch.distributed as dist

import bean_amd  # noqa: F401
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


REPEATS = int(os.environ.get("XCHG_REPEATS", "5"))


def time_it(fn, steps, spread=None, key=None):
    """us per step of `steps` steps: the median of REPEATS timings in this process (min / median / max recorded
    in `spread[key]`: a host-driven loop is what a noisy neighbour or a bad stream -> queue mapping shows up in)."""
    fn(20)
    torch.cuda.synchronize()
    ts = []
    for _ in range(REPEATS):
        t = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) / steps * 1e6)
    ts.sort()
    if spread is not None:
        spread[key] = {"min": round(ts[0], 2), "median": round(ts[len(ts) // 2], 2), "max": round(ts[-1], 2), "n": len(ts)}
    return ts[len(ts) // 2]


def case(family, data, steps=300, **kw):
    out = {}
    if family == "MultiMixtureNormal":  # as run_inference hands a tiling screen over: guides ordered by allele count
        from bean_amd import parallel
        data, ids = parallel.order_by_alleles(data)
        if ids is not None:
            kw = dict(kw, guide_ids=ids)
    data = data.to(dev)
    cap = (REPEATS + 1) * steps + 100
    spread = out["spread_us"] = {}
    eng = engine.HipSVI(family, data, num_steps=cap, device=dev, **kw)
    out["svi_run_us"] = round(time_it(lambda n: eng.run(n), steps, spread, "svi_run"), 2)
    eng.close()
    eng = engine.HipSVI(family, data, num_steps=cap, device=dev, **kw)
    x = eng.exchange_buffers()
    out["exchange"] = {k: int(v.numel()) for k, v in x.items()}
    out["exchanged_us"] = round(time_it(lambda n: eng.run_exchanged(n, dist.all_reduce), steps, spread, "exchanged"), 2)
    out["exchanged_path"] = getattr(eng, "last_exchange_path", "python loop: 3 ctypes calls + torch.distributed.all_reduce per step")
    eng.close()
    out["overhead_us"] = round(out["exchanged_us"] - out["svi_run_us"], 2)
    for key, chunk in (("native_us", 0), ("native_graph_us", 32)):
        eng = engine.HipSVI(family, data, num_steps=cap, device=dev, **kw)
        if not eng.init_native_comm():
            out[key] = None
            eng.close()
            continue
        out[key] = round(time_it(lambda n: eng.run_exchanged(n, None, graph_chunk=chunk), steps, spread, key[:-3]), 2)
        out[key.replace("_us", "_path")] = eng.last_exchange_path
        msg = eng.lib.bean_hip_last_error().decode()
        if chunk and "could not be captured" in msg:
            out[key.replace("_us", "_path")] += " (capture refused: eager)"
        eng.close()
    return out


res = {}
for name, make, family, full in (("config3 tiling", lambda g: syn.make_sorting_tiling_screen(g, 5, seed=20240503), "MultiMixtureNormal", 50000),
                                 ("config5 survival", lambda g: syn.make_survival_variant_screen(g, 3, seed=20240506), "MixtureNormal", 100000)):
    for frac in (1, 8):
        only = os.environ.get("XCHG_CASES")  # e.g. "tiling/8,survival/1"
        if only and f"{name.split()[1]}/{frac}" not in only.split(","):
            continue
        g = full // frac
        data = make(g)
        kw = {}
        if family == "MixtureNormal":
            kw["t0_totals"] = (data.X[:, 0, :].float() + 1).sum(-1)
        res[f"{name}, {g} guides ({'full' if frac == 1 else '1/8: one rank of 8'})"] = case(family, data, **kw)
        print(json.dumps(res, indent=1), flush=True)
tag = ("_" + sys.argv[1]) if len(sys.argv) > 1 else ""
os.makedirs("gpurun_out", exist_ok=True)
out_path = f"gpurun_out/exchange_1rank{tag}.json"
if os.environ.get("XCHG_CASES") and os.path.exists(out_path):  # one case per process: merge into the file
    res = dict(json.load(open(out_path)), **res)
json.dump(res, open(out_path, "w"), indent=1)
dist.destroy_process_group()

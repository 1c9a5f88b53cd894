#!/usr/bin/env python
"""Headline benchmark: SVI steps/s of the variant sorting MixtureNormal model.

    python bench.py --gpus N --steps K --warmup W

One "step" is one full SVI step (draw, ELBO, gradient, ClippedAdam) of the
``bean run sorting variant`` model (``bean/model/run.py:376-377``) over a
synthetic 50k-guide x 5-replicate x (4 sort bins + bulk) screen that is already
resident in HBM.  With N > 1 (launched by ``torch.distributed.run``) every rank
holds its own 50k-guide shard (guides shard on target boundaries, no parameter is
shared in this family) and the per-step losses are summed across ranks with one
RCCL all-reduce per 100 steps, the cadence at which the reference reports the
loss (``run.py:378``).  ``value`` counts 50k-guide steps over all ranks.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definitions of
the ``roofline`` and ``cpu_baseline`` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GUIDES_PER_GPU = 50_000
N_REPS = 5
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
LOSS_SYNC_EVERY = 100


def cpu_baseline(data, seconds_budget=20.0):
    """Reference CPU path = the float64 eager-torch oracle, timed on this box's
    host cores on a bounded sample of the same workload."""
    import torch

    from oracle import elbo, svi

    torch.manual_seed(101)
    params = elbo.init_params("MixtureNormal", data)
    optim = svi.ClippedAdam(params, lr=0.01, lrd=0.1 ** (1 / 2000))
    # the eager path is partly overhead bound: give the CPU its best thread count (2 timed steps
    # each at a few counts), then spend the budget there
    avail = torch.get_num_threads()
    best, n_threads = None, avail
    for cand in sorted({min(avail, c) for c in (8, 16, 32, 64, avail)}):
        torch.set_num_threads(cand)
        svi.svi_step(elbo.mixture_normal_loss, data, params, optim)  # warm-up at this count
        t = time.perf_counter()
        for _ in range(2):
            svi.svi_step(elbo.mixture_normal_loss, data, params, optim)
        t = (time.perf_counter() - t) / 2
        if best is None or t < best:
            best, n_threads = t, cand
    torch.set_num_threads(n_threads)
    t0 = time.perf_counter()
    n = 0
    while n < 4 or (time.perf_counter() - t0 < seconds_budget and n < 200):
        svi.svi_step(elbo.mixture_normal_loss, data, params, optim)
        n += 1
    dt = time.perf_counter() - t0
    return {
        "value": n / dt,
        "unit": "steps/s",
        "cores": n_threads,
        "kind": "port",
        "sample": f"{n} SVI steps of the same 50k-guide screen (float64 eager-torch oracle, "
                  f"anomaly detection off, {dt:.1f} s; fastest of 8/16/32/64/{avail} threads)",
    }


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the newest committed
    rocprofv3 --pmc summary (profiles/*_traffic.json, written by
    scripts/summarize_prof.py with the gfx950 FETCH_SIZE correction), or None."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        return float(json.load(fh)["hbm_bytes_per_launch"]), os.path.basename(files[-1])


def valu_issue_roofline(kernel_name, kernel_ms):
    """Secondary roofline of the dominant kernel (it is bound by float64 VALU issue, not by HBM):
    VALU wave-instructions per launch from the committed rocprofv3 --pmc summary
    (profiles/*_counters.txt, SQ_INSTS_VALU) over the live kernel duration, against what a gfx950
    SIMD sustains on v_fma_f64 with 4 resident waves (profiles/*_valu_issue.txt, measured by
    scripts/micro/valu_rate.hip) times the 1024 SIMDs.  None when the summaries are absent."""
    import glob
    import re

    cfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.txt")))
    vfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_issue.txt")))
    if not cfiles or not vfiles or kernel_ms <= 0:
        return None
    insts = None
    lines = open(cfiles[-1]).read().splitlines()
    for i, ln in enumerate(lines):
        if ln.startswith("bean::" + kernel_name):
            for l2 in lines[i + 1:i + 12]:
                m = re.match(r"\s+SQ_INSTS_VALU\s+n=\s*\d+\s+mean=([0-9.e+]+)", l2)
                if m:
                    insts = float(m.group(1))
                    break
            if insts:
                break
    ns = None
    for ln in open(vfiles[-1]):
        m = re.match(r"fma_f64\s+waves/SIMD 4: .*-> ([0-9.]+) ns per wave-instr per SIMD", ln)
        if m:
            ns = float(m.group(1))
    if not insts or not ns:
        return None
    peak = 1024 / (ns * 1e-9)  # wave-instructions per second, 256 CUs x 4 SIMDs
    achieved = insts / (kernel_ms * 1e-3)
    return {"bound": "f64 VALU issue", "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s",
            "frac": achieved / peak, "valu_insts_per_launch": insts,
            "source": [os.path.basename(cfiles[-1]), os.path.basename(vfiles[-1])]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--guides", type=int, default=GUIDES_PER_GPU)
    ap.add_argument("--graph-chunk", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scale-by-acc", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import bean_amd  # noqa: F401
    from bean_amd import engine
    from bean_amd.preprocessing.synthetic import BASE_SEED, make_sorting_variant_screen

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal on a one-GPU box only: BEAN_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # exchanges over gloo (RCCL refuses two ranks on one device); never set by the driver
    rehearsal = os.environ.get("BEAN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    data_cpu = make_sorting_variant_screen(
        args.guides, N_REPS, seed=BASE_SEED + 1 + rank, with_accessibility=args.scale_by_acc
    )
    data = data_cpu.to(dev)
    total = args.warmup + args.steps
    eng = engine.HipSVI(
        "MixtureNormal", data, num_steps=max(total, 1), loss_capacity=total + 64,
        scale_by_accessibility=args.scale_by_acc, device=dev,
        guide_offset=rank * args.guides, target_offset=rank * data.n_targets, n_guides_total=world * args.guides,
    )

    def run_steps(n):
        done = 0
        while done < n:
            k = min(LOSS_SYNC_EVERY, n - done)
            first = eng.steps_done
            eng.run(k, seed=101, graph_chunk=args.graph_chunk)
            if world > 1:
                with torch.cuda.stream(eng.stream):
                    dist.all_reduce(eng.loss_hist[first:first + k])
            done += k

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    losses = eng.losses()

    # ---- dominant-kernel timing (HIP events on the launch stream, eager launches)
    eng_steps = 50
    prof = engine.HipSVI("MixtureNormal", data, num_steps=eng_steps, scale_by_accessibility=args.scale_by_acc,
                         device=dev)
    prof.set_profile(True)
    prof.run(eng_steps, seed=101, graph_chunk=0)
    torch.cuda.synchronize(dev)
    k_ms, k_n = prof.get_profile()
    step_bytes = prof.step_bytes
    kernel_name = prof.dominant_kernel
    prof.close()

    if rank == 0:
        value = (world * args.guides / GUIDES_PER_GPU) * args.steps / dt
        traffic, traffic_src = pmc_traffic() if args.guides == GUIDES_PER_GPU else (None, None)
        achieved = step_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "SVI steps/sec, 50k-guide x 20-sample sorting model",
            "value": value,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "variant sorting MixtureNormal" + ("+Acc" if args.scale_by_acc else "")
                            + f": {args.guides} guides x {N_REPS} reps x (4 sort bins + bulk) per GPU, "
                              f"{data.n_targets} targets, one SVI step = draw + ELBO + grad + ClippedAdam",
                "guides_per_gpu": args.guides,
                "n_reps": N_REPS,
                "n_condits": data.n_condits,
                "parallelism": f"guide-sharded x{world}, loss all-reduce every {LOSS_SYNC_EVERY} steps",
                "graph_chunk": args.graph_chunk,
                "value_counts": "50k-guide steps summed over ranks",
                "final_loss": losses[-1] if losses else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": step_bytes,
                "kernel_ms": k_ms,
                "kernel_launches_timed": k_n,
            },
        }
        if args.guides == GUIDES_PER_GPU:
            valu = valu_issue_roofline(kernel_name, k_ms)
            if valu is not None:
                out["roofline"]["valu_issue"] = valu
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data_cpu)
            out["config"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

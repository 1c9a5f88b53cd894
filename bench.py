#!/usr/bin/env python
"""Headline benchmark: SVI steps/s of the `bean run` hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config metric|tiling|survival] [--scaling weak|strong]

One "step" is one full SVI step (draw, ELBO, gradient, ClippedAdam; ``bean/model/run.py:376-377``) over
a synthetic screen that is already resident in HBM.

``--config`` (BASELINE.json ``configs``; default = the configuration ``metric`` is quoted on):
  metric    variant sorting MixtureNormal, 50k guides x 5 replicates x (4 sort bins + bulk)   [configs[1] shape x10]
  tiling    tiling sorting MultiMixtureNormal, 50k guides / ~193k edited alleles x 5 reps     [configs[2]]
  survival  survival MixtureNormal, 100k guides x 6 timepoints x 3 replicates                 [configs[4]]

``--scaling`` with N > 1 ranks (launched by ``torch.distributed.run``, one process per GPU, RCCL):
  weak    (default) every rank holds its own screen of the configured size; ``value`` counts steps of
          one such screen summed over ranks.
  strong  ONE screen (default 500k guides: BASELINE configs[3]) is cut on target boundaries into N
          shards (``parallel.plan_shards``), one per rank, exactly as ``run_inference`` does under
          torchrun; ``value`` = steps/s of the WHOLE screen.  The variant sorting family shares no
          parameter across shards, so the only collective is the all-reduce of the loss window every 100
          steps (the reference's reporting cadence, ``run.py:378``).

Rank 0 prints ONE JSON line; DESIGN.md section 4 defines the ``roofline`` and ``cpu_baseline`` objects.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_VALU_DATASHEET = 78.6e12 / 2 / 64  # wave-instructions / s: 78.6 TFLOP/s FP64 vector, 2 flop x 64 lanes per FMA
LOSS_SYNC_EVERY = 100
STRONG_GUIDES = 500_000


def workload(name, guides, rank, acc=False):
    """(family, screen on CPU, oracle loss fn + kwargs, description)."""
    from bean_amd.preprocessing import synthetic as syn
    from oracle import elbo
    from oracle import survival as osurv

    if name == "metric":
        data = syn.make_sorting_variant_screen(guides, 5, seed=syn.BASE_SEED + 1 + rank, with_accessibility=acc)
        return ("MixtureNormal", data, elbo.mixture_normal_loss, dict(scale_by_accessibility=acc) if acc else {},
                elbo.init_params, f"variant sorting MixtureNormal{'+Acc' if acc else ''}: {guides} guides x 5 reps x "
                f"(4 sort bins + bulk), {data.n_targets} targets")
    if name == "tiling":
        data = syn.make_sorting_tiling_screen(guides, 5, seed=20240503 + rank)
        n_alleles = int(data.allele_mask.sum()) - data.n_guides
        return ("MultiMixtureNormal", data, elbo.multi_mixture_normal_loss, dict(sparse=True), elbo.init_params,
                f"tiling sorting MultiMixtureNormal: {guides} guides, {n_alleles} edited alleles, {data.n_edits} edits "
                f"x 5 reps x (4 sort bins + bulk)")
    if name == "survival":
        data = syn.make_survival_variant_screen(guides, 3, seed=20240506 + rank)
        return ("MixtureNormal", data, osurv.mixture_normal_loss, {}, osurv.init_params,
                f"survival MixtureNormal: {guides} guides x {data.n_condits} timepoints x 3 reps")
    raise SystemExit(f"unknown --config {name}")


DEFAULT_GUIDES = {"metric": 50_000, "tiling": 50_000, "survival": 100_000}


def cpu_baseline(family, data, loss_fn, loss_kw, init_fn, seconds_budget=20.0):
    """Reference CPU path = the float64 eager-torch oracle, timed on this box's host cores on a
    bounded sample (SVI steps of the same screen)."""
    import torch

    from oracle import svi

    torch.manual_seed(101)
    params = init_fn(family, data)
    optim = svi.ClippedAdam(params, lr=0.01, lrd=0.1 ** (1 / 2000))
    # the eager path is partly overhead bound: give the CPU its best thread count (2 timed steps each
    # at a few counts), then spend the budget there
    avail = torch.get_num_threads()
    best, n_threads = None, avail
    for cand in sorted({min(avail, c) for c in (8, 16, 32, 64, avail)}):
        torch.set_num_threads(cand)
        svi.svi_step(loss_fn, data, params, optim, **loss_kw)  # warm-up at this count
        t = time.perf_counter()
        for _ in range(2):
            svi.svi_step(loss_fn, data, params, optim, **loss_kw)
        t = (time.perf_counter() - t) / 2
        if best is None or t < best:
            best, n_threads = t, cand
    torch.set_num_threads(n_threads)
    t0 = time.perf_counter()
    n = 0
    while n < 4 or (time.perf_counter() - t0 < seconds_budget and n < 200):
        svi.svi_step(loss_fn, data, params, optim, **loss_kw)
        n += 1
    dt = time.perf_counter() - t0
    return {
        "value": n / dt,
        "unit": "steps/s",
        "cores": n_threads,
        "kind": "port",
        "sample": f"{n} SVI steps of the same {data.n_guides}-guide screen (float64 eager-torch oracle, "
                  f"anomaly detection off, {dt:.1f} s; fastest of 8/16/32/64/{avail} threads)",
    }


def _newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def pmc_traffic(config):
    """HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 --pmc summary
    of this configuration (profiles/rNN_traffic[_<config>].json, written by scripts/summarize_prof.py
    with the gfx950 FETCH_SIZE correction), or None."""
    suffix = "" if config == "metric" else "_" + config
    path = _newest(f"r[0-9][0-9]_traffic{suffix}.json")
    if not path:
        return None, None
    with open(path) as fh:
        return float(json.load(fh)["hbm_bytes_per_launch"]), os.path.basename(path)


def valu_issue_roofline(config, kernel_name, kernel_ms):
    """Secondary roofline of the dominant kernel (it is bound by float64 VALU issue, not by HBM):
    VALU wave-instructions per launch (SQ_INSTS_VALU, newest profiles/rNN_counters[_<config>].txt)
    over the live kernel duration, against (a) the data-sheet float64 vector rate (78.6 TFLOP/s =
    614 G wave-instr/s) and (b) what a gfx950 SIMD sustained on v_fma_f64 with 4 resident waves in
    scripts/micro/valu_rate.hip (profiles/r01_valu_issue.txt) times the 1024 SIMDs."""
    suffix = "" if config == "metric" else "_" + config
    cfile = _newest(f"r[0-9][0-9]_counters{suffix}.txt")
    vfile = _newest("r[0-9][0-9]_valu_issue.txt")
    if not cfile or kernel_ms <= 0:
        return None
    insts = None
    lines = open(cfile).read().splitlines()
    for i, ln in enumerate(lines):
        if ln.startswith("bean::" + kernel_name + "<") or ln.startswith("bean::" + kernel_name + " "):
            for l2 in lines[i + 1:i + 12]:
                m = re.match(r"\s+SQ_INSTS_VALU\s+n=\s*\d+\s+mean=([0-9.e+]+)", l2)
                if m:
                    insts = float(m.group(1))
                    break
            if insts:
                break
    if not insts:
        return None
    achieved = insts / (kernel_ms * 1e-3)
    out = {"bound": "f64 VALU issue", "achieved": achieved / 1e9, "unit": "G wave-instr/s",
           "peak_datasheet": F64_VALU_DATASHEET / 1e9, "frac_datasheet": achieved / F64_VALU_DATASHEET,
           "valu_insts_per_launch": insts, "source": [os.path.basename(cfile)]}
    ns = None
    if vfile:
        for ln in open(vfile):
            m = re.match(r"fma_f64\s+waves/SIMD 4: .*-> ([0-9.]+) ns per wave-instr per SIMD", ln)
            if m:
                ns = float(m.group(1))
    if ns:
        peak = 1024 / (ns * 1e-9)  # 256 CUs x 4 SIMDs
        out.update({"peak": peak / 1e9, "frac": achieved / peak})
        out["source"].append(os.path.basename(vfile))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", choices=("metric", "tiling", "survival"), default="metric")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--guides", type=int, default=0, help="guides per GPU (weak) or of the whole screen (strong)")
    ap.add_argument("--graph-chunk", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scale-by-acc", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import bean_amd  # noqa: F401
    from bean_amd import engine, parallel

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

    strong = args.scaling == "strong"
    if strong and args.config != "metric":
        raise SystemExit("--scaling strong is defined for --config metric (BASELINE configs[3])")
    guides = args.guides or (STRONG_GUIDES if strong else DEFAULT_GUIDES[args.config])
    # strong: every rank builds the SAME whole screen and keeps its target-aligned shard
    family, data_cpu, loss_fn, loss_kw, init_fn, desc = workload(args.config, guides, 0 if strong else rank,
                                                                 acc=args.scale_by_acc)
    eng_kw = dict(scale_by_accessibility=True) if args.scale_by_acc else {}
    if strong:
        shards = parallel.plan_shards(data_cpu.target_lengths.numpy(), world)
        sh = shards[rank]
        shard_cpu = parallel.shard_screen(data_cpu, sh)
        offsets = dict(guide_offset=sh[0], target_offset=sh[2], n_guides_total=data_cpu.n_guides)
    else:
        shard_cpu = data_cpu
        n_t = getattr(data_cpu, "n_targets", 0)
        offsets = dict(guide_offset=rank * guides, target_offset=rank * n_t, n_guides_total=world * guides)
        if family == "MultiMixtureNormal" or data_cpu.selection == "survival":
            offsets = {}  # independent screens: these families couple guides through shared quantities
    data = shard_cpu.to(dev)
    total = args.warmup + args.steps
    eng = engine.HipSVI(family, data, num_steps=max(total, 1), loss_capacity=total + 64, device=dev, **eng_kw,
                        **offsets)

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

    # ---- dominant-kernel timing (HIP events with the kernel's own timestamps, on the launch stream)
    eng_steps = 50
    prof = engine.HipSVI(family, data, num_steps=eng_steps, device=dev, **eng_kw, **offsets)
    prof.set_profile(True)
    prof.run(eng_steps, seed=101, graph_chunk=0)
    torch.cuda.synchronize(dev)
    k_ms, k_n = prof.get_profile()
    step_bytes = prof.step_bytes
    kernel_name = prof.dominant_kernel
    prof.close()

    if rank == 0:
        value = args.steps / dt if strong else world * args.steps / dt
        std_size = guides == (STRONG_GUIDES if strong else DEFAULT_GUIDES[args.config]) and not args.scale_by_acc
        traffic, traffic_src = pmc_traffic(args.config) if (std_size and not strong) else (None, None)
        achieved = step_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        metric = {"metric": "SVI steps/sec, 50k-guide x 20-sample sorting model",
                  "tiling": "SVI steps/sec, 50k-guide x 200k-allele tiling sorting model",
                  "survival": "SVI steps/sec, 100k-guide x 6-timepoint x 3-rep survival model"}[args.config]
        if strong:
            metric = f"SVI steps/sec, {guides}-guide x 20-sample sorting screen guide-sharded over the ranks"
        out = {
            "metric": metric,
            "value": value,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": desc + "; one SVI step = draw + ELBO + grad + ClippedAdam"
                            + (f"; this is the WHOLE screen, cut into {world} target-aligned shards" if strong
                               else "; per GPU"),
                "config": args.config,
                "guides": guides,
                "guides_this_rank": int(data.n_guides),
                "n_reps": int(data.n_reps),
                "n_condits": int(data.n_condits),
                "parallelism": (f"one screen guide-sharded x{world}" if strong else f"one screen per GPU x{world}")
                               + f", loss all-reduce every {LOSS_SYNC_EVERY} steps",
                "graph_chunk": args.graph_chunk,
                "value_counts": "steps of the whole screen" if strong else "steps of one screen, summed over ranks",
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
        if std_size and not strong:
            valu = valu_issue_roofline(args.config, kernel_name, k_ms)
            if valu is not None:
                out["roofline"]["valu_issue"] = valu
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(family, shard_cpu, loss_fn, loss_kw, init_fn)
            out["config"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Time one SVI step of BASELINE.json's configurations (the non-default ones are not bench lines): the
metric shape, tiling sorting (config 3), one rank's shard of the 500k-guide variant screen and the whole
screen (config 4: the one-GPU ingredients of the strong-scaling projection), survival (config 5); plus
the guide kernel over a range of screen sizes.  Writes gpurun_out/configs.json (copied to
profiles/rNN_configs.json)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bean_amd
from bean_amd import engine
from bean_amd.preprocessing import synthetic as syn

def fit(family, data, steps, **kw):
    if family == "MultiMixtureNormal":  # as run_inference hands a tiling screen over: guides ordered by allele count
        from bean_amd import parallel
        data, ids = parallel.order_by_alleles(data)
        if ids is not None:
            kw = dict(kw, guide_ids=ids)
    data = data.to("cuda:0")
    eng = engine.HipSVI(family, data, num_steps=2 * steps + 200, **kw)
    # (one untimed call of the timed call's shape: whatever the first such call sets up - graphs of the chunk sizes it
    # needs - is then in place)
    eng.run(50); eng.run(steps); torch.cuda.synchronize()
    t = time.perf_counter(); eng.run(steps); torch.cuda.synchronize(); dt = time.perf_counter() - t
    out = {"us_per_step": dt / steps * 1e6, "loss_first": eng.losses()[0], "loss_last": eng.losses()[-1]}
    eng.close()
    for mode, key in ((1, "kernel_us"), (2, "k_param_us")):
        prof = engine.HipSVI(family, data, num_steps=100, **kw); prof.set_profile(mode); prof.run(30, graph_chunk=0)
        torch.cuda.synchronize(); ms, n = prof.get_profile()
        out[key] = ms * 1e3
        if mode == 1:
            out.update(dominant_kernel=prof.dominant_kernel, algorithmic_bytes=prof.step_bytes,
                       GBps=prof.step_bytes / (ms * 1e-3) / 1e9)
        prof.close()
    return out

res = {}
which = sys.argv[1:] or ["metric", "tiling", "variant500k_shard", "survival", "sizes"]
if "metric" in which:
    d = syn.make_sorting_variant_screen(50000, 5, seed=syn.BASE_SEED + 1)
    res["metric: variant sorting 50k guides x (4 bins+bulk) x 5 reps, MixtureNormal"] = fit("MixtureNormal", d, 1000)
if "tiling" in which:
    d = syn.make_sorting_tiling_screen(50000, 5, seed=20240503)
    res["config3 tiling sorting 50k guides x ~200k alleles x (4 bins+bulk) x 5 reps, MultiMixtureNormal"] = dict(
        fit("MultiMixtureNormal", d, 300), n_edits=int(d.n_edits), n_alleles=int(d.allele_mask.sum()) - 50000)
if "variant500k_shard" in which:
    d = syn.make_sorting_variant_screen(62500, 5, seed=20240504)
    a = fit("MixtureNormal", d, 1000)
    res["config4 one of 8 shards (62.5k guides) of the 500k-guide variant sorting screen, MixtureNormal"] = a
    d = syn.make_sorting_variant_screen(500000, 5, seed=20240505)
    b = fit("MixtureNormal", d, 200)
    res["config4 whole 500k-guide variant sorting screen on ONE GPU, MixtureNormal"] = b
    res["config4 strong-scaling projection 1 -> 8 GPUs (whole-screen step / shard step; no RCCL cost: the family exchanges nothing per step)"] = b["us_per_step"] / a["us_per_step"]
if "survival" in which:
    d = syn.make_survival_variant_screen(100000, 3, seed=20240506)
    res["config5 survival 100k guides x 6 timepoints x 3 reps, MixtureNormal"] = fit("MixtureNormal", d, 500)
if "sizes" in which:
    rows = {}
    for G in (25000, 50000, 62500, 100000, 250000, 500000):
        d = syn.make_sorting_variant_screen(G, 5, seed=20240600 + G // 1000)
        r = fit("MixtureNormal", d, 300)
        rows[str(G)] = {k: round(r[k], 2) for k in ("us_per_step", "kernel_us", "k_param_us")}
        rows[str(G)]["kernel_us_per_50k_guides"] = round(r["kernel_us"] * 50000 / G, 2)
    res["guide kernel and step time over screen sizes (variant sorting MixtureNormal, 5 reps)"] = rows
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/configs.json", "w"), indent=1)
print(json.dumps(res, indent=1))

"""Condense rocprofv3 CSV output (scripts/profile.sh) into profiles/<tag>_*.

    python scripts/summarize_prof.py gpurun_out/prof_<tag>[_<config>] <tag> [config]

Writes (with suffix _<config> for the tiling / survival configurations) profiles/<tag>_kernel_stats.csv (the --stats table restricted to this
library's kernels), profiles/<tag>_counters.txt (mean per dispatch of every PMC
counter collected, per kernel) and profiles/<tag>_traffic.json (HBM bytes per
launch of the dominant kernel: FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950, both counters are
in KiB).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, tag = sys.argv[1], sys.argv[2]
config = sys.argv[3] if len(sys.argv) > 3 else "metric"
suf = "" if config == "metric" else "_" + config
os.makedirs("profiles", exist_ok=True)


def newest(pattern):
    """The newest match only: gpurun MERGES a call's files into gpurun_out/, so an earlier call's CSVs (other
    process ids in their names) are still lying beside this one's."""
    fs = glob.glob(pattern, recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


for f in newest(os.path.join(root, "kt", "**", "*kernel_stats.csv")):
    rows = list(csv.reader(open(f)))
    keep = [rows[0]] + [r for r in rows[1:] if "bean::" in r[0]]
    with open(f"profiles/{tag}_kernel_stats{suf}.csv", "w", newline="") as out:
        csv.writer(out).writerows(keep)
# register / scratch / LDS columns from the code object (what occupancy is decided by), not from the profiler's
# VGPR_Count / LDS_Block_Size (granulated allocation of one register file; static LDS segment only)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    import kernel_resources as _kr

    _lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crispr-bean_amd", "lib", "libbean_hip.so")
    RES = {r["kernel"]: r for r in _kr.kernels(_lib)}
except Exception as exc:  # noqa: BLE001
    print("code-object notes unavailable:", exc)
    RES = {}
LDS_DYN = None
try:
    bj = json.loads(open(os.path.join(root, "bench_kt.json")).read().strip().splitlines()[-1])
    LDS_DYN = (bj["roofline"]["kernel"], bj["roofline"].get("kernel_resources", {}).get("lds_dynamic_bytes"))
except Exception:  # noqa: BLE001
    pass


def resources(k):
    r = RES.get(k)
    if not r:
        return ""
    dyn = f" + {LDS_DYN[1]} B dynamic (requested by the library)" if LDS_DYN and LDS_DYN[1] is not None and ("bean::" + LDS_DYN[0]) in k else ""
    return (f"code object: vgpr {r.get('vgpr_count')} agpr {r.get('agpr_count')} sgpr {r.get('sgpr_count')} "
            f"vgpr_spills {r.get('vgpr_spill_count')} scratch {r.get('private_segment_fixed_size')} B/lane "
            f"lds {r.get('group_segment_fixed_size')} B static{dyn}")


lines, traffic, calib = [], {}, {}
dominant = None
# k_svi_async runs ALL the steps of a call in one launch: scripts/profile_async.sh makes every launch of a pass the
# same length and names it here, so that counters are also given per SVI step (what bench.py's roofline prices)
ASYNC_STEPS = int(os.environ.get("ASYNC_STEPS_PER_LAUNCH", "0"))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in newest(os.path.join(root, sub, "**", "*counter_collection.csv")):
        agg = defaultdict(lambda: defaultdict(list))
        meta = {}
        for row in csv.DictReader(open(f)):
            if "bean::" not in row["Kernel_Name"]:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta[k] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["Scratch_Size"],
                       row["LDS_Block_Size"], row["Workgroup_Size"], row["Grid_Size"])
        lines.append(f"== {sub}: mean per dispatch")
        for k, cs in sorted(agg.items()):
            lines.append(f"{k}  [{resources(k) or 'code object: n/a'}; wg {meta[k][5]} grid {meta[k][6]}]")
            for c, v in sorted(cs.items()):
                lines.append(f"    {c:24s} n={len(v):5d} mean={sum(v) / len(v):.6g}")
                if "k_svi_async" in k and ASYNC_STEPS:
                    lines.append(f"    {c + ' / step':24s} n={len(v):5d} mean={sum(v) / len(v) / ASYNC_STEPS:.6g}"
                                 f"   (launches of {ASYNC_STEPS} steps)")
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):
                        traffic[c] = sum(v) / len(v) / ASYNC_STEPS
                        dominant = k + f" (per SVI step: launches of {ASYNC_STEPS} steps)"
                elif "k_guide" in k and c in ("FETCH_SIZE", "WRITE_SIZE") and not (dominant and "k_svi_async" in dominant):
                    traffic[c] = sum(v) / len(v)
                    dominant = k
                if "k_prepare" in k and c == "FETCH_SIZE":
                    calib[c] = sum(v) / len(v)
open(f"profiles/{tag}_counters{suf}.txt", "w").write("\n".join(lines) + "\n")
if traffic:
    fetch = traffic.get("FETCH_SIZE", 0.0) * 1024 * 2  # KiB; gfx950 reports half of a coalesced stream
    write = traffic.get("WRITE_SIZE", 0.0) * 1024
    out = {"kernel": dominant, "fetch_bytes_corrected": fetch, "write_bytes": write,
           "hbm_bytes_per_launch": fetch + write,
           "note": "FETCH_SIZE x2 (gfx950 correction of MI355X_MICROARCH.md, calibrated there for 16-B/lane streams; "
                   "these kernels load 4-8 B per lane, so the corrected figure is an upper bound), WRITE_SIZE as read; "
                   "separate --pmc passes"}
    if calib:
        # k_prepare reads every count / mask tensor exactly once with the same 4-B/lane loads: its raw
        # FETCH_SIZE against that known byte count calibrates the correction for this access width
        out["k_prepare_fetch_size_raw_bytes"] = calib["FETCH_SIZE"] * 1024
    json.dump(out, open(f"profiles/{tag}_traffic{suf}.json", "w"), indent=1)
print("\n".join(lines))

"""Condense rocprofv3 CSV output (scripts/profile.sh) into a small text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = []
for f in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
    out.append(f"== kernel stats ({os.path.relpath(f, root)})")
    out.append(open(f).read().strip())
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        out.append(f"== counters, mean per dispatch ({sub})")
        for k, cs in sorted(agg.items()):
            for c, v in sorted(cs.items()):
                out.append(f"{k:60s} {c:24s} n={len(v):5d} mean={sum(v)/len(v):.6g}")
print("\n".join(out))

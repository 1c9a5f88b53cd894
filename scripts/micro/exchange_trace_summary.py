"""Summarise a rocprofv3 --kernel-trace of scripts/micro/exchange_trace.py: the last 40 steps' kernels in time order,
per kernel name: count, mean duration, the queue(s) it ran on, and the mean idle gap in FRONT of it."""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
guides = [i for i, r in enumerate(rows) if "k_guide" in r["Kernel_Name"]]
first = guides[-40]
rows = rows[first:]
stat = defaultdict(lambda: [0, 0.0, set(), 0.0])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-60:]
    st = stat[name]
    st[0] += 1
    st[1] += e - s
    st[2].add(r.get("Queue_Id", "?"))
    if prev_end is not None:
        st[3] += max(0, s - prev_end)
    prev_end = e if prev_end is None else max(prev_end, e)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} kernels over {span / 1e3:.1f} us = {span / 40e3:.2f} us per step")
for name, (n, dur, q, gap) in stat.items():
    print(f"{name:62s} n={n:4d} mean {dur / n / 1e3:7.2f} us  gap before {gap / n / 1e3:6.2f} us  queues {sorted(q)}")

#!/bin/bash
# quick SQ counter pass: bash scripts/pmc_quick.sh <tag> [counters...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 10 --warmup 2 --graph-chunk 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list); dur=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bean::k_" in row["Kernel_Name"] and "prepare" not in row["Kernel_Name"] and "set_step" not in row["Kernel_Name"]:
            k=row["Kernel_Name"].split("(")[0][-36:]
            agg[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
            dur[k].append((int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e3)
for k,v in sorted(dur.items()): print(k, "mean_us=%.1f n=%d"%(sum(v)/len(v), len(v)))
for k,v in sorted(agg.items()): print(k, "mean=%.6g"%(sum(v)/len(v)))
PY

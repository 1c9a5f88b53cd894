#!/bin/bash
# quick SQ counter pass for the dominant kernel: bash scripts/pmc_quick.sh <tag> [counters...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 10 --warmup 2 --graph-chunk 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_guide" in row["Kernel_Name"] or "k_param<true, true, true>" in row["Kernel_Name"]:
            agg[(row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
            meta=(row["VGPR_Count"],row["Accum_VGPR_Count"],row["SGPR_Count"],row["Scratch_Size"],row["LDS_Block_Size"],row["Workgroup_Size"],row["Grid_Size"])
for k,v in sorted(agg.items()): print(k, "n=%d mean=%.6g"%(len(v), sum(v)/len(v)))
PY

#!/bin/bash
# SQ stall counters of the tiling step's kernels (two passes of <= 8 SQ counters): bash scripts/pmc_tiling.sh <tag>
TAG=${1:-dev}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --config tiling --steps 12 --warmup 3 --graph-chunk 0 --no-cpu-baseline --no-strong --no-other-configs"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1)); OUT=$REPO/gpurun_out/pmc_tiling_$TAG/p$i; mkdir -p $OUT
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -- $B > $OUT/bench.json 2> $OUT/err.log
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$REPO/gpurun_out/pmc_tiling_$TAG/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "bean::k_guide_tiling" in row["Kernel_Name"] or "k_param<true, true, true" in row["Kernel_Name"]:
            k=row["Kernel_Name"].split("(")[0][-40:]
            agg[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
for k,v in sorted(agg.items()): print(k[0], k[1], "mean=%.6g n=%d"%(sum(v)/len(v), len(v)))
PY

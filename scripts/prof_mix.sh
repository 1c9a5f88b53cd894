#!/bin/bash
# Instruction mix of the dominant kernel (separate PMC pass, kernel-trace only): which part of the VALU
# stream is float64 arithmetic.  gpurun -- 'bash scripts/prof_mix.sh [metric|tiling|survival]'
set -e
CONFIG=${1:-metric}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_mix_$CONFIG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --config $CONFIG --no-cpu-baseline --steps 20 --warmup 4 --graph-chunk 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $OUT/a -- $B > $OUT/a.json 2> $OUT/a.err
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -- $B > $OUT/b.json 2> $OUT/b.err
cd $REPO && python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "k_guide" not in k and "k_param<true, true, true>" not in k and "k_step" not in k: continue
    print(k[:70])
    for c, v in sorted(d.items()): print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
PY

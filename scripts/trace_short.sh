#!/bin/bash
# kernel timeline of the driver's short bench call (20 steps): gpurun -- 'bash scripts/trace_short.sh'
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_short
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong --no-other-configs > $OUT/bench.json 2> $OUT/err.txt
cd $REPO && python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_short/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a call = the launches up to and including a k_loss_finalize (a resumed call has no k_set_step at its head)
ends = [i for i, r in enumerate(rows) if "k_loss_finalize" in r["Kernel_Name"]]
starts = [0] + [e + 1 for e in ends[:-1]]
for s, e in list(zip(starts, ends))[-3:]:
    grp = rows[s:e + 1]
    n_guide = sum("k_guide_wave2" in r["Kernel_Name"] for r in grp)
    t0 = int(grp[0]["Start_Timestamp"])
    span = (int(grp[-1]["End_Timestamp"]) - t0) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in grp) / 1e3
    print(f"call with {n_guide} steps: span {span:.1f} us, kernels busy {busy:.1f} us, gaps {span - busy:.1f} us")
    n_async = sum("k_svi_async" in r["Kernel_Name"] for r in grp)
    if n_async:  # (round 5: the call is k_async_head, ONE k_svi_async launch of all its steps, k_loss_finalize)
        prev = t0
        for r in grp:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            name = r["Kernel_Name"].split("(")[0][-40:]
            print(f"   +{(st - prev) / 1e3:6.1f} gap  {(en - st) / 1e3:7.1f} us  {name}")
            prev = en
    if n_guide == 20:
        prev = t0
        for r in grp:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            name = r["Kernel_Name"].split("(")[0][-40:]
            print(f"   +{(st - prev) / 1e3:6.1f} gap  {(en - st) / 1e3:7.1f} us  {name}")
            prev = en
PY

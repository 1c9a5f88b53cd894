#!/bin/bash
# One gpurun call: a pytest selection, then kernel A/B timings unless the tests were killed.
#   bash scripts/gpu_round.sh "<pytest args>" "<kernel_ab args>"
mkdir -p gpurun_out
timeout -k 10 ${PYTEST_TIMEOUT:-900} python -m pytest $1 > gpurun_out/pytest.log 2>&1
rc=$?
tail -25 gpurun_out/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then echo "tests killed: skipping the rest"; exit $rc; fi
if [ -n "$2" ]; then
  timeout -k 10 ${AB_TIMEOUT:-900} python scripts/kernel_ab.py $2 2>&1 | tee gpurun_out/ab.log
fi
exit 0

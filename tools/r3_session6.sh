#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== full GPU suite" | tee $OUT/s6.log
timeout -k 10 1700 python -m pytest tests -x -q -m gpu > $OUT/pytest_s6.log 2>&1; tail -n 12 $OUT/pytest_s6.log | tee -a $OUT/s6.log

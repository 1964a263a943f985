#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== replays" | tee $OUT/s9.log
for s in 31004365 31006136; do timeout -k 10 200 python tools/fuzz_parity.py --only $s 2>/dev/null | tail -n 1 | cut -c1-500 | tee -a $OUT/s9.log; done
echo "== campaign with the watertight test forced, 5 minutes" | tee -a $OUT/s9.log
timeout -k 10 500 python tools/fuzz_parity.py --seconds 300 --seed 37 --force-opt 10=1 --out $OUT/fuzz_parity_watertight.json > $OUT/fuzz_wt.log 2>&1; tail -n 2 $OUT/fuzz_wt.log | cut -c1-900 | tee -a $OUT/s9.log

#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== config matrix" | tee $OUT/s8.log
timeout -k 10 900 python tools/config_matrix.py --out $OUT/configs.json > $OUT/configs.log 2>&1; tail -n 2 $OUT/configs.log | cut -c1-300 | tee -a $OUT/s8.log
echo "== shard probe (device-built tree)" | tee -a $OUT/s8.log
timeout -k 10 300 python tools/shard_probe.py 2>/dev/null | tail -n 1 | tee $OUT/shard_probe.jsonl | cut -c1-600 | tee -a $OUT/s8.log
echo "== randomised differential campaign, 8 minutes" | tee -a $OUT/s8.log
timeout -k 10 700 python tools/fuzz_parity.py --seconds 480 --seed 31 --out $OUT/fuzz_parity.json > $OUT/fuzz.log 2>&1; tail -n 3 $OUT/fuzz.log | cut -c1-900 | tee -a $OUT/s8.log

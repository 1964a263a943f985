#!/bin/bash
# long randomised campaigns on the final code: mixed options, the watertight test forced, the dissolve stage forced
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
timeout -k 10 1000 python tools/fuzz_parity.py --seconds ${CAMPAIGN_SECONDS:-420} --seed ${CAMPAIGN_SEED:-41} --out $OUT/campaign_mixed.json > $OUT/campaign_mixed.log 2>&1; tail -n 1 $OUT/campaign_mixed.log | cut -c1-700
timeout -k 10 1000 python tools/fuzz_parity.py --seconds ${CAMPAIGN_SECONDS:-420} --seed ${CAMPAIGN_SEED:-41}1 --force-opt 10=1 --out $OUT/campaign_watertight.json > $OUT/campaign_wt.log 2>&1; tail -n 1 $OUT/campaign_wt.log | cut -c1-700
timeout -k 10 1000 python tools/fuzz_parity.py --seconds ${CAMPAIGN_SECONDS:-300} --seed ${CAMPAIGN_SEED:-41}2 --force-opt 12=1 --out $OUT/campaign_dissolve.json > $OUT/campaign_dis.log 2>&1; tail -n 1 $OUT/campaign_dis.log | cut -c1-700

#!/bin/bash
# randomised campaigns on the final code of round 3: mixed options, and the child-order rules of any-hit walks forced (farthest first
# always; farthest first for rays ending outside the scene)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
S=${CAMPAIGN_SEED:-57}
timeout -k 10 400 python tools/fuzz_parity.py --seconds ${CAMPAIGN_SECONDS:-300} --seed ${S} --out $OUT/campaign2_mixed.json > $OUT/campaign2_mixed.log 2>&1; tail -n 1 $OUT/campaign2_mixed.log | cut -c1-700
timeout -k 10 400 python tools/fuzz_parity.py --seconds ${CAMPAIGN_SECONDS:-300} --seed ${S}1 --force-opt 8=3 --out $OUT/campaign2_far.json > $OUT/campaign2_far.log 2>&1; tail -n 1 $OUT/campaign2_far.log | cut -c1-700
timeout -k 10 400 python tools/fuzz_parity.py --seconds ${CAMPAIGN_SECONDS:-300} --seed ${S}2 --force-opt 8=5 --out $OUT/campaign2_outside.json > $OUT/campaign2_outside.log 2>&1; tail -n 1 $OUT/campaign2_outside.log | cut -c1-700

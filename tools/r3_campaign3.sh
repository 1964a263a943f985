#!/bin/bash
# campaigns with the non-default traversals forced next to the farthest-first order of any-hit walks: BVH2 nodes, the megakernel
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
S=${CAMPAIGN_SEED:-91}
timeout -k 10 400 python tools/fuzz_parity.py --seconds 300 --seed ${S} --force-opt 2=0 --force-opt 8=3 --out $OUT/campaign3_bvh2.json > $OUT/campaign3_bvh2.log 2>&1; tail -n 1 $OUT/campaign3_bvh2.log | cut -c1-400
timeout -k 10 400 python tools/fuzz_parity.py --seconds 300 --seed ${S}1 --force-opt 1=0 --force-opt 8=3 --out $OUT/campaign3_mega.json > $OUT/campaign3_mega.log 2>&1; tail -n 1 $OUT/campaign3_mega.log | cut -c1-400
timeout -k 10 400 python tools/fuzz_parity.py --seconds 300 --seed ${S}2 --force-opt 5=0 --force-opt 8=3 --out $OUT/campaign3_noshare.json > $OUT/campaign3_noshare.log 2>&1; tail -n 1 $OUT/campaign3_noshare.log | cut -c1-400

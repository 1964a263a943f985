#!/bin/bash
# the whole GPU test pass under process-wide hooks: every any-hit walk farthest first, the per-ray rule forced, BVH2 nodes, the
# megakernel (the child-order rule of round 3 lives in all three traversals)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
: > $OUT/modes.log
run() { local tag=$1; shift; env "$@" timeout -k 10 500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_configs.py -p no:cacheprovider > $OUT/modes_$tag.log 2>&1; echo "$tag ($*): $(tail -n 1 $OUT/modes_$tag.log)" | tee -a $OUT/modes.log; }
run far VKRT_WF_SHARE_FLAGS=3
run outside VKRT_WF_SHARE_FLAGS=5

#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
run() { local tag=$1; shift; env "$@" timeout -k 10 500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_configs.py -p no:cacheprovider > $OUT/modes_$tag.log 2>&1; echo "$tag ($*): $(tail -n 1 $OUT/modes_$tag.log)" | tee -a $OUT/modes.log; grep -E "^FAILED" $OUT/modes_$tag.log | cut -c1-200 | tee -a $OUT/modes.log; }
run sub1 VKRT_WF_SUBFRAMES=1
run sub2 VKRT_WF_SUBFRAMES=2

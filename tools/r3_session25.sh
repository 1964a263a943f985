#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== strong-scaling rehearsal by sub-frame count" | tee $OUT/s25.log
for sf in 1 2 3; do VKRT_WF_SUBFRAMES=$sf timeout -k 10 300 python tools/shard_probe.py 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('subframes $sf full_ms', d['full_ms'], {k:(v['ms'],v['efficiency']) for k,v in d['shards'].items()})" | tee -a $OUT/s25.log; done

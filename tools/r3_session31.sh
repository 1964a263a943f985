#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== sub-frames x hardware queues again, now with the interleaved enqueue" | tee $OUT/s31.log
for q in 4 8; do for sf in 3 4 6; do
  GPU_MAX_HW_QUEUES=$q VKRT_WF_SUBFRAMES=$sf timeout -k 10 300 python tools/shard_probe.py 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q subframes $sf full_ms', d['full_ms'], {k:(v['ms'],v['efficiency']) for k,v in d['shards'].items()})" | tee -a $OUT/s31.log
done; done
for q in 8; do for sf in 4 6; do GPU_MAX_HW_QUEUES=$q VKRT_WF_SUBFRAMES=$sf timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('1080p queues $q subframes $sf  Mrays/s %.1f ms/step %.2f'%(d['value'], d['ms_per_step']))" | tee -a $OUT/s31.log; done; done

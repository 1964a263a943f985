#!/bin/bash
# measurements that depend on the final sources, re-taken in one session: the option lines beside the headline, the tessellation
# block, the config matrix and the strong-scaling rehearsal
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
: > $OUT/refresh.log
for o in VKRT_NONE VKRT_WATERTIGHT VKRT_ANYHIT_DISSOLVE VKRT_SKIP_DEAD_SHADOW_RAYS; do
  env $o=1 timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | tail -n 1 > $OUT/final_$o.json || exit 1
  python -c "
import json; d=json.load(open('$OUT/final_$o.json')); r=d['roofline']
print('$o Mrays/s %.1f ms/step %.2f kernel_ms %.4f rays/frame %.1f M'%(d['value'], d['ms_per_step'], r['kernel_ms'], d['value']*d['ms_per_step']/1e3))" | tee -a $OUT/refresh.log
done
timeout -k 10 300 python bench.py --variant nonuniform --no-cpu-baseline 2>/dev/null | tail -n 1 > $OUT/bench_nonuniform.json && python -c "
import json; d=json.load(open('$OUT/bench_nonuniform.json')); print('nonuniform', d['value'], d['ms_per_step'], {k:v['Mrays_s'] for k,v in d['config']['builds'].items()})" | tee -a $OUT/refresh.log
timeout -k 10 600 python tools/config_matrix.py --only-tessellation --cpu-rows 12 --out $OUT/tess.json > $OUT/tess.log 2>&1; tail -n 4 $OUT/tess.log | cut -c1-1500 | tee -a $OUT/refresh.log
timeout -k 10 300 python tools/shard_probe.py 2>/dev/null | tail -n 1 | tee $OUT/shard_probe.jsonl | cut -c1-600 | tee -a $OUT/refresh.log
timeout -k 10 900 python tools/config_matrix.py --out $OUT/configs.json > $OUT/configs.log 2>&1; tail -n 2 $OUT/configs.log | cut -c1-300 | tee -a $OUT/refresh.log

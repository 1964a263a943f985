#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== watertight tests" | tee $OUT/s10.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "watertight or needle or sliver or dissolve" > $OUT/pytest_s10.log 2>&1; tail -n 4 $OUT/pytest_s10.log | tee -a $OUT/s10.log
echo "== cost" | tee -a $OUT/s10.log
for v in VKRT_NONE VKRT_WATERTIGHT; do env $v=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/s10_$v.json 2>/dev/null; python -c "
import json
d=json.loads(open('$OUT/s10_$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'Mrays/s %.1f ms/step %.2f kernel_ms %.4f'%(d['value'], d['ms_per_step'], r['kernel_ms']))
" | tee -a $OUT/s10.log; done
echo "== campaign, watertight forced, 4 minutes" | tee -a $OUT/s10.log
timeout -k 10 500 python tools/fuzz_parity.py --seconds 240 --seed 53 --force-opt 10=1 --out $OUT/campaign_watertight2.json > $OUT/campaign_wt2.log 2>&1; tail -n 1 $OUT/campaign_wt2.log | cut -c1-700 | tee -a $OUT/s10.log

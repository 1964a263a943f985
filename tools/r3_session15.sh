#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== any-hit order: K front-to-back steps, then farthest first (flags 3)" | tee $OUT/s15.log
for variant in default nonuniform; do for k in 0 2 4 8 16; do VKRT_ANYHIT_NEAR_STEPS=$k VKRT_WF_SHARE_FLAGS=3 timeout -k 10 300 python bench.py --variant $variant --steps 3 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant K=$k  Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested']))" | tee -a $OUT/s15.log; done; done

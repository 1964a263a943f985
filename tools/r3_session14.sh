#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== any-hit child order: flags 1 = front to back, 3 = far first, 5 = far first when the ray ends outside the scene bounds" | tee $OUT/s14.log
for variant in default nonuniform; do for f in 1 3 5; do VKRT_WF_SHARE_FLAGS=$f timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant flags $f  Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested']))" | tee -a $OUT/s14.log; done; done

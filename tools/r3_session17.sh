#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== any-hit child order, automatic rule (flags 9 = default) against 1 (front to back) and 5 (per-ray rule forced)" | tee $OUT/s17.log
for variant in default nonuniform; do for f in 9 1; do VKRT_WF_SHARE_FLAGS=$f timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant flags $f  Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f build_ms %s'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested'], d.get('config',{}).get('build_ms')))" | tee -a $OUT/s17.log; done; done

#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== #98 shadow-ray occluder cache (bits = log2 buckets, cells = key grid cells along the scene diagonal)" | tee $OUT/s21.log
for variant in default nonuniform; do for cfg in "22 256" "22 512" "24 1024" "20 256"; do set -- $cfg; VKRT_OCCLUDER_CACHE_BITS=$1 VKRT_OCCLUDER_CACHE_CELLS=$2 timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant bits $1 cells $2  Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested']))" | tee -a $OUT/s21.log; done; done

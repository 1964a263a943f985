#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== #101 cost of a triangle test relative to a node visit in the wide collapse (product 1.0; exp14 0.3, exp12 0.6, exp13 1.5)" | tee $OUT/s24.log
for variant in default nonuniform; do for e in none 14 12 13; do lib=$R/vk-raytracing-engine_amd/libvkrt_exp$e.so; [ $e = none ] && lib=""; VKRT_LIB=$lib timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; b=d['config']['builds']['ploc']
print('$variant exp $e  Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f nodes %d sah %.2f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested'], b['nodes'], b['sah_cost']))" | tee -a $OUT/s24.log; done; done

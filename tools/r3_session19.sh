#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
L8=$R/vk-raytracing-engine_amd/libvkrt_exp8.so; L9=$R/vk-raytracing-engine_amd/libvkrt_exp9.so
echo "== #97 the root and its children read from a per-wave LDS copy (exp8)" | tee $OUT/s19.log
for lib in "" $L8; do VKRT_LIB=$lib BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s19.log; done
for variant in default nonuniform; do for lib in "" $L8 $L9; do VKRT_LIB=$lib timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant lib=$(basename "$lib")  Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested']))" | tee -a $OUT/s19.log; done; done

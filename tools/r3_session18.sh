#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
L6=$R/vk-raytracing-engine_amd/libvkrt_exp6.so; L7=$R/vk-raytracing-engine_amd/libvkrt_exp7.so
echo "== #95 128-B wide nodes with binary16 planes (exp6: 8-bit grid, exp7: 11-bit grid)" | tee $OUT/s18.log
for lib in "" $L6 $L7; do VKRT_LIB=$lib BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s18.log; done
for lib in "" $L7; do VKRT_LIB=$lib BUILD=sah timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s18.log; done
for variant in default nonuniform; do for lib in "" $L6 $L7; do VKRT_LIB=$lib timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant lib=$(basename "$lib")  Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f instr/ray %.1f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested'], r['achieved']*r['kernel_ms']*1e6/r['rays_per_launch']))" | tee -a $OUT/s18.log; done; done

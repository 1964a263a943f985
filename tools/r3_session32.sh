#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
L=$R/vk-raytracing-engine_amd/libvkrt_exp23.so
echo "== #107 non-temporal stores of the state planes S0..S3 only (exp23)" | tee $OUT/s32.log
VKRT_LIB=$L BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s32.log
for variant in default nonuniform; do for lib in "" $L; do VKRT_LIB=$lib timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant lib=$(basename "$lib")  Mrays/s %.1f ms/step %.2f kernel_ms %.4f'%(d['value'], d['ms_per_step'], r['kernel_ms']))" | tee -a $OUT/s32.log; done; done

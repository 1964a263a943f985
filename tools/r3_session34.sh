#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== #108 workgroup size of the shade / init kernels (product 256; exp24 128, exp25 512)" | tee $OUT/s34.log
for e in 24 25; do VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp$e.so BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s34.log; done
for e in none 24 25; do lib=$R/vk-raytracing-engine_amd/libvkrt_exp$e.so; [ $e = none ] && lib=""; VKRT_LIB=$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('exp $e  Mrays/s %.1f ms/step %.2f kernel_ms %.4f'%(d['value'], d['ms_per_step'], r['kernel_ms']))" | tee -a $OUT/s34.log; done

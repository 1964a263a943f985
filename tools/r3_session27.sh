#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== #103 adopted (non-temporal stream loads = product) and #104 (exp18: non-temporal triangle fetches)" | tee $OUT/s28.log
for e in 18; do VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp$e.so BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s28.log; done
for variant in default nonuniform; do for e in none 18; do lib=$R/vk-raytracing-engine_amd/libvkrt_exp$e.so; [ $e = none ] && lib=""; VKRT_LIB=$lib timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant exp $e  Mrays/s %.1f ms/step %.2f kernel_ms %.4f'%(d['value'], d['ms_per_step'], r['kernel_ms']))" | tee -a $OUT/s28.log; done; done

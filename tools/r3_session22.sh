#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
L=$R/vk-raytracing-engine_amd/libvkrt_exp10.so
echo "== #99 six waves per SIMD for the traversal kernel (exp10: 80 VGPRs; stack words per lane 24 = product, 22, 20 -> LDS 7680 / 7168 / 6656 B per wave)" | tee $OUT/s22.log
VKRT_LIB=$L VKRT_STACK_CAP_WORDS=20 BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s22.log
for variant in default nonuniform; do for cfg in "none 24" "$L 24" "$L 22" "$L 20"; do set -- $cfg; lib=$1; [ $lib = none ] && lib=""; VKRT_LIB=$lib VKRT_STACK_CAP_WORDS=$2 timeout -k 10 300 python bench.py --variant $variant --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$variant lib=$(basename $1) stack words $2  Mrays/s %.1f ms/step %.2f kernel_ms %.4f faults %s'%(d['value'], d['ms_per_step'], r['kernel_ms'], d['config'].get('traversal_faults')))" | tee -a $OUT/s22.log; done; done

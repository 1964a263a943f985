#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
L=$R/vk-raytracing-engine_amd/libvkrt_exp11.so
echo "== #100 traversal workgroups padded in LDS (pad bytes -> workgroups per CU: 0 -> 20 by VGPRs, 512 -> 20, 2560 -> 16, 5888 -> 12): room for shade waves of other sub-frames" | tee $OUT/s23.log
for sf in 3 2 4; do for pad in 0 2560 5888; do VKRT_LIB=$L VKRT_TRAV_LDS_PAD=$pad VKRT_WF_SUBFRAMES=$sf timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('subframes $sf pad $pad  Mrays/s %.1f ms/step %.2f kernel_ms %.4f'%(d['value'], d['ms_per_step'], r['kernel_ms']))" | tee -a $OUT/s23.log; done; done

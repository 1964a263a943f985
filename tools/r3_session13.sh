#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== far-first any-hit walks as a runtime flag (VKRT_WF_SHARE_FLAGS 1 = off, 3 = on)" | tee $OUT/s13.log
for f in 1 3; do VKRT_WF_SHARE_FLAGS=$f BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s13.log; done
BUILD=ploc bash tools/probe_variants.sh "VKRT_WF_SHARE_FLAGS=1" "VKRT_WF_SHARE_FLAGS=3" "VKRT_WF_SHARE_FLAGS=1" "VKRT_WF_SHARE_FLAGS=3" 2>&1 | tee -a $OUT/s13.log
for f in 1 3; do VKRT_WF_SHARE_FLAGS=$f timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('flags $f bench Mrays/s %.1f ms/step %.2f kernel_ms %.4f nodes/ray %.2f tris/ray %.2f'%(d['value'], d['ms_per_step'], r['kernel_ms'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested']))" | tee -a $OUT/s13.log; done
VKRT_WF_SHARE_FLAGS=3 timeout -k 10 300 python bench.py --variant nonuniform --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('nonuniform flags 3 Mrays/s %.1f ms/step %.2f nodes/ray %.2f tris/ray %.2f'%(d['value'], d['ms_per_step'], r['per_ray']['nodes_visited'], r['per_ray']['tris_tested']))" | tee -a $OUT/s13.log
echo "== GPU suite" | tee -a $OUT/s13.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/pytest_s13.log 2>&1; tail -n 4 $OUT/pytest_s13.log | tee -a $OUT/s13.log

#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
L=$R/vk-raytracing-engine_amd/libvkrt_exp22.so
echo "== #106 rounds of the sub-frames enqueued interleaved (exp22)" | tee $OUT/s30.log
VKRT_LIB=$L BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s30.log
for sf in 3 2 4; do for lib in "" $L; do VKRT_LIB=$lib VKRT_WF_SUBFRAMES=$sf timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('subframes $sf lib=$(basename "$lib")  Mrays/s %.1f ms/step %.2f'%(d['value'], d['ms_per_step']))" | tee -a $OUT/s30.log; done; done
for lib in "" $L; do VKRT_LIB=$lib timeout -k 10 300 python tools/shard_probe.py 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shards lib=$(basename "$lib") full_ms', d['full_ms'], {k:(v['ms'],v['efficiency']) for k,v in d['shards'].items()})" | tee -a $OUT/s30.log; done

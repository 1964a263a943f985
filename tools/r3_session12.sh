#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== #94 any-hit walks farthest child first" | tee $OUT/s12.log
for lib in "" "$R/vk-raytracing-engine_amd/libvkrt_exp6.so"; do VKRT_LIB=$lib BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s12.log; done
BUILD=ploc bash tools/probe_variants.sh "" "VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp6.so" "" 2>&1 | tee -a $OUT/s12.log

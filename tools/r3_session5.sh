#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== shade-stage experiments: default / #87 doubled compaction atomics / #88 four waves per SIMD (serial timing pass: rest = init + shade)" | tee $OUT/s5.log
BUILD=ploc bash tools/probe_variants.sh "" "VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp3.so" "VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp4.so" "" 2>&1 | tee -a $OUT/s5.log

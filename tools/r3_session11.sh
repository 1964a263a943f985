#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== shade kernel register footprint vs overlap with the traversal (3 sub-frames): default 142 VGPRs / #88 128 / #93 96" | tee $OUT/s11.log
BUILD=ploc bash tools/probe_variants.sh "" "VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp4.so" "VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp5.so" "" 2>&1 | tee -a $OUT/s11.log

#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== tests" | tee $OUT/s4.log
timeout -k 10 1200 python -m pytest tests/test_host_layer.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "watertight or skipping or hybrid_frames_in_strips or rank_launcher or bench_runs or nonuniform" > $OUT/pytest_s4.log 2>&1; tail -n 8 $OUT/pytest_s4.log | tee -a $OUT/s4.log
echo "== traversal experiments: default / #80 two triangle tests per step / #81 triangle hand-off" | tee -a $OUT/s4.log
for lib in "" "$R/vk-raytracing-engine_amd/libvkrt_exp1.so" "$R/vk-raytracing-engine_amd/libvkrt_exp2.so"; do
  VKRT_LIB=$lib BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s4.log
done
BUILD=ploc bash tools/probe_variants.sh "" "VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp1.so" "VKRT_LIB=$R/vk-raytracing-engine_amd/libvkrt_exp2.so" "" 2>&1 | tee -a $OUT/s4.log
echo "== uniform vs Sponza-like tessellation, three builders" | tee -a $OUT/s4.log
timeout -k 10 900 python tools/config_matrix.py --only-tessellation --cpu-rows 12 --out $OUT/tess.json > $OUT/tess.log 2>&1; tail -n 4 $OUT/tess.log | cut -c1-1500 | tee -a $OUT/s4.log

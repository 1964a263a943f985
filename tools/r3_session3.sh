#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== new parity tests" | tee $OUT/s3.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_hybrid.py tests/test_host_layer.py tests/test_sharding.py tests/test_texture_lod.py tests/test_second_restatement.py tests/test_abi_library.py -x -q -m gpu -k "not (closest_hit_rays or any_hit_rays or config1)" > $OUT/pytest_s3.log 2>&1; tail -n 8 $OUT/pytest_s3.log | tee -a $OUT/s3.log
echo "== frames in flight on a 4K/8 shard (handles:subframes -> ms/frame), 8 hardware queues" | tee -a $OUT/s3.log
GPU_MAX_HW_QUEUES=8 PROBE_W=3840 PROBE_H=2160 PROBE_SPP=16 PROBE_FRAMES=6 PROBE_SHARDS=8 PROBE_CONFIGS="1:1,1:3,2:1,2:2,2:3,3:1,3:2" timeout -k 10 600 python tools/overlap_probe.py 2>/dev/null | tail -n 1 | tee -a $OUT/s3.log
echo "== the same at 1080p whole frame" | tee -a $OUT/s3.log
GPU_MAX_HW_QUEUES=8 PROBE_SPP=16 PROBE_FRAMES=6 PROBE_CONFIGS="1:3,2:1,2:2,2:3,3:1" timeout -k 10 600 python tools/overlap_probe.py 2>/dev/null | tail -n 1 | tee -a $OUT/s3.log

#!/bin/bash
# round-3 GPU session 1: new parity tests, A/B of the new options, launch anatomy of a 4K/8 shard
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== replay of seed 1301004260 (MT with padded boxes, then watertight)" | tee $OUT/s1.log
timeout -k 10 300 python tools/debug_case.py 1301004260 > $OUT/debug_case_mt.log 2>&1; tail -n 3 $OUT/debug_case_mt.log | cut -c1-600 | tee -a $OUT/s1.log
timeout -k 10 300 python tools/debug_case.py 1301004260 --watertight > $OUT/debug_case_wt.log 2>&1; tail -n 3 $OUT/debug_case_wt.log | cut -c1-600 | tee -a $OUT/s1.log
echo "== pytest fuzz + new parity tests" | tee -a $OUT/s1.log
timeout -k 10 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_s1.log 2>&1; tail -n 15 $OUT/pytest_s1.log | tee -a $OUT/s1.log
echo "== bench A/B (no cpu baseline)" | tee -a $OUT/s1.log
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/bench_default.json 2> $OUT/bench_default.err
VKRT_SKIP_DEAD_SHADOW_RAYS=1 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/bench_skipdead.json 2> $OUT/bench_skipdead.err
VKRT_WATERTIGHT=1 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/bench_watertight.json 2> $OUT/bench_watertight.err
python - <<'PY' | tee -a gpurun_out/r3/s1.log
import json
for k in ("default","skipdead","watertight"):
    try:
        d=json.loads(open(f"gpurun_out/r3/bench_{k}.json").read().strip().splitlines()[-1])
        r=d["roofline"]
        print(k, "Mrays/s %.1f ms/step %.2f rays/step %.4g kernel_ms %.4f nodes/ray %.2f tris/ray %.2f"%(d["value"], d["ms_per_step"], d["config"]["rays_per_step"], r["kernel_ms"], r["per_ray"]["nodes_visited"], r["per_ray"]["tris_tested"]))
    except Exception as e:
        print(k, "failed", e)
PY

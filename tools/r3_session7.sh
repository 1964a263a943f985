#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== paired texel loads: hash (product c95e8b8e1cb46dcf), probe (before: frame 16.01-16.13 ms, rest 4.80-4.82 ms)" | tee $OUT/s7.log
BUILD=ploc timeout -k 10 300 python tools/variant_hash.py 1920 1080 4 8 2 2>/dev/null | grep HASH | tee -a $OUT/s7.log
BUILD=ploc bash tools/probe_variants.sh "" "" 2>&1 | tee -a $OUT/s7.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_texture_lod.py tests/test_hybrid.py -x -q -m gpu -k "atrium or texture or lod or emissive or hybrid" > $OUT/pytest_s7.log 2>&1; tail -n 5 $OUT/pytest_s7.log | tee -a $OUT/s7.log
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/bench_s7.json 2> $OUT/bench_s7.err
python - <<'PY' | tee -a gpurun_out/r3/s7.log
import json
d=json.loads(open("gpurun_out/r3/bench_s7.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("bench Mrays/s %.1f ms/step %.2f kernel_ms %.4f frame(serial) %.2f traverse %.2f"%(d["value"], d["ms_per_step"], r["kernel_ms"], r["frame"]["ms"], r["frame"]["traverse_ms"]))
PY

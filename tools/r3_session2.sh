#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r3
mkdir -p $OUT
echo "== full GPU suite" | tee $OUT/s2.log
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_s2.log 2>&1; tail -n 8 $OUT/pytest_s2.log | tee -a $OUT/s2.log
echo "== bench default / watertight" | tee -a $OUT/s2.log
timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/bench_default2.json 2> $OUT/bench_default2.err
VKRT_WATERTIGHT=1 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/bench_watertight2.json 2> $OUT/bench_watertight2.err
python - <<'PY' | tee -a gpurun_out/r3/s2.log
import json
for k in ("default2","watertight2"):
    try:
        d=json.loads(open(f"gpurun_out/r3/bench_{k}.json").read().strip().splitlines()[-1])
        r=d["roofline"]
        print(k, "Mrays/s %.1f ms/step %.2f rays/step %.4g kernel_ms %.4f nodes/ray %.2f tris/ray %.2f"%(d["value"], d["ms_per_step"], d["config"]["rays_per_step"], r["kernel_ms"], r["per_ray"]["nodes_visited"], r["per_ray"]["tris_tested"]))
    except Exception as e:
        print(k, "failed", e)
PY
echo "== shard anatomy (kernel trace, 1 and 3 sub-frames)" | tee -a $OUT/s2.log
(cd /tmp && VKRT_WF_SUBFRAMES=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_sf1 -o shard -- python3 $GRAFT_REPO_ROOT/tools/shard_trace.py) 2>&1 | grep "^frame" | tee -a $OUT/s2.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_sf3 -o shard -- python3 $GRAFT_REPO_ROOT/tools/shard_trace.py) 2>&1 | grep "^frame" | tee -a $OUT/s2.log
echo "== hardware queues x sub-frames (shard probe)" | tee -a $OUT/s2.log
for q in 4 8; do for sf in 3 4 6; do
  GPU_MAX_HW_QUEUES=$q VKRT_WF_SUBFRAMES=$sf timeout -k 10 300 python tools/shard_probe.py 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q subframes $sf full_ms', d['full_ms'], {k:(v['ms'],v['efficiency']) for k,v in d['shards'].items()})" | tee -a $OUT/s2.log
done; done
ls $OUT/trace_sf1 $OUT/trace_sf3 | head

#!/bin/bash
# One gpurun call: GPU tests -> smoke -> bench -> rocprofv3 kernel trace.  Stops after a timeout.
set -o pipefail
mkdir -p gpurun_out
run() { # name, seconds, cmd...
  local name=$1 secs=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name exit $rc" | tee -a gpurun_out/round.log
  if [ $rc -ge 124 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/round.log; exit $rc; fi
  return 0
}
: > gpurun_out/round.log
run pytest_gpu 600 python -m pytest tests -m gpu -q
run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
run bench 900 python bench.py ${BENCH_ARGS:---steps 3 --warmup 1}
export TMPDIR=/tmp
( cd /tmp && run_dir=$GRAFT_REPO_ROOT/gpurun_out/prof && rm -rf $run_dir && \
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $run_dir -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/round.log )
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log; tail -3 gpurun_out/rocprof.log

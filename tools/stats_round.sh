#!/bin/bash
# rocprofv3 kernel-trace stats of the bench workload (one gpurun call). Usage: stats_round.sh <tag> [bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT; TAG=${1:-run}; shift
OUT=$R/gpurun_out/stats_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
echo "rc=$?"
tail -1 $OUT/bench.log | cut -c1-200
cat $OUT/*/*kernel_stats.csv | cut -c1-160 | head -12

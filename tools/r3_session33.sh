#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3
mkdir -p $OUT
cd /tmp
export VKRT_WF_SUBFRAMES=1
for t in tex notex; do
  flag=""; [ $t = notex ] && flag="--no-textures"
  rm -rf $OUT/stats_$t
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$t -- python3 $R/bench.py $flag --steps 2 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/stats_$t.log 2>&1 || exit 1
  echo "== $t"; tail -n 1 $OUT/stats_$t.log | cut -c1-120; cat $OUT/stats_$t/*/*kernel_stats.csv | cut -c1-130 | head -4
done 2>&1 | tee $OUT/s33.log

#!/bin/bash
# One gpurun call producing the artefacts kept under profiles/: GPU tests, default bench line, rocprofv3
# kernel-trace stats of the same command, and separate --pmc passes for FETCH_SIZE / WRITE_SIZE.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; TAG=${1:-r04}
rm -rf $OUT; mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > $OUT/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a $OUT/round.log; [ $rc -ge 124 ] && { echo TIMEOUT | tee -a $OUT/round.log; exit $rc; }; return 0; }
: > $OUT/round.log
step pytest 900 python -m pytest tests -m gpu -q
export TMPDIR=/tmp
cd /tmp
# per-kernel durations that bench.py's roofline must agree with are those of un-overlapped launches: one lane
export VKRT_WF_SUBFRAMES=1 VKRT_WF_FRAMES_IN_FLIGHT=1
step stats 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-cpu-baseline --no-other-builder
step pmc_valu 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu -- python3 $R/bench.py --no-cpu-baseline --no-other-builder
step pmc_fetch 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-other-builder
step pmc_write 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-other-builder
unset VKRT_WF_SUBFRAMES VKRT_WF_FRAMES_IN_FLIGHT
# the default configuration (frames in flight on internal streams: kernels overlap, durations are not exclusive)
step stats_pipelined 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipelined -- python3 $R/bench.py --no-cpu-baseline --no-other-builder
cd $R
# the bench line quotes profiles/pmc_traffic.json and pmc_issue.json: derive them from the passes above first (the same
# collector runs again on the workstation over the merged gpurun_out/), then take the line
python tools/collect_profiles.py $TAG --no-bench > $OUT/collect.log 2>&1 || echo "collect failed" | tee -a $OUT/round.log
step bench 600 python bench.py
tail -3 $OUT/pytest.log; tail -1 $OUT/bench.log | cut -c1-400; cat $OUT/stats/*/*kernel_stats.csv | cut -c1-150 | head -6

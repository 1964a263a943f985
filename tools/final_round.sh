#!/bin/bash
# The artefacts kept under profiles/, in parts that each fit one gpurun call (<= 20 min):
#   tools/final_round.sh a   GPU tests, rocprofv3 kernel-trace stats of bench.py (one lane: exclusive durations), --pmc VALU issue pass
#   tools/final_round.sh b   --pmc FETCH_SIZE / WRITE_SIZE passes, kernel-trace stats of the default (pipelined) configuration,
#                            per-kernel SQ / TCP / TCC counter passes (tools/pmc_summary.py), kernel stats of the hybrid frame
#   (on the workstation: python tools/collect_profiles.py rNN --no-bench; commit)
#   tools/final_round.sh c   the bench line on the committed summaries (pmc_stale false)
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/final; PART=${1:-a}
mkdir -p $OUT
step() { local name=$1 secs=$2; shift 2; timeout -k 10 $secs "$@" > $OUT/$name.log 2>&1; local rc=$?; echo "$name rc=$rc" | tee -a $OUT/round_$PART.log; [ $rc -ge 124 ] && { echo TIMEOUT | tee -a $OUT/round_$PART.log; exit $rc; }; return 0; }
: > $OUT/round_$PART.log
export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-other-builder"
if [ $PART = a ]; then
  rm -rf $OUT/stats $OUT/pmc_valu
  step pytest 900 python -m pytest $R/tests -m gpu -q -p no:cacheprovider
  cd /tmp
  # per-kernel durations that bench.py's roofline must agree with are those of un-overlapped launches: one lane
  export VKRT_WF_SUBFRAMES=1 VKRT_WF_FRAMES_IN_FLIGHT=1
  step stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B
  step pmc_valu 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu -- $B
  tail -3 $OUT/pytest.log; cat $OUT/stats/*/*kernel_stats.csv | cut -c1-150 | head -6
elif [ $PART = b ]; then
  rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/stats_pipelined $OUT/stats_hybrid $R/gpurun_out/pmc
  cd /tmp
  export VKRT_WF_SUBFRAMES=1 VKRT_WF_FRAMES_IN_FLIGHT=1
  step pmc_fetch 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B
  step pmc_write 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B
  mkdir -p $R/gpurun_out/pmc
  i=0
  while read -r SET; do
    [ -z "$SET" ] && continue
    i=$((i+1))
    step pmc_set$i 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc/set$i -- $B --steps 2 --warmup 1
  done <<SETS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum
TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum
TCC_MISS_sum
SETS
  unset VKRT_WF_SUBFRAMES VKRT_WF_FRAMES_IN_FLIGHT
  # the default configuration (frames in flight on internal streams: kernels overlap, durations are not exclusive)
  step stats_pipelined 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pipelined -- $B
  # the hybrid frame (BASELINE config 5 stand-in): G-buffer ray cast, shadows + AO + GI depth 8, post -- 6 frames
  step stats_hybrid 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_hybrid -- python3 $R/tools/config_matrix.py --only-hybrid-frames 6
  cat $OUT/round_b.log; cat $OUT/stats_hybrid/*/*kernel_stats.csv | cut -c1-150 | head -12
else
  cd $R
  step bench 900 python bench.py ${BENCH_ARGS:---steps 20 --warmup 5}
  tail -1 $OUT/bench.log | cut -c1-600
fi

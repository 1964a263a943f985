#!/bin/bash
# One gpurun call: PMC counter passes on the bench workload, one SMALL counter set per pass (kernel-trace only).
#
# gfx950 per-pass slots (MI355X_MICROARCH.md "rocprofv3 PMC slots"): SQ 8, TCC 4 (FETCH_SIZE costs 3, WRITE_SIZE 2), GRBM 2.
# A set that does not fit makes rocprofv3 abort inside the first HIP call of the program with
#   "Could not construct profile cfg ... error code 38: Request exceeds the capabilities of the hardware to collect"
# (signal 6 caught by its handler; the process then idles).  Round 1 ran a 4-counter TCC set and a 4-counter TCP set into that
# and waited out two 400-s timeouts (profiles/r01_experiments.md #25).  Here every derived TCC / TCP counter gets its own
# pass, and each pass is watched: the moment its log shows the abort, it is killed and the script stops with an error.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT/pmc
: > $OUT/profile_round.log
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-builder ${BENCH_EXTRA}"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/pmc/counters_list.txt 2>&1 || true
# run one pass in the background and watch its log for the profile-config abort
pass() { # index, counters...
  local i=$1; shift
  echo "=== pmc set $i: $*" >> $OUT/profile_round.log
  rm -rf $OUT/pmc/set$i
  # the pass runs in a session (= process group) of its own: a background job of a non-interactive script otherwise shares the
  # script's group, and killing "the pass's group" would take this script -- and whatever launched it -- down with it
  setsid timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc/set$i -- $B > $OUT/pmc/set$i.log 2>&1 &
  local pid=$!
  while kill -0 $pid 2>/dev/null; do
    if grep -qE "error code 38|Could not construct profile cfg" $OUT/pmc/set$i.log 2>/dev/null; then
      echo "pmc set $i: counter set does not fit the hardware (error 38): killing the pass and stopping" | tee -a $OUT/profile_round.log
      local pg=$(ps -o pgid= $pid | tr -d ' ')
      if [ -n "$pg" ] && [ "$pg" != "$(ps -o pgid= $$ | tr -d ' ')" ]; then kill -TERM -- -$pg 2>/dev/null; sleep 2; kill -KILL -- -$pg 2>/dev/null; else kill $pid 2>/dev/null; fi
      wait $pid 2>/dev/null
      exit 38
    fi
    sleep 2
  done
  wait $pid
  local rc=$?
  echo "pmc set $i rc=$rc" >> $OUT/profile_round.log
  if [ $rc -ge 124 ]; then echo "TIMEOUT: stop" >> $OUT/profile_round.log; exit $rc; fi
}
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  pass $i $SET
done <<SETS
${PMC_SETS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum
TCC_MISS_sum
TCC_REQ_sum
TCC_EA0_RDREQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum
TCP_TCC_READ_REQ_sum
TCP_PENDING_STALL_CYCLES_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum}
SETS
grep -E "===|rc=" $OUT/profile_round.log | cut -c1-200

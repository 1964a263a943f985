#!/bin/bash
# One gpurun call: occupancy variants + PMC counter passes on the bench workload.
# Counters are collected in their own runs (kernel-trace only), one small set per pass.
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
mkdir -p $OUT/pmc
: > $OUT/profile_round.log
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_EXTRA}"
for v in ${VARIANTS:-0 3 4}; do
  echo "=== variant VKRT_MINWAVES=$v" >> $OUT/profile_round.log
  VKRT_MINWAVES=$v timeout -k 10 300 $B >> $OUT/profile_round.log 2>&1 || { echo "variant $v failed/timeout rc=$?" >> $OUT/profile_round.log; exit 1; }
done
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/pmc/counters_list.txt 2>&1 || true
i=0
while read -r SET; do
  [ -z "$SET" ] && continue
  i=$((i+1))
  echo "=== pmc set $i: $SET" >> $OUT/profile_round.log
  rm -rf $OUT/pmc/set$i
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc/set$i -- $B > $OUT/pmc/set$i.log 2>&1
  rc=$?
  echo "pmc set $i rc=$rc" >> $OUT/profile_round.log
  if [ $rc -ge 124 ]; then echo "TIMEOUT: stop" >> $OUT/profile_round.log; exit $rc; fi
done <<SETS
${PMC_SETS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
FETCH_SIZE
WRITE_SIZE
SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum}
SETS
cat $OUT/profile_round.log | grep -E "===|value|rc=" | cut -c1-400

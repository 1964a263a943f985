#!/bin/bash
# round-3 closing session: final_round (tests, kernel stats, PMC for the bench line, bench) + the SQ / TCP / TCC counter sets of every
# wavefront kernel (profiles/r03_pmc_counters.json; the shade stage's numbers are what DESIGN.md section 5 argues from)
set -o pipefail
bash tools/final_round.sh r03 || exit $?
export VKRT_WF_SUBFRAMES=1
export PMC_SETS="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
TCP_TOTAL_CACHE_ACCESSES_sum
TCP_PENDING_STALL_CYCLES_sum
TCC_HIT_sum
TCC_MISS_sum"
bash tools/profile_round.sh
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py > gpurun_out/pmc/summary.json && head -c 600 gpurun_out/pmc/summary.json

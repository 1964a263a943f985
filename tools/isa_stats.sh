#!/bin/bash
# Device assembly of the traversal kernel's translation unit (csrc/wf_traverse.hip, with the flags csrc/Makefile gives it and
# optional extra flags) and the register / instruction statistics of the product instantiations:
#   tools/isa_stats.sh [name "-DFOO=1"] -> /tmp/vkrt_isa/wf_traverse[_name].s
set -e
NAME=$1; DEFS=$2; OUT=/tmp/vkrt_isa; mkdir -p $OUT
S=$OUT/wf_traverse${NAME:+_$NAME}.s
cd "$(dirname "$0")/../vk-raytracing-engine_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=max-memory-clause -fno-slp-vectorize \
  $DEFS --cuda-device-only -S -o $S wf_traverse.hip 2>/dev/null
for k in _Z13k_wf_traverseILb0ELb1ELi64ELi0EEv11TraceParams9WfBuffersi _Z13k_wf_traverseILb1ELb1ELi64ELi0EEv11TraceParams9WfBuffersi; do
  awk -v k="$k" '$0 ~ "^"k":" {on=1} on {print} on && /^\.Lfunc_end/ {exit}' $S > $OUT/body.s
  echo "$k: instr $(grep -c -E '^\s+(v_|s_|global_|ds_|buffer_|scratch_)' $OUT/body.s) valu $(grep -c -E '^\s+v_' $OUT/body.s) v_mov $(grep -c -E '^\s+v_mov_b32' $OUT/body.s) v_pk $(grep -c v_pk_ $OUT/body.s) cvt_ubyte $(grep -c v_cvt_f32_ubyte $OUT/body.s) cndmask $(grep -c v_cndmask $OUT/body.s) sdwa $(grep -c _sdwa $OUT/body.s) loadx4 $(grep -c global_load_dwordx4 $OUT/body.s)"
  grep -A40 "^	.amdhsa_kernel $k" $S | grep -E "next_free_vgpr|private_segment_fixed_size|accum_offset" | tr -s ' \t' ' ' | tr '\n' ';'; echo
done

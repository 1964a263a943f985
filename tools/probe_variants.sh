#!/bin/bash
# One gpurun call: wf_probe (4 spp, depth 8, 1080p atrium) under a list of environment variants, one JSON line each.
# usage: tools/probe_variants.sh "VAR=1 VAR2=3" "VAR=2" ...   (an empty string = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/probe_variants.log; : > $OUT
export PROBE_QUICK=1
for v in "$@"; do
  echo "=== $v" >> $OUT
  env $v timeout -k 10 200 python $R/tools/wf_probe.py 2>/dev/null | grep '^{' >> $OUT || { echo "variant failed: $v" >> $OUT; }
done
python - <<PY
import json
for l in open("$OUT"):
    if l.startswith("==="): print(l.strip()); continue
    if not l.startswith("{"): print(l.strip()); continue
    d=json.loads(l); print("   frame %.3f ms (%s Mrays/s)  serial: traverse %.3f + rest %.3f ms  nodes/ray %.2f tris/ray %.2f  node_eff %.3f tri_eff %.3f steps %d/%d"%(d["frame_ms"],d["Mrays_s_frame"],d["traverse_ms"],d["total_ms"]-d["traverse_ms"],d["nodes_per_ray"],d["tris_per_ray"],d["node_lane_eff"],d["tri_lane_eff"],d["wave_node_steps"],d["wave_tri_steps"]))
PY

#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r3
mkdir -p $OUT
cd /tmp
for sf in 3 4; do
  export VKRT_WF_SUBFRAMES=$sf
  rm -rf $OUT/trace_sub$sf
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_sub$sf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-builder > $OUT/trace_sub$sf.log 2>&1 || exit 1
  tail -n 1 $OUT/trace_sub$sf.log | cut -c1-200
done
cd $R
python - <<'PY'
import csv, glob, collections
for sf in (3, 4):
    f = glob.glob(f'gpurun_out/r3/trace_sub{sf}/*/*kernel_trace.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith(('void k_wf', 'k_wf'))]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # last frame only: take the last third of the launches
    n = len(rows); rows = rows[n * 2 // 3:]
    t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
    q = collections.Counter(r['Queue_Id'] for r in rows)
    ev = []
    for r in rows: ev += [(int(r['Start_Timestamp']), 1), (int(r['End_Timestamp']), -1)]
    ev.sort(); cur = 0; last = ev[0][0]; hist = collections.Counter()
    for t, d in ev:
        hist[cur] += t - last; last = t; cur += d
    tot = sum(hist.values())
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    print(f'subframes {sf}: launches {len(rows)} span {(t1 - t0) / 1e6:.2f} ms, sum of kernel durations {busy / 1e6:.2f} ms, queues {dict(q)}, time by concurrency ' + ', '.join(f'{k}: {v / tot:.2f}' for k, v in sorted(hist.items())))
    trav = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if 'traverse' in r['Kernel_Name']]
    shade = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if 'shade' in r['Kernel_Name']]
    print(f'   traverse mean {sum(trav) / len(trav) / 1e3:.1f} us x {len(trav)}, shade mean {sum(shade) / len(shade) / 1e3:.1f} us x {len(shade)}')
PY

"""Per-kernel sums of every counter found under gpurun_out/pmc/set*/ (tools/profile_round.sh) -> JSON on stdout."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc")
out = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
files = {}
for f in glob.glob(os.path.join(base, "set*", "*", "*counter_collection.csv")):
    d = os.path.dirname(f)  # gpurun merges every session into the same directories: keep the newest pass of each set only
    if d not in files or os.path.getmtime(f) > os.path.getmtime(files[d]):
        files[d] = f
for f in sorted(files.values()):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if not k.startswith(("k_wf", "k_pathtrace", "k_hybrid", "k_gbuffer")):
            continue
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
res = {k: {c: {"per_launch": v / max(calls[k][c], 1), "launches": calls[k][c]} for c, v in cs.items()} for k, cs in out.items()}
json.dump(res, sys.stdout, indent=1)

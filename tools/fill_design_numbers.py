"""Fills the round-4 result placeholders of DESIGN.md (R4_*) from profiles/r04_bench.json and profiles/r04_kernel_stats.csv."""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
stats = {r["Name"].split("(")[0].strip(): float(r["AverageNs"]) for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r04_kernel_stats.csv")))}
shade = next(v for k, v in stats.items() if k == "k_wf_shade")
nu = b["config"].get("nonuniform_variant", {})
vals = {"R4_MRAYS": f"{b['value'] / 1e3:.2f}", "R4_MS": f"{b['ms_per_step']:.1f}", "R4_TRAV": f"{b['roofline']['kernel_ms']:.4f}", "R4_SHADE": f"{shade / 1e6:.3f}",
        "R4_NU": f"{nu.get('Mrays_s', 0) / 1e3:.2f}", "R4_RATIO": f"{nu.get('ratio_to_headline', 0):.3f}", "R4_FRAC": f"{b['roofline']['frac']:.2f}",
        "R4_VALU": f"{b['roofline']['valu_wave_instr_per_ray']:.1f}"}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for k, v in vals.items():
    s = s.replace(k, v)
open(p, "w").write(s)
print(vals)

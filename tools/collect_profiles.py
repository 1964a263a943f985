"""Turns gpurun_out/final/ (tools/final_round.sh) into the tracked files under profiles/:
  <tag>_bench.json, <tag>_kernel_stats.csv, pmc_traffic.json (HBM-side bytes per launch of the dominant kernel)."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r01"
no_bench = "--no-bench" in sys.argv
src = os.path.join(ROOT, "gpurun_out", "final")
prof = os.path.join(ROOT, "profiles")
if not no_bench:
    line = [l for l in open(os.path.join(src, "bench.log")) if l.startswith("{")][-1]
    open(os.path.join(prof, f"{tag}_bench.json"), "w").write(line)


def newest(pattern):
    """gpurun merges new files into gpurun_out/ without deleting those of earlier calls: take the latest."""
    return max(glob.glob(pattern), key=os.path.getmtime)


shutil.copy(newest(os.path.join(src, "stats", "*", "*kernel_stats.csv")), os.path.join(prof, f"{tag}_kernel_stats.csv"))
if glob.glob(os.path.join(src, "stats_pipelined", "*", "*kernel_stats.csv")):
    shutil.copy(newest(os.path.join(src, "stats_pipelined", "*", "*kernel_stats.csv")), os.path.join(prof, f"{tag}_kernel_stats_pipelined.csv"))


if glob.glob(os.path.join(src, "stats_hybrid", "*", "*kernel_stats.csv")):  # the hybrid frame (k_gbuffer / k_hybrid / k_hy_gi_init / k_wf_shade_hybrid / k_post)
    shutil.copy(newest(os.path.join(src, "stats_hybrid", "*", "*kernel_stats.csv")), os.path.join(prof, f"{tag}_hybrid_kernel_stats.csv"))


def bench_line(logname):
    """The JSON line bench.py printed in a profiling pass: names the sources and the workload the counters belong to."""
    try:
        line = [l for l in open(os.path.join(src, logname)) if l.startswith("{")][-1]
        return json.loads(line)
    except Exception:
        return None


def tie(out, logname, dom_key_per_launch):
    """Adds source_hash / workload / per-ray figures so bench.py can tell a stale summary from a fresh one."""
    b = bench_line(logname)
    if not b or "roofline" not in b:
        return
    out["source_hash"] = b["config"]["source_hash"]
    out["workload"] = b["config"]["workload_key"]
    out["rays_per_launch"] = b["roofline"]["rays_per_launch"]
    for k_launch, k_ray in dom_key_per_launch:
        if k_launch in out:
            out[k_ray] = out[k_launch] / out["rays_per_launch"]


def per_kernel(dirname, counter, durations=None):
    """Sum and launch count of `counter` per kernel; durations (dict): summed End - Start of those launches in ns."""
    f = newest(os.path.join(src, dirname, "*", "*counter_collection.csv"))
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
            if durations is not None:
                durations[k] = durations.get(k, 0.0) + float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return tot, n


fetch, nf = per_kernel("pmc_fetch", "FETCH_SIZE")
write, nw = per_kernel("pmc_write", "WRITE_SIZE")
out = {"unit": "bytes", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) on `python3 bench.py --no-cpu-baseline`",
       "note": "FETCH_SIZE/WRITE_SIZE are in KiB at the L2<->fabric interface (Infinity-Cache hits included). On gfx950 FETCH_SIZE "
               "reads 1/2 of the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM); the guide's x2 correction is applied to the "
               "read side as an upper bound -- this kernel's reads are scattered 16-B gathers, for which the counter is uncalibrated, so "
               "both raw and corrected figures are kept.", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.strip().startswith(("void k_wf", "k_wf", "void k_pathtrace", "k_pathtrace")):
        continue
    launches = max(nf.get(k, 0), nw.get(k, 0), 1)
    rd_raw = fetch.get(k, 0.0) * 1024 / max(nf.get(k, 1), 1)
    wr = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    out["kernels"][k.strip()] = {"launches_profiled": launches, "read_bytes_per_launch_raw": rd_raw, "read_bytes_per_launch_x2": 2 * rd_raw,
                                 "write_bytes_per_launch": wr}
dom = next((k for k in out["kernels"] if "traverse" in k or "pathtrace" in k), None)
if dom:
    out["dominant_kernel"] = dom
    out["hbm_bytes_per_launch"] = out["kernels"][dom]["read_bytes_per_launch_x2"] + out["kernels"][dom]["write_bytes_per_launch"]
    out["hbm_bytes_per_launch_raw"] = out["kernels"][dom]["read_bytes_per_launch_raw"] + out["kernels"][dom]["write_bytes_per_launch"]
    out["read_bytes_per_launch_raw"] = out["kernels"][dom]["read_bytes_per_launch_raw"]
    out["write_bytes_per_launch"] = out["kernels"][dom]["write_bytes_per_launch"]
    tie(out, "pmc_fetch.log", [("hbm_bytes_per_launch", "hbm_bytes_per_ray"), ("read_bytes_per_launch_raw", "read_bytes_per_ray_raw"),
                               ("write_bytes_per_launch", "write_bytes_per_ray")])
# per-ray figures of every kernel of the frame and of the frame as a whole (bench.py: roofline.shade, roofline.frame_hbm): bytes of
# all launches of a kernel over the rays of all traversal launches of the same pass
if dom and "rays_per_launch" in out:
    rounds = sum(v["launches_profiled"] for k, v in out["kernels"].items() if "traverse" in k)
    rays = out["rays_per_launch"] * max(rounds, 1)
    fr = {"read_bytes_per_ray_raw": 0.0, "write_bytes_per_ray": 0.0}
    for k, v in out["kernels"].items():
        v["read_bytes_per_ray_raw"] = v["read_bytes_per_launch_raw"] * v["launches_profiled"] / rays
        v["write_bytes_per_ray"] = v["write_bytes_per_launch"] * v["launches_profiled"] / rays
        v["launches_per_round"] = v["launches_profiled"] / max(rounds, 1)
        fr["read_bytes_per_ray_raw"] += v["read_bytes_per_ray_raw"]
        fr["write_bytes_per_ray"] += v["write_bytes_per_ray"]
    out["frame"] = fr
json.dump(out, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
# VALU issue rate of the dominant kernel (the resource that actually binds it): wave-instructions per launch
if glob.glob(os.path.join(src, "pmc_valu", "*", "*counter_collection.csv")):
    valu, nv = per_kernel("pmc_valu", "SQ_INSTS_VALU")
    waves, _ = per_kernel("pmc_valu", "SQ_WAVES")
    dur = {}
    gui, ng = per_kernel("pmc_valu", "GRBM_GUI_ACTIVE", dur)  # (absent from passes older than round 4)
    issue = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE (kernel-trace only, one lane: VKRT_WF_SUBFRAMES=1 VKRT_WF_FRAMES_IN_FLIGHT=1) on "
                       "`python3 bench.py --no-cpu-baseline`",
             "unit": "VALU wave-instructions", "kernels": {}}
    for k in valu:
        if k.strip().startswith(("void k_wf", "k_wf")):
            issue["kernels"][k.strip()] = {"launches_profiled": nv[k], "valu_wave_instr_per_launch": valu[k] / max(nv[k], 1),
                                           "waves_per_launch": waves.get(k, 0.0) / max(nv[k], 1)}
    d = next((k for k in issue["kernels"] if "traverse" in k), None)
    if d:
        raw = next((k for k in gui if k.strip() == d), None)
        if raw and dur.get(raw):
            # GRBM_GUI_ACTIVE counts busy cycles of every XCD (8 on this part): cycles per XCD / the launches' own durations = the shader
            # clock the kernel ran at under the counter pass (the issue microbenchmark throttles to 1.5-1.8 GHz, this kernel does not)
            issue["kernel_clock_ghz"] = gui[raw] / 8.0 / dur[raw]
            issue["kernel_clock_note"] = "GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / summed launch durations of the dominant kernel in the same pass"
        issue["dominant_kernel"] = d
        issue["valu_wave_instr_per_launch"] = issue["kernels"][d]["valu_wave_instr_per_launch"]
        tie(issue, "pmc_valu.log", [("valu_wave_instr_per_launch", "valu_wave_instr_per_ray")])
    json.dump(issue, open(os.path.join(prof, "pmc_issue.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])

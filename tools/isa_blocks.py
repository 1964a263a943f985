"""Basic-block view of the traversal kernel's assembly (the method behind profiles/r04_experiments.md #126-#128 and the budget in
DESIGN.md section 5): for every loop of k_wf_traverse<false, true, 64, 0> the blocks in layout order with their VALU / v_mov / SALU /
LDS / global-memory instruction counts and branch targets, and per loop the totals.  The sharing loops are the ones with ds_bpermute.

    tools/isa_stats.sh            # writes /tmp/vkrt_isa/wf_traverse.s
    python tools/isa_blocks.py [/tmp/vkrt_isa/wf_traverse.s] [--blocks]
    python tools/isa_blocks.py --json      # writes profiles/isa_mix.json: the static opcode mix of the sharing closest-hit loop (node
                                           # test block / rest of the loop), tied to the sources by vkrt_amd.source_hash: bench.py's
                                           # roofline.issue_mix prices the kernel against the ceiling that mix allows
"""
import re
import sys

KERNEL = "_Z13k_wf_traverseILb0ELb1ELi64ELi0EEv11TraceParams9WfBuffersi"
HALF_RATE = ("cvt", "min", "max", "cmp", "bfe", "_sdwa", "or3", "mul_lo", "perm", "mad_u", "bcnt", "lshl_add", "lshl_or", "and_or", "add3", "v_pk_", "fma_mix")


def kernel_body(path):
    text = open(path).read()
    a = text.index(f"\n{KERNEL}:")
    return text[a:text.index("\n.Lfunc_end", a)].split("\n")


def blocks(lines):
    """blocks in layout order; `loop` = name of the depth-1 loop a block belongs to (fall-through blocks inherit it)"""
    out, cur = [], None
    for l in lines:
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        s = l.strip()
        if m or s.startswith("; %bb."):
            if m:
                name, cmt = m.group(1), m.group(2)
                hdr = re.search(r"(?:Header=|Parent Loop )(BB\d+_\d+)", cmt)
                loop = name[2:] if "Loop Header: Depth=1" in cmt else (hdr.group(1) if hdr else None)
                header = "Loop Header: Depth=1" in cmt
            else:
                name, loop, header = s.split(":")[0][2:], (cur["loop"] if cur else None), False
            cur = {"name": name, "loop": loop, "header": header, "valu": 0, "half": 0, "mov": 0, "salu": 0, "lds": 0, "vmem": 0, "bperm": 0, "br": []}
            out.append(cur)
            continue
        if cur is None or not s or s.startswith(";"):
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            cur["valu"] += 1
            cur["half"] += any(h in op for h in HALF_RATE)
            cur["mov"] += op.startswith("v_mov")
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur["br"].append(op[2:] + "->" + s.split()[1].replace(".LBB", "B"))
            cur["salu"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
            cur["bperm"] += "bpermute" in op
        elif op.startswith(("global_", "buffer_", "scratch_")):
            cur["vmem"] += 1
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    path = args[0] if args else "/tmp/vkrt_isa/wf_traverse.s"
    bb = blocks(kernel_body(path))
    headers = [b for b in bb if b["header"]]
    for h in headers:
        members = [b for b in bb if b["loop"] == h["name"][2:]]
        tot = {k: sum(b[k] for b in members) for k in ("valu", "half", "mov", "salu", "lds", "vmem", "bperm")}
        node = max(members, key=lambda b: b["valu"])
        print(f"loop {h['name']}: {len(members)} blocks, VALU {tot['valu']} (half-rate {tot['half']}, v_mov {tot['mov']}), SALU {tot['salu']}, "
              f"LDS {tot['lds']} (bpermute {tot['bperm']}), memory {tot['vmem']}; node test {node['name']}: {node['valu']} VALU, "
              f"{node['valu'] - node['half']} full-rate + {node['half']} half-rate = {2 * (node['valu'] - node['half']) + 4 * node['half']} cycles")
        if "--json" in sys.argv and tot["bperm"] and "mix" not in locals():  # the first sharing loop = closest-hit walks
            import json
            import os

            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            sys.path.insert(0, root)
            import vkrt_amd

            import subprocess
            hv = re.search(r"HIP version:\s*(\S+)", subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout)
            mix = {"source_hash": vkrt_amd.source_hash(), "hipcc": hv.group(1) if hv else "unknown",
                   "kernel": "k_wf_traverse<false, true, 64, 0>", "loop": "sharing closest-hit walk",
                   "node_test": {"valu": node["valu"], "half_rate": node["half"]},
                   "loop_rest": {"valu": tot["valu"] - node["valu"], "half_rate": tot["half"] - node["half"]},
                   "half_rate_classes": list(HALF_RATE),
                   "note": "static counts from the assembly (tools/isa_stats.sh + tools/isa_blocks.py --json); a half-rate opcode takes two issue slots "
                           "of a full-rate one (profiles/r02_issue_microbench.json)"}
            json.dump(mix, open(os.path.join(root, "profiles", "isa_mix.json"), "w"), indent=1)
        if "--blocks" in sys.argv:
            for b in members:
                print(f"   {b['name']:>12} valu {b['valu']:3d} mov {b['mov']:2d} salu {b['salu']:2d} lds {b['lds']:2d} mem {b['vmem']} {' '.join(b['br'])}")


if __name__ == "__main__":
    main()

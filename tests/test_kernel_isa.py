"""What the traversal kernel's assembly must keep (profiles/r04_experiments.md #126-#128): cross-compiled for gfx950 on the CPU with
the flags csrc/Makefile gives csrc/wf_traverse.hip, the product instantiation k_wf_traverse<false, true, 64, 0> is checked for the
properties its launch time rests on.  No GPU needed; skipped when hipcc is absent."""
import os
import re
import shutil
import subprocess

import pytest

import vkrt_amd

CSRC = os.path.join(vkrt_amd.PKG_DIR, "csrc")
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
KERNEL = "_Z13k_wf_traverseILb0ELb1ELi64ELi0EEv11TraceParams9WfBuffersi"


def makefile_var(name):
    """value of a `NAME := ...` / `NAME ?= ...` line of csrc/Makefile"""
    for line in open(os.path.join(CSRC, "Makefile")):
        m = re.match(rf"^{name}\s*[:?]?=\s*(.*)$", line)
        if m:
            return m.group(1).strip()
    raise KeyError(name)


def hipcc_version():
    out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    m = re.search(r"HIP version:\s*(\S+)", out)
    return m.group(1) if m else "unknown"


@pytest.fixture(scope="module")
def traverse_asm(tmp_path_factory):
    if not HIPCC:
        pytest.skip("hipcc not found")
    # the properties below are properties of what THIS compiler makes of the sources: profiles/isa_mix.json records the compiler the
    # committed figures were taken with, and another one (a ROCm update) gets to re-take them instead of failing here
    import json

    want = json.load(open(os.path.join(vkrt_amd.REPO_ROOT, "profiles", "isa_mix.json"))).get("hipcc")
    if want and want != hipcc_version():
        pytest.skip(f"hipcc {hipcc_version()} is not the compiler of profiles/isa_mix.json ({want}): re-take tools/isa_blocks.py --json")
    flags = makefile_var("FLAGS").replace("$(ARCH)", makefile_var("ARCH")).replace("-fPIC", "").split()
    flags += makefile_var("FLAGS_wf_traverse").split()
    out = tmp_path_factory.mktemp("isa") / "wf_traverse.s"
    subprocess.run([HIPCC] + flags + ["--cuda-device-only", "-S", "-o", str(out), "wf_traverse.hip"], cwd=CSRC, check=True,
                   stderr=subprocess.DEVNULL)
    text = open(out).read()
    start = text.index(f"\n{KERNEL}:")
    body = text[start:text.index("\n.Lfunc_end", start)]
    meta = text[text.index(f".amdhsa_kernel {KERNEL}"):]
    meta = meta[:meta.index(".end_amdhsa_kernel")]
    return body, meta


def test_makefile_builds_the_traversal_unit_without_the_slp_vectoriser():
    assert "wf_traverse.hip" in makefile_var("SRCS").split()
    assert "-fno-slp-vectorize" in makefile_var("FLAGS_wf_traverse").split()
    assert "-fno-slp-vectorize" in makefile_var("FLAGS_pathtrace").split()
    assert "-fno-slp-vectorize" not in makefile_var("FLAGS_wavefront").split()  # the gather-bound shade kernel keeps it (#126)


def test_traversal_kernel_has_no_packed_fp32_and_keeps_five_waves(traverse_asm):
    body, meta = traverse_asm
    # packed FP32 issues at half rate on gfx950 and needs its operands moved into register pairs (#126)
    assert not re.search(r"\bv_pk_(mul|fma|add)_f32\b", body)
    # five waves per SIMD: at most 96 VGPRs of the 512 (one wave less costs 20 %, profiles/r02_experiments.md #74), nothing spilled
    vgprs = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
    scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta).group(1))
    assert vgprs <= 96, vgprs
    assert scratch == 0, scratch
    assert "scratch_" not in body


def test_node_test_uses_one_sdwa_shift_per_child(traverse_asm):
    body, _ = traverse_asm
    # six copies of the node test in the kernel (two sharing loops, two plain loops, two flush paths), eight children each (#127)
    sdwa = re.findall(r"v_lshlrev_b32_sdwa v\d+, v\d+, v\d+ dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_(\d) src1_sel:BYTE_(\d)", body)
    assert len(sdwa) >= 48, len(sdwa)
    assert all(a == b for a, b in sdwa)
    # ... and the planes stay 8-bit with at most one conversion each: 6 planes per child (a range, not a count: the number of copies
    # of the node test is the compiler's business)
    cvt = len(re.findall(r"v_cvt_f32_ubyte[0-3]", body))
    assert 0 < cvt <= 6 * len(sdwa), (cvt, len(sdwa))


def test_candidate_registers_are_not_reinitialised_per_nesting_level(traverse_asm):
    body, _ = traverse_asm
    # #128: the kernel had 460 v_mov_b32 with the candidate hit live across a whole step (358 without the SLP moves); local candidates: ~340
    movs = len(re.findall(r"^\s+v_mov_b32", body, flags=re.M))
    assert movs < 400, movs

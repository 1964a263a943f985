"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, validates
arguments, and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

import vkrt_amd
from vkrt_amd import abi

ROOT = vkrt_amd.REPO_ROOT


def test_headers_compile_as_c_and_cxx(tmp_path):
    for lang, cc in (("c", "gcc"), ("c++", "g++")):
        src = tmp_path / f"t.{ 'c' if lang == 'c' else 'cpp'}"
        src.write_text('#include "vkrt.h"\nint main(void){return sizeof(PushConstantRay)==44?0:1;}\n')
        exe = tmp_path / f"t_{cc}"
        subprocess.run([cc, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "vkrt.h")).read()
    declared = set(re.findall(r"\b(vkrt_[a-z_]+)\s*\(", header))
    assert declared == set(abi.VKRT_SYMBOLS), declared ^ set(abi.VKRT_SYMBOLS)
    assert os.path.exists(vkrt_amd.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(vkrt_amd.LIB_PATH)
    for s in abi.VKRT_SYMBOLS:
        assert hasattr(lib, s), s
    assert lib.vkrt_abi_version() == abi.VKRT_ABI_VERSION


def test_argument_validation_and_no_cpu_fallback(cornell_flat):
    from vkrt_amd.renderer import load_library

    lib = load_library()
    h = C.c_void_p()
    desc, keep = cornell_flat.to_desc()
    desc.struct_size = 8
    assert lib.vkrt_scene_create(C.byref(desc), 0, C.byref(h)) == 1  # VKRT_ERR_INVALID_ARGUMENT
    assert b"struct_size" in lib.vkrt_last_error()
    desc, keep = cornell_flat.to_desc()
    desc.light_count = 0
    assert lib.vkrt_scene_create(C.byref(desc), 0, C.byref(h)) == 1
    # non-finite geometry has no place in an acceleration structure: refused up front (round 3), positions and node matrices alike
    import copy

    import numpy as np

    for field, index in (("positions", (5, 1)), ("nodes", 0)):
        bad = copy.deepcopy(cornell_flat)
        if field == "positions":
            bad.positions[index] = np.nan
        else:
            m = np.array(bad.nodes[index]["worldMatrix"], np.float32); m[3] = np.inf
            bad.nodes[index]["worldMatrix"] = m
        desc, keep = bad.to_desc()
        assert lib.vkrt_scene_create(C.byref(desc), 0, C.byref(h)) == 1
        assert b"not finite" in lib.vkrt_last_error() and not h.value
    # options 10-12 (watertight test, dead-shadow-ray skipping, any-hit dissolve stage) and 13-14 (frames in flight, triangle
    # pre-splitting) exist; 15 does not.  ABI version 3 is the first that promises them (and vkrt_pathtrace_frames)
    from vkrt_amd import abi

    assert (abi.VKRT_OPT_WATERTIGHT, abi.VKRT_OPT_SKIP_DEAD_SHADOW_RAYS, abi.VKRT_OPT_ANYHIT_DISSOLVE) == (10, 11, 12)
    hdr = open(os.path.join(ROOT, "include", "vkrt.h")).read()
    assert (abi.VKRT_OPT_WF_FRAMES_IN_FLIGHT, abi.VKRT_OPT_SPLIT_BUDGET) == (13, 14)
    assert "VKRT_OPT_LAST            = 14" in hdr and "#define VKRT_ABI_VERSION 4" in hdr and abi.VKRT_ABI_VERSION == 4
    desc, keep = cornell_flat.to_desc()
    if lib.vkrt_device_count() == 0:
        # the product never computes on the CPU: without a device creation must fail loudly
        assert lib.vkrt_scene_create(C.byref(desc), 0, C.byref(h)) == 2  # VKRT_ERR_NO_DEVICE
        assert b"no HIP device" in lib.vkrt_last_error()
        assert not h.value


def test_shard_rows_math():
    from vkrt_amd.renderer import load_library
    from vkrt_amd.sharding import shard_row_indices

    lib = load_library()
    for H in (1, 15, 16, 17, 200, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            total = 0
            for rank in range(world):
                sh = abi.Shard(64, H, 16 if world > 1 else 0, world, rank)
                rows = int(lib.vkrt_shard_rows(C.byref(sh)))
                assert rows == len(shard_row_indices(H, world, rank))
                total += rows
            assert total == H


def test_ctypes_mirror_matches_header_struct_sizes(tmp_path):
    """abi.py restates include/vkrt.h by hand: compile the header and compare sizeof of every struct the harness passes."""
    pairs = {"vkrt_prim_mesh": abi.PrimMesh, "vkrt_node": abi.Node, "vkrt_texture": abi.Texture, "vkrt_scene_desc": abi.SceneDesc,
             "vkrt_shard": abi.Shard, "vkrt_trace_opts": abi.TraceOpts, "vkrt_counters": abi.Counters, "vkrt_accel_info": abi.AccelInfo, "vkrt_accel_check": abi.AccelCheck,
             "vkrt_gbuffer": abi.Gbuffer, "vkrt_trace_timing": abi.TraceTiming, "GlobalUniforms": abi.GlobalUniforms,
             "PushConstantRay": abi.PushConstantRay, "PushConstantPost": abi.PushConstantPost, "GltfPBRMaterial": abi.GltfPBRMaterial,
             "GltfLight": abi.GltfLight, "PrimMeshInfo": abi.PrimMeshInfo}
    body = "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in pairs)
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "vkrt.h"\nint main(void){\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for name, ct in pairs.items():
        assert int(got[name]) == C.sizeof(ct), (name, got[name], C.sizeof(ct))

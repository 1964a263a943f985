"""Eight times the bench scene: 2,097,152 instanced triangles (device-built tree, with and without pre-split references) at 2560x1440,
three progressive frames through vkrt_pathtrace_frames, against oracle rows -- the sizes the library is meant for do not stop at the
bench workload (32-bit offsets of the hit shader's tables, reference counts, stack depth, event pool, working set of three frames in
flight: 9.5 GB)."""
import os
import sys

import numpy as np
import pytest

from conftest import default_camera

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
THREADS = min(16, os.cpu_count() or 1)


def test_two_million_triangles_1440p_three_frames_rows_match_oracle():
    import atrium
    import oracle_py
    from vkrt_amd import abi
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer

    flat, info = atrium.build_atrium(2097152, seed=7, with_textures=True)
    assert 2000000 < info["triangles"] <= 2097152 + 64
    W, H = 2560, 1440
    cam = default_camera(W, H, **atrium.DEFAULT_CAMERA)
    lights = len(flat.lights)
    rows = np.unique(np.linspace(0, H - 1, 6).astype(np.uint32))
    orc = oracle_py.OracleScene(flat)
    ref = None
    for f in range(3):
        ref, cref = orc.render(make_push_constants(samples=2, depth=6, frame=f, lights_count=lights), cam, W, H, seed=11 + f, rows=rows, image=ref, threads=THREADS)
    hashes = []
    for budget in (0, 20):
        r = Renderer(flat, device=0, build="ploc", options={abi.VKRT_OPT_SPLIT_BUDGET: budget})
        a = r.accel_info()
        assert a["triangle_count"] == info["triangles"] and a["triangle_count"] <= a["reference_count"] <= a["triangle_count"] * (100 + budget) // 100
        chk = r.check_accel()
        assert chk["triangles_missing"] == 0 and chk["triangles_repeated"] == 0 and chk["box_violations"] == 0 and chk["bad_references"] == 0 and chk["triangles_uncovered"] == 0, chk
        img = r.pathtrace_frames(make_push_constants(samples=2, depth=6, frame=0, lights_count=lights), cam, W, H, 3, seed=11)
        c = r.counters()
        assert c["traversal_faults"] == 0 and c["pixels"] == 3 * W * H
        got = img.cpu().numpy()
        r.close()
        part = got[rows]
        rmse = float(np.sqrt(np.mean((part[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64)) ** 2)))
        assert rmse < 1e-3
        assert float(np.mean(np.any(part.view(np.uint32) != ref.view(np.uint32), axis=-1))) < 1e-4
        hashes.append(got.tobytes())
    assert hashes[0] == hashes[1]  # the split tree renders the unsplit tree's image, bit for bit

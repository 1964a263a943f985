"""Replay one case of tools/fuzz_parity.py and localise a hybrid-frame mismatch down to single rays:

    python tools/debug_case.py SEED [--layout 0|1]

For every pixel where the HIP accumulation image differs from the oracle's brute-force image, the rays raytraceHybrid.rgen
traces for that pixel (oracle ray tap, brute force) are replayed one by one through vkrt_debug_trace_rays (the case's own
node layout and the other one) and the oracle's brute-force / tree queries; rays whose results differ are printed."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

import fuzz_parity
from vkrt_amd import abi
from vkrt_amd.renderer import Renderer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("seed", type=int)
    ap.add_argument("--watertight", action="store_true")
    ap.add_argument("--force-opt", action="append", default=[], metavar="ID=VALUE", help="execution options forced on top of the case's draw, as in fuzz_parity.py")
    a = ap.parse_args()
    from vkrt_amd import abi as _abi
    force = {int(k): int(v) for k, v in (x.split("=") for x in a.force_opt)}
    if a.watertight:
        force[_abi.VKRT_OPT_WATERTIGHT] = 1
    force = force or None

    def replay_ray(S, r2, k, o, d, tmin, tmax, anyh):
        orc, r = S["oracle"], S["renderer"]
        tg, ug, vg, gg = r.trace_rays(o, d, tmin, tmax, anyh)
        t2, u2, v2, g2 = r2.trace_rays(o, d, tmin, tmax, anyh)
        tb, ub, vb, gb, _ = orc.trace_rays(o, d, tmin, tmax, anyh, use_bvh=False)
        tt, ut, vt, gt, _ = orc.trace_rays(o, d, tmin, tmax, anyh, use_bvh=True)
        hg, h2, hb, ht = int(gg[0]), int(g2[0]), int(gb[0]), int(gt[0])
        same = ((hg >= 0) == (hb >= 0)) if anyh else (hg == hb and tg[0] == tb[0])
        mark = "" if same else "   <<<<<< differs"
        print(f"  ray {k} any={int(anyh)} o={np.asarray(o).tolist()} d={np.asarray(d).tolist()} [{tmin!r}, {tmax!r}]: case-layout (t={tg[0]!r}, gid={hg})  other-layout (t={t2[0]!r}, gid={h2})  "
              f"brute (t={tb[0]!r}, gid={hb})  oracle-tree (t={tt[0]!r}, gid={ht}){mark}")
        if not same:
            v9, _ = orc.triangles()
            for name, g in (("gpu", hg), ("brute", hb)):
                if g >= 0 and not anyh:
                    print(f"    {name} triangle {g}: v0 {v9[g][0:3].tolist()} e1 {v9[g][3:6].tolist()} e2 {v9[g][6:9].tolist()}")
            print(f"    case-layout closest over the interval: {r.trace_rays(o, d, tmin, tmax, False)}; brute: {orc.trace_rays(o, d, tmin, tmax, False, use_bvh=False)[:4]}")
        return same

    def hook_pathtrace(S):
        from vkrt_amd.flat_scene import make_push_constants
        orc, r, cam, W, H = S["oracle"], S["renderer"], S["cam"], S["W"], S["H"]
        # brute-force reference of the same frames
        ref = None
        for f in range(S["first_frame"], S["first_frame"] + S["frames"]):
            pc = make_push_constants(samples=S["spp"], depth=S["depth"], frame=f, lights_count=S["L"])
            ref, _ = orc.render(pc, cam, W, H, seed=S["seed"] + f, flags=S["flags"], image=ref, use_bvh=False)
        got = S["got"]
        diff = ((got.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(got) & np.isnan(ref))).any(-1)
        print(json.dumps(dict(stage="pathtrace", kind=S["kind"], opts={int(k): int(v) for k, v in S["opts"].items()}, W=W, H=H, spp=S["spp"], depth=S["depth"], frames=S["frames"],
                              differing_pixels=int(diff.sum()))))
        other_opts = dict(S["opts"])
        other_opts[abi.VKRT_OPT_BVH_LAYOUT] = 1 if S["opts"].get(abi.VKRT_OPT_BVH_LAYOUT, 1) == 0 else 0
        other_opts.pop(abi.VKRT_OPT_MODE, None)
        r2 = Renderer(S["flat"], device=0, build=S["kind"], options=other_opts)
        for y, x in list(zip(*np.nonzero(diff)))[:4]:
            print(f"pixel ({x},{y}): gpu {got[y, x]} brute {ref[y, x]}")
            for f in range(S["first_frame"], S["first_frame"] + S["frames"]):  # every frame of the sequence (frames > 0 jitter the camera ray)
                pc = make_push_constants(samples=S["spp"], depth=S["depth"], frame=f, lights_count=S["L"])
                _, log = orc.pixel_log(pc, cam, W, H, int(x), int(y), seed=S["seed"] + f, flags=S["flags"], use_bvh=False)
                print(f" frame {f}:")
                k = 0
                for rec in log:
                    if rec[0] in (-2.0, -3.0):
                        anyh = rec[0] == -3.0
                        if not replay_ray(S, r2, k, rec[1:4].copy(), rec[4:7].copy(), 0.001, float(rec[7]), anyh):
                            break
                        k += 1
        r2.close()

    def hook(S):
        if S.get("stage") == "pathtrace":
            return hook_pathtrace(S)
        orc, r, pc, cam, W, H, g = S["oracle"], S["renderer"], S["pc"], S["cam"], S["W"], S["H"], S["gbuffer"]
        accb, _ = orc.hybrid(pc, cam, W, H, g, seed=S["seed"], flags=S["flags"], use_bvh=False)
        acc = S["acc"]
        diff = ((acc.view(np.uint32) != accb.view(np.uint32)) & ~(np.isnan(acc) & np.isnan(accb))).any(-1)
        print(json.dumps(dict(kind=S["kind"], opts={int(k): int(v) for k, v in S["opts"].items()}, W=W, H=H, depth=pc.depth, shadows=pc.useShadows, ao=pc.useAO, gi=pc.useGI,
                              differing_pixels=int(diff.sum()))))
        other_opts = dict(S["opts"])
        other_opts[abi.VKRT_OPT_BVH_LAYOUT] = 1 if S["opts"].get(abi.VKRT_OPT_BVH_LAYOUT, 1) == 0 else 0
        other_opts.pop(abi.VKRT_OPT_MODE, None)
        r2 = Renderer(S["flat"], device=0, build=S["kind"], options=other_opts)
        for y, x in zip(*np.nonzero(diff)):
            print(f"pixel ({x},{y}): gpu {acc[y, x]} brute {accb[y, x]}")
            _, rays = orc.hybrid_pixel_rays(pc, cam, W, int(x), int(y), g, seed=S["seed"], flags=S["flags"], use_bvh=False)
            for k, ray in enumerate(rays):
                o, d, tmin, tmax, anyh = ray[0:3], ray[3:6], float(ray[6]), float(ray[7]), bool(ray[8])
                tg, ug, vg, gg = r.trace_rays(o, d, tmin, tmax, anyh)
                t2, u2, v2, g2 = r2.trace_rays(o, d, tmin, tmax, anyh)
                tb, ub, vb, gb, _ = orc.trace_rays(o, d, tmin, tmax, anyh, use_bvh=False)
                tt, ut, vt, gt, _ = orc.trace_rays(o, d, tmin, tmax, anyh, use_bvh=True)
                hg, h2, hb, ht = int(gg[0]), int(g2[0]), int(gb[0]), int(gt[0])
                same = ((hg >= 0) == (hb >= 0)) if anyh else (hg == hb and tg[0] == tb[0])
                mark = "" if same else "   <<<<<< differs"
                print(f"  ray {k} any={int(anyh)} o={o.tolist()} d={d.tolist()} [{tmin!r}, {tmax!r}]: case-layout (t={tg[0]!r}, gid={hg})  other-layout (t={t2[0]!r}, gid={h2})  "
                      f"brute (t={tb[0]!r}, gid={hb})  oracle-tree (t={tt[0]!r}, gid={ht}){mark}")
                if not same:
                    # every triangle the brute-force loop accepts on the UNBOUNDED interval, to see how close to the interval's ends the deciding hit is
                    tc, _, _, gc, _ = orc.trace_rays(o, d, 0.0, 1e30, False, use_bvh=False)
                    print(f"    closest over (0, inf): brute t={tc[0]!r} gid={int(gc[0])}; case-layout closest over the ray's interval: {r.trace_rays(o, d, tmin, tmax, False)}")
                    print(f"    other-layout closest over the interval: {r2.trace_rays(o, d, tmin, tmax, False)}; brute: {orc.trace_rays(o, d, tmin, tmax, False, use_bvh=False)[:4]}")
        r2.close()

    fuzz_parity.run_case(a.seed, verbose=True, hook=hook, force_opts=force)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing path (BASELINE.json metric: Mrays/s +
achieved HBM GB/s, Sponza 1080p 8-bounce).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          (no WORLD_SIZE in the environment: starts the N ranks itself, see self_launch)
    python bench.py --gltf Sponza.gltf --eye X Y Z --center X Y Z [--fov F]   (a supplied scene file instead of the atrium)

A "step" is one progressive frame of the workload: the seeded procedural Sponza-class atrium
(tools/atrium.py; the real Sponza.gltf is not available offline), 16 spp per frame
(PushConstantRay.samples = 16), depth 8, 8 fallback lights, frame index = step index (frames > 0
jitter and blend into the resident rgba32f image, raytrace.rgen:44,136-141).  The K timed steps are
handed to the library --frames-per-call at a time (vkrt_pathtrace_frames: the reference's frame loop
with the camera at rest, main.cpp:503-508), which keeps consecutive frames in flight; every step is
still one whole frame with all its rays, and the image after K steps is bit-identical to K
single-frame calls (tests/test_gpu_frames.py).  Scene, BVH, working set and image are resident in HBM
before the timed region.

  N = 1   BASELINE config 3: 1920x1080.
  N > 1   BASELINE config 4: ONE 3840x2160 frame split into 16-row strips dealt round-robin to the N ranks
          (strong scaling; every rank holds the whole scene), one RCCL all_gather of the strips per frame.
          --weak renders an N x 1080p-pixel 16:9 image instead (per-GPU work fixed).
Mrays/s counts the closest-hit + shadow traceRay calls actually issued (device counters), summed over ranks.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     : dominant kernel k_wf_traverse, priced against the resource that BINDS it: VALU issue.  achieved = VALU
                 wave-instructions per launch (SQ_INSTS_VALU per ray from the committed PMC pass x rays per launch of this
                 run) / the kernel's mean launch duration (HIP events on the launch stream, this run); peak = what follows from
                 the guide: 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per full-rate wave64 instruction = 1228.8 G wave-instr/s;
                 `frac_calibrated` prices the same rate against the full-rate issue rate tools/issue_microbench.hip sustained
                 (profiles/r02_issue_microbench.json: a pure-VALU load, clock-throttled to 1.5-1.8 GHz; the kernel's own
                 counter pass runs at ~2.3-2.4 GHz, so that figure flatters and is kept only for continuity with round 3).
                 `issue_mix` (an estimate kept beside `frac`): the ceiling the kernel's opcode mix allows -- a half-rate opcode
                 takes two issue slots, so a share h of them caps the rate at 1 / (1 + h) of the peak; h from the static opcode classes of the
                 assembly (profiles/isa_mix.json, tools/isa_blocks.py --json) weighted by this run's dynamic step counts.
                 Sub-blocks, none of them a bound for this kernel: `hbm_own_bytes` (the kernel's own algorithmic bytes --
                 80 B per 8-wide node visited + 48 B per triangle tested + ray / hit records, visit counts from an
                 instrumented launch of this run -- against the 8 TB/s HBM3E peak), `contract` (SURVEY 8d accounting on the
                 oracle's BVH2; exceeds 1), `l2_gather` (own bytes against the guide's measured L2 gather rate: the tree is
                 L2-resident), `traffic` / `traffic_detail` (fabric-side HBM bytes per launch from separate PMC passes).
                 PMC-derived numbers come from profiles/pmc_*.json; they are tied to the kernel sources by
                 vkrt_amd.source_hash and flagged `pmc_stale` when that file was measured on other sources.
  cpu_baseline : the CPU oracle (scalar C++ restatement, kind "port") timed on bounded row samples of the same frame: on every
                 hardware thread this process may use (`value`, `cores`), on 16 threads (`threads_16`: the share of the host a
                 1-GPU box is sized for) and on one thread (`single_thread`), with the host's CPU model string.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / tensor sharing fail with hipIpcGetMemHandle otherwise); the
# environment normally exports it already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, HBM)
L2_GATHER_GBS = 16800.0   # same guide, "Indexed rows": rows shared by every workgroup, served by the XCDs' L2 (16.8-18.8 TB/s)
ISSUE_NOMINAL_G = 256 * 4 * 2.4 / 2.0  # 1228.8 G wave64 VALU instructions/s: 256 CUs x 4 SIMDs x 2.4 GHz, 2 cycles per full-rate instruction


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (one process per GPU, the same
    torch.distributed.run command the driver would use) BEFORE this process touches the GPU, relay rank 0's JSON line and
    return the launcher's exit code.  (Never re-exec: a process that has initialised the GPU must not be replaced.)"""
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] no WORLD_SIZE in the environment: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in p.stdout:  # rank 0's line (and nothing else) goes to our stdout
        if line.lstrip().startswith("{"):
            print(line, end="", flush=True)
        else:
            print(line, end="", file=sys.stderr, flush=True)
    return p.wait()


def launch_selftest(args):
    """--launch-selftest: the control flow of an N-rank run without a GPU (gloo): rendezvous, one all_reduce, rank 0's line."""
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_selftest": True, "n_gpus": world, "gpus_arg": args.gpus, "rank_sum": float(t.item())}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota():
    """CPUs the cgroup of this process may use at once (cgroup v2 cpu.max, v1 cpu.cfs_quota_us), rounded up; None = unlimited / unknown."""
    import math

    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, math.ceil(int(q) / int(per)))
        return None
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            return max(1, math.ceil(q / per))
    except (OSError, ValueError):
        pass
    return None


def image_size(n_gpus, base_w, base_h):
    """16:9 image with n_gpus x (base_w x base_h) pixels, width a multiple of 8 (--weak)."""
    if n_gpus <= 1:
        return base_w, base_h
    import math

    w = int(round(base_w * math.sqrt(n_gpus) / 8.0)) * 8
    h = int(round(w * base_h / base_w))
    return w, h


def workload_key(args, W, H, triangles):
    scene = f"atrium{triangles}" if not args.gltf else f"gltf-{os.path.basename(args.gltf)}-{triangles}"
    if not args.gltf and args.variant != "default":
        scene += "-" + args.variant
    return f"{scene}_{W}x{H}_{args.spp}spp_d{args.depth}_{'tex' if not args.no_textures else 'notex'}_{args.build}"


def fresh_profile(path, key):
    """A committed PMC summary, or None when it was measured on other sources / another workload."""
    import vkrt_amd

    try:
        d = json.load(open(path))
    except Exception:
        return None
    if d.get("source_hash") != vkrt_amd.source_hash() or d.get("workload") != key:
        return None
    return d


def issue_mix_block(mix, valu_per_ray, node_wave_steps_per_ray, frac_of_peak):
    """roofline.issue_mix: what the kernel's opcode mix allows.  A half-rate opcode (v_cvt_f32_ubyte*, v_min/max(3)_f32, v_cmp, SDWA and
    VOP3 integer work; profiles/r02_issue_microbench.json) takes the issue slots of two full-rate ones, so a wave whose instructions
    are a share h half-rate can reach at most 1 / (1 + h) of the full-rate peak.  h is estimated from the STATIC mix of the assembly
    (profiles/isa_mix.json, tools/isa_blocks.py --json: the node-test block and the rest of the sharing loop) weighted by this run's
    DYNAMIC counts: node wave-steps per wave (the kernel's own counters) x the node test's instructions; every other VALU instruction
    the counters saw (profiles/pmc_issue.json) gets the static mix of the rest of the loop."""
    per_wave = 64.0 * valu_per_ray
    node_steps = 64.0 * node_wave_steps_per_ray
    node_instr = min(node_steps * mix["node_test"]["valu"], per_wave)
    rest_instr = per_wave - node_instr
    rest_share = mix["loop_rest"]["half_rate"] / max(mix["loop_rest"]["valu"], 1)
    half = node_instr * mix["node_test"]["half_rate"] / max(mix["node_test"]["valu"], 1) + rest_instr * rest_share
    h = half / per_wave
    ceiling = 1.0 / (1.0 + h)
    return {"half_rate_share": h, "ceiling_frac_of_peak": ceiling, "frac_of_mix_ceiling": frac_of_peak / ceiling,
            "valu_instr_per_wave": per_wave, "node_test_instr_per_wave": node_instr, "node_wave_steps_per_wave": node_steps,
            "node_test": mix["node_test"], "loop_rest": mix["loop_rest"],
            "note": "an estimate beside roofline.frac, not a replacement: static opcode classes of the assembly (profiles/isa_mix.json) weighted by "
                    "the dynamic step counts of this run; frac_of_mix_ceiling = frac / ceiling_frac_of_peak"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=0, help="default: 1920 (N = 1) / 3840 (N > 1)")
    ap.add_argument("--height", type=int, default=0, help="default: 1080 (N = 1) / 2160 (N > 1)")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--triangles", type=int, default=262144)
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--build", choices=["sah", "lbvh", "ploc"], default="ploc")
    ap.add_argument("--no-textures", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-builder", action="store_true", help="skip the short runs with the builders that were not selected (and with the Sponza-like tessellation)")
    ap.add_argument("--frames-per-call", type=int, default=6, help="timed steps handed to the library per vkrt_pathtrace_frames call (1 = one vkrt_pathtrace per step)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    ap.add_argument("--weak", action="store_true", help="N > 1: N x 1080p pixels instead of one fixed 3840x2160 frame")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI (default); gloo for rehearsals")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="all ranks use cuda:0 (control-flow rehearsal on a 1-GPU box; needs --backend gloo)")
    ap.add_argument("--variant", default="default", help="atrium variant (tools/atrium.py), e.g. nonuniform = wall-sized triangles next to millimetre trim and drapery")
    ap.add_argument("--gltf", default=None, help="render this .gltf / .glb through the product's C++ loader instead of the procedural atrium "
                                                 "(e.g. the Khronos Sponza the reference's config.json names)")
    ap.add_argument("--eye", type=float, nargs=3, default=None, help="camera position (with --gltf; default: the reference's (0, 0, 15))")
    ap.add_argument("--center", type=float, nargs=3, default=None, help="look-at point (default: the origin)")
    ap.add_argument("--up", type=float, nargs=3, default=None)
    ap.add_argument("--fov", type=float, default=None, help="vertical field of view in degrees (default 60)")
    ap.add_argument("--launch-selftest", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.launch_selftest:
        sys.exit(launch_selftest(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    import vkrt_amd
    from vkrt_amd import abi, host_py
    from vkrt_amd.flat_scene import make_push_constants
    from vkrt_amd.renderer import Renderer
    from vkrt_amd.sharding import gather_image, make_shard
    import atrium

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    if args.rehearse_on_one_gpu:
        assert args.backend == "gloo", "--rehearse-on-one-gpu needs --backend gloo (RCCL refuses two ranks on one device)"
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")

    if world > 1 and args.weak:
        W, H = image_size(world, args.width or 1920, args.height or 1080)
        scaling = "weak"
    elif args.width and args.height:
        W, H = args.width, args.height
        scaling = "strong"
    else:
        W, H = (3840, 2160) if world > 1 else (1920, 1080)
        scaling = "strong"  # the frame does not grow with N (N = 1: BASELINE config 3; N > 1: config 4, see config.same_frame_single_gpu_Mrays_s)
    if args.gltf:
        # a supplied scene file (the reference's config.json scenes: Sponza, suntemple, ... -- git-ignored upstream): the product's
        # C++ loader flattens it exactly as for vkrt_render; camera from the command line, defaults = the reference's start-up camera
        flat = host_py.load_gltf(args.gltf)
        info = {"triangles": int(flat.instanced_triangle_count), "seed": None}
        camkw = {k: tuple(v) for k, v in (("eye", args.eye), ("center", args.center), ("up", args.up)) if v is not None}
        if args.fov is not None:
            camkw["fov"] = args.fov
        scene_text = f"{os.path.basename(args.gltf)} ({info['triangles']} instanced tris, {len(flat.textures)} textures)"
    else:
        kw = {} if args.variant == "default" else {"variant": args.variant}
        flat, info = atrium.build_atrium(args.triangles, seed=args.scene_seed, with_textures=not args.no_textures, **kw)
        camkw = dict(atrium.DEFAULT_CAMERA)
        scene_text = (f"procedural Sponza-class atrium ({info['triangles']} tris, seed {info['seed']}, "
                      f"{'textured' if not args.no_textures else 'untextured'}{'' if args.variant == 'default' else ', variant ' + args.variant})")
    cam = host_py.global_uniforms(width=W, height=H, **camkw)  # the product's camera (host/camera.h)
    lights = int(flat.lights.shape[0])

    # ---- scene upload + acceleration structure (timed apart from the trace) ----------------------------------
    r = Renderer(flat, device=local_rank, build=None)
    builds = {}
    others = [k for k in ("lbvh", "ploc", "sah") if k != args.build]
    torch.cuda.synchronize(dev)
    for kind in (others if (rank == 0 and world == 1 and not args.no_other_builder) else []) + [args.build]:
        t0 = time.perf_counter()
        r.build(kind)
        torch.cuda.synchronize(dev)
        a = r.accel_info()
        builds[kind] = {"build_ms": (time.perf_counter() - t0) * 1e3, "build_ms_library": a["build_ms"], "nodes": a["node_count"],
                        "depth": a["max_depth"], "sah_cost": a["sah_cost"], "node_bytes": a["node_bytes"], "triangle_bytes": a["triangle_bytes"]}
    accel = r.accel_info()
    shard = make_shard(W, H, world, rank)
    rows_local = r.shard_rows(shard)
    image = torch.zeros((rows_local, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    per_call = max(1, args.frames_per_call)
    r.reserve(shard, stream, frames_per_call=per_call)  # no allocation / host synchronisation inside the timed launches

    def render(first_frame, n=1, flags=0, rr=None, sh=None, img=None):
        """n progressive frames from `first_frame` on: one vkrt_pathtrace (n = 1) or one vkrt_pathtrace_frames call"""
        rr, sh, img = rr or r, sh or shard, image if img is None else img
        pc = make_push_constants(samples=args.spp, depth=args.depth, frame=first_frame, lights_count=lights)
        if n == 1:
            rr.pathtrace(pc, cam, W, H, seed=first_frame, flags=flags, shard=sh, image=img, stream=stream)
        else:
            rr.pathtrace_frames(pc, cam, W, H, n, seed=first_frame, flags=flags, shard=sh, image=img, stream=stream)

    def step(frame, flags=0, n=1):
        render(frame, n, flags)
        if world > 1:
            # one gather per call (the image exists once its last frame is blended), overlapped with the next call's kernels (side
            # stream); the timed region ends with a device-wide synchronisation, so the last gathered image is complete inside it
            return gather_image(image if args.backend == "nccl" else image.cpu(), H, world, rank, overlap=args.backend == "nccl")
        return image

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_frames(first_frame, n, rr=None):
        sync()
        (rr or r).reset_counters(stream)
        sync()
        t0 = time.perf_counter()
        k = 0
        while k < n:  # exactly n steps (frames), per_call of them per library call
            m = min(per_call, n - k)
            if rr is None:
                step(first_frame + k, n=m)
            else:
                render(first_frame + k, m, rr=rr)
            k += m
        torch.cuda.synchronize(dev)
        t_local = time.perf_counter() - t0  # this rank's own work (before the barrier): load-balance figure
        sync()
        dt = time.perf_counter() - t0
        c = (rr or r).counters()
        return dt, t_local, c

    if per_call > 1:  # touch the frames-in-flight path once before anything is timed (a throw-away image; lanes, events and staging planes exist after vkrt_reserve)
        render(0, 2, img=torch.zeros_like(image))
    for f in range(args.warmup):
        step(f)
    elapsed, local_s, cnt = timed_frames(args.warmup, args.steps)
    rays_local = cnt["rays_closest"] + cnt["rays_shadow"]
    assert cnt["traversal_faults"] == 0, "traversal faults: a stack push was dropped or a walk hit the step bound"

    # ---- untimed extra passes: per-kernel durations (HIP events around every traversal launch) and the kernel's own
    #      work counters (instrumented launch) ---------------------------------------------------------------------------
    frame_ms, trav_ms, trav_launches, shade_launch_ms = [], [], [], []
    next_frame = args.warmup + args.steps
    for k in range(min(args.steps, 3)):
        step(next_frame + k, flags=abi.VKRT_TRACE_TIME_KERNELS)
        torch.cuda.synchronize(dev)
        tm = r.last_trace_timing()
        frame_ms.append(tm["total_ms"])
        trav_ms.append(tm["traverse_ms"])
        trav_launches.append(tm["traverse_launches"])
        if tm["shade_launches"]:
            shade_launch_ms.append(tm["shade_ms"] / tm["shade_launches"])
    mode = tm["mode"]
    r.reset_counters(stream)
    step(next_frame + 3, flags=abi.VKRT_TRACE_COUNT_TRAVERSAL)
    torch.cuda.synchronize(dev)
    work = r.counters()
    rays_frame = work["rays_closest"] + work["rays_shadow"]

    # ---- N > 1: the SAME frame on one GPU (rank 0 renders the whole WxH frame, untimed region; the other ranks wait in the
    #      collectives below): the single-GPU rate the scaling of this frame is to be read against -- the N = 1 line of the
    #      contract renders BASELINE config 3 (1080p), whose ray rate differs from the 4K frame's by a few per cent -----------
    same_frame = None
    if world > 1 and rank == 0:
        from vkrt_amd.renderer import whole_image_shard

        wsh = whole_image_shard(W, H)
        whole = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        r.reserve(wsh, stream)
        render(0, 1, sh=wsh, img=whole)
        torch.cuda.synchronize(dev)
        r.reset_counters(stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        render(1, 2, sh=wsh, img=whole)
        torch.cuda.synchronize(dev)
        dts = time.perf_counter() - t0
        cs = r.counters()
        same_frame = {"Mrays_s": (cs["rays_closest"] + cs["rays_shadow"]) / dts / 1e6, "ms_per_frame": dts / 2 * 1e3, "frames": 2}
        del whole

    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    rr = torch.tensor([float(rays_local)], dtype=torch.float64, device=coll_dev)
    per_rank = torch.tensor([local_s / args.steps * 1e3], dtype=torch.float64, device=coll_dev)
    all_ms = [per_rank]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        all_ms = [torch.zeros_like(per_rank) for _ in range(world)]
        dist.all_gather(all_ms, per_rank)
    elapsed = float(t.item())
    rays_total = float(rr.item())
    rank_ms = [float(x.item()) for x in all_ms]

    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        key = workload_key(args, W, H, info["triangles"])
        out = {
            "metric": "Mrays/s",
            "value": mrays,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{scene_text} {W}x{H}, {args.spp} spp/frame, depth {args.depth}, "
                            f"{lights} lights, progressive frames"
                            + (f" = BASELINE config 4 frame sharded over {world} GPUs" if world > 1 and (W, H) == (3840, 2160) else ""),
                "width": W, "height": H, "spp": args.spp, "depth": args.depth, "triangles": info["triangles"],
                "bvh": args.build, "bvh_nodes": accel["node_count"], "bvh_depth": accel["max_depth"],
                "parallelism": f"image strips x{world} (16 rows, round-robin) + one all_gather per library call (overlapped with the next call)" if world > 1 else "single GPU",
                "frames_per_call": per_call, "frames_in_flight": r.get_option(abi.VKRT_OPT_WF_FRAMES_IN_FLIGHT),
                "rays_per_step": rays_total / args.steps,
                "upload_ms": r.upload_ms, "builds": builds,
                "per_rank_ms_per_step": rank_ms, "imbalance_max_over_mean": max(rank_ms) / (sum(rank_ms) / len(rank_ms)),
                "mode": mode, "source_hash": vkrt_amd.source_hash(), "workload_key": key,
            },
        }
        if same_frame:
            out["config"]["same_frame_single_gpu_Mrays_s"] = same_frame["Mrays_s"]
            out["config"]["same_frame_single_gpu_ms_per_frame"] = same_frame["ms_per_frame"]
            out["config"]["efficiency_vs_same_frame"] = mrays / (world * same_frame["Mrays_s"])
            out["config"]["same_frame_note"] = ("rank 0 rendered the whole frame alone outside the timed region (2 frames in one call): value / (n_gpus x this) is the "
                                                "strong-scaling efficiency of THIS frame; the N = 1 line of the contract is BASELINE config 3 (1920x1080), another frame")
        # ---- roofline of the dominant kernel ------------------------------------------------------------------------
        if mode == "wavefront" and trav_launches and trav_launches[-1] > 0:
            n_l = float(np.mean(trav_launches))
            per_launch_ms = float(np.mean(trav_ms)) / n_l
            rays_launch = rays_frame / n_l
            # the kernel's own data: 80-B wide nodes / 64-B BVH2 nodes, 48-B triangle records, ray record in (2 float4),
            # hit record out (1 float4 + the 16-B shading record of closest hits)
            node_b = 80 if accel["node_bytes"] and accel["node_bytes"] // max(accel["node_count"], 1) >= 80 else 64
            own = (node_b * work["nodes_visited"] + 48 * work["tris_tested"] + 32 * rays_frame + 16 * rays_frame + 16 * work["hits"]) / n_l
            own_gbs = own / (per_launch_ms * 1e-3) / 1e9
            common = {
                "kernel": "k_wf_traverse", "kernel_ms": per_launch_ms, "launches_per_frame": n_l, "rays_per_launch": rays_launch,
                "per_ray": {"nodes_visited": work["nodes_visited"] / rays_frame, "tris_tested": work["tris_tested"] / rays_frame,
                            "node_step_lane_efficiency": work["nodes_visited"] / max(64 * work["wave_node_steps"], 1),
                            "tri_step_lane_efficiency": work["tris_tested"] / max(64 * work["wave_tri_steps"], 1)},
                "frame": {"ms": float(np.mean(frame_ms)), "traverse_ms": float(np.mean(trav_ms))},
                # sub-blocks: other ways of pricing the same launch; none of them bounds this kernel
                "hbm_own_bytes": {"achieved": own_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": own_gbs / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_launch": own, "algorithmic_bytes_per_ray": own / rays_launch,
                                  "note": "NOT a bound: what the kernel's algorithm fetches from its own data structure (80 B per 8-wide node visited + 48 B "
                                          "per triangle tested + ray / hit records, counted by the kernel) against the HBM peak; the tree is cache-resident, so "
                                          "these bytes come from L2 / Infinity Cache (see traffic)"},
                "l2_gather": {"achieved": own_gbs, "peak": L2_GATHER_GBS, "unit": "GB/s", "frac": own_gbs / L2_GATHER_GBS,
                              "note": "NOT a bound: the node array and most of the triangle array are served by the XCDs' L2s: own bytes against the "
                                      "guide's measured L2 gather rate"},
                "note": "kernel_ms = mean duration of un-overlapped k_wf_traverse launches (timing pass, one lane); value is measured with the "
                        "frames of a call in flight.  The kernel is bound by VALU issue (about 40 % of its instructions are half-rate opcodes: "
                        "issue_mix prices the ceiling that mix allows); HBM moves 0.3x the algorithmic bytes (traffic).",
            }
            # the binding resource: VALU issue.  Instructions per ray come from a separate --pmc SQ_INSTS_VALU pass (profiles/pmc_issue.json)
            roof = None
            try:
                mb = json.load(open(os.path.join(ROOT, "profiles", "r02_issue_microbench.json")))
                peak = max(v["8"]["G_wave_instr_per_s"] for k, v in mb.items() if k in ("v_fma_f32", "v_mul_f32", "v_add_f32"))
                half = float(np.mean([mb[k]["8"]["G_wave_instr_per_s"] for k in ("v_cvt_f32_ubyte0", "v_max3_f32", "v_min3_f32", "v_cmp_lt_f32")]))
                pi, stale = fresh_profile(os.path.join(ROOT, "profiles", "pmc_issue.json"), key), False
                if pi is None:  # measured on other sources or another workload: still the best figure there is -- used, and flagged
                    pi, stale = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json"))), True
                instr = pi["valu_wave_instr_per_ray"] * rays_launch
                got = instr / (per_launch_ms * 1e-3) / 1e9
                roof = {"bound": "valu-issue", "achieved": got, "peak": ISSUE_NOMINAL_G, "unit": "G wave-instr/s", "frac": got / ISSUE_NOMINAL_G,
                        "peak_calibrated": peak, "frac_calibrated": got / peak, "traffic": None,
                        "calibration_note": "frac_calibrated prices the same rate against the full-rate issue rate a pure-VALU microbenchmark sustained (clock "
                                            "throttled to 1.5-1.8 GHz under that load); this kernel's own counter pass runs at the clock in "
                                            "profiles/pmc_issue.json `kernel_clock_ghz` (~2.3-2.4 GHz), so the guide-derived peak is the honest denominator",
                        "kernel_clock_ghz": pi.get("kernel_clock_ghz"),
                        "valu_wave_instr_per_launch": instr, "valu_wave_instr_per_ray": pi["valu_wave_instr_per_ray"], "pmc_stale": stale,
                        "source": "profiles/pmc_issue.json (SQ_INSTS_VALU per ray" + (", measured on OTHER sources / workload: " + str(pi.get("workload"))
                                  if stale else ", same sources and workload") + ") x rays per launch of this run",
                        "peak_source": "MI355X_MICROARCH.md: 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per full-rate wave64 VALU instruction (= 157.3 TFLOP/s / 128 flop); "
                                       "peak_calibrated: profiles/r02_issue_microbench.json, best full-rate opcode class at 8 waves/SIMD; v_cvt_f32_ubyte*, "
                                       f"v_min/max(3)_f32, v_cmp, VOP3 integer ops issue at half the full rate ({half:.0f} G/s measured)"}
                try:  # the ceiling the opcode mix allows (an estimate, kept apart from frac)
                    im = json.load(open(os.path.join(ROOT, "profiles", "isa_mix.json")))
                    if im.get("source_hash") == vkrt_amd.source_hash() and not stale and work["wave_node_steps"]:
                        roof["issue_mix"] = issue_mix_block(im, pi["valu_wave_instr_per_ray"], work["wave_node_steps"] / rays_frame, roof["frac"])
                except Exception:
                    pass
            except Exception:
                roof = None
            if roof is None:  # no issue profile at all: fall back to the byte view, labelled as what it is
                roof = {"bound": "hbm", "achieved": own_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": own_gbs / HBM_PEAK_GBS, "traffic": None,
                        "fallback": "no VALU-issue profile (profiles/pmc_issue.json) available: own algorithmic bytes against the HBM peak; not a bound"}
            roof.update(common)
            pm, tstale = fresh_profile(os.path.join(ROOT, "profiles", "pmc_traffic.json"), key), False
            if pm is None:
                try:
                    pm, tstale = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))), True
                except Exception:
                    pm = None
            if pm:
                roof["traffic"] = pm["hbm_bytes_per_ray"] * rays_launch
                tg = roof["traffic"] / (per_launch_ms * 1e-3) / 1e9
                roof["traffic_detail"] = {"read_bytes_per_ray_raw": pm["read_bytes_per_ray_raw"], "write_bytes_per_ray": pm["write_bytes_per_ray"],
                                          "achieved": tg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tg / HBM_PEAK_GBS, "pmc_stale": tstale,
                                          "over_algorithmic_bytes": roof["traffic"] / own,
                                          "source": "profiles/pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes, reads x2 per the guide's gfx950 note)"}
            # ---- the second kernel of the frame: k_wf_shade (raytrace.rchit / .rmiss, rgen:79-120).  What it moves: path records in and
            # out (sequential, counted from the stream counts) + the hit shader's gathers; priced against the HBM peak with the fabric-side
            # bytes of the separate --pmc FETCH_SIZE / WRITE_SIZE passes, raw and with the guide's x2 on the read side
            if shade_launch_ms and pm and "k_wf_shade" in pm.get("kernels", {}) and "read_bytes_per_ray_raw" in pm["kernels"]["k_wf_shade"]:
                sk = pm["kernels"]["k_wf_shade"]
                s_ms = float(np.mean(shade_launch_ms))
                rd, wr = sk["read_bytes_per_ray_raw"] * rays_launch, sk["write_bytes_per_ray"] * rays_launch
                nP = cnt.get("pair_records", 0) / args.steps
                nC, nS = cnt["rays_closest"] / args.steps - nP, cnt["rays_shadow"] / args.steps - nP
                rec_in = (96 * nC + 80 * nS + 112 * nP) / n_l
                rec_out = (80 * max(nC - cnt["pixels"] / args.steps, 0.0) + 96 * nS + 112 * nP + 16 * cnt["pixels"] / args.steps) / n_l
                g = lambda b: b / (s_ms * 1e-3) / 1e9
                roof["shade"] = {
                    "kernel": "k_wf_shade", "bound": "hbm", "kernel_ms": s_ms, "records_per_launch": (nC + nS + nP) / n_l,
                    "achieved": g(rd + wr), "achieved_reads_x2": g(2 * rd + wr), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": g(rd + wr) / HBM_PEAK_GBS, "frac_reads_x2": g(2 * rd + wr) / HBM_PEAK_GBS,
                    "traffic": rd + wr, "traffic_reads_x2": 2 * rd + wr, "read_bytes_per_launch_raw": rd, "write_bytes_per_launch": wr,
                    "record_bytes_in_per_launch": rec_in, "record_bytes_out_per_launch": rec_out,
                    "record_GBs": g(rec_in + rec_out), "record_frac": g(rec_in + rec_out) / HBM_PEAK_GBS, "pmc_stale": tstale,
                    "note": "kernel_ms = mean time between the end of a traversal launch and the start of the next one in the timing pass (one lane): "
                            "the k_wf_shade launch between them plus the launch gaps.  record bytes: 96 / 80 / 112 B read and 80 / 96 / 112 B written per "
                            "closest / shadow / pair record (wf_streams.h), counts from the stream counters of the timed region; traffic: "
                            "profiles/pmc_traffic.json per ray x rays per launch of this run (fabric side of L2, Infinity-Cache hits included)"}
                fr = pm.get("frame")
                if fr:
                    fb_raw = (fr["read_bytes_per_ray_raw"] + fr["write_bytes_per_ray"]) * rays_frame
                    fb_x2 = (2 * fr["read_bytes_per_ray_raw"] + fr["write_bytes_per_ray"]) * rays_frame
                    step_s = elapsed / args.steps
                    roof["frame_hbm"] = {"achieved": fb_raw / step_s / 1e9, "achieved_reads_x2": fb_x2 / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": fb_raw / step_s / 1e9 / HBM_PEAK_GBS, "frac_reads_x2": fb_x2 / step_s / 1e9 / HBM_PEAK_GBS,
                                         "bytes_per_frame_raw": fb_raw, "pmc_stale": tstale,
                                         "note": "achieved HBM GB/s of the whole frame (north star): FETCH_SIZE + WRITE_SIZE of every kernel of the frame per ray "
                                                 "(separate --pmc passes) x the rays of a timed step / ms_per_step; reads raw and x2 (guide's gfx950 correction)"}
            out["roofline"] = roof
        elif frame_ms:
            own = (64 * work["nodes_visited"] + 48 * work["tris_tested"]) if work["nodes_visited"] else 0
            ach = own / (float(np.mean(frame_ms)) * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "k_pathtrace", "kernel_ms": float(np.mean(frame_ms)), "algorithmic_bytes_per_launch": own}

        # ---- the builders that were not selected: a short run each beside the headline ---------------------------------
        if world == 1 and not args.no_other_builder:
            for other in others:
                r.build(other)
                for f in range(1):
                    step(f)
                dt, _, c2 = timed_frames(1, 2)
                builds[other]["Mrays_s"] = (c2["rays_closest"] + c2["rays_shadow"]) / dt / 1e6
            builds[args.build]["Mrays_s"] = mrays
            r.build(args.build)
            # the same building with artist-like tessellation (room-sized wall triangles, 15-m needles, drapery strips; tools/atrium.py
            # variant "nonuniform"): BASELINE config 3 is Sponza, whose geometry looks like this, not like uniform grids
            if not args.gltf and args.variant == "default":
                flat2, info2 = atrium.build_atrium(args.triangles, seed=args.scene_seed, with_textures=not args.no_textures, variant="nonuniform")
                r2 = Renderer(flat2, device=local_rank, build=args.build)
                r2.reserve(shard, stream)
                img2 = torch.zeros_like(image)
                render(0, 1, rr=r2, img=img2)
                sync()
                r2.reset_counters(stream)
                sync()
                t0 = time.perf_counter()
                render(1, 4, rr=r2, img=img2)
                torch.cuda.synchronize(dev)
                dt2 = time.perf_counter() - t0
                c2 = r2.counters()
                nu = (c2["rays_closest"] + c2["rays_shadow"]) / dt2 / 1e6
                out["config"]["nonuniform_variant"] = {"Mrays_s": nu, "ratio_to_headline": nu / mrays, "triangles": info2["triangles"], "ms_per_step": dt2 / 4 * 1e3,
                                                       "note": "same frame, same builder, 4 steps in one call: tools/atrium.py variant nonuniform (Sponza-like tessellation)"}
                r2.close()
                del img2

        # ---- CPU oracle on a bounded sample: reported CPU baseline + the SURVEY 8d contract accounting ------------------
        if not args.no_cpu_baseline and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))  # test infrastructure: only this leg touches oracle/
            import oracle_py

            orc = oracle_py.OracleScene(flat, build_bvh=True, max_leaf=4)
            frame = args.warmup  # the first timed frame
            pc = make_push_constants(samples=args.spp, depth=args.depth, frame=frame, lights_count=lights)
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            quota = cpu_quota()  # CPUs the container's cgroup lets this process run at once (None = no limit below the affinity mask)
            # every hardware thread this process may USE (SURVEY 8d "all hardware threads"): the affinity mask of a 1-GPU box shows all
            # 256 threads of the host while its cgroup grants ~16 CPUs -- 240 more threads only add context switches
            threads = max(1, min(avail, quota)) if quota else avail

            def sample(nthreads, seconds):
                """rows of the frame rendered by the oracle for about `seconds` of wall time on `nthreads` threads"""
                probe_rows = np.linspace(0, H - 1, max(2, min(4, nthreads))).astype(np.uint32)
                buf = np.zeros((len(probe_rows), W, 4), np.float32)
                tp = time.perf_counter()
                orc.render(pc, cam, W, H, seed=frame, rows=probe_rows, image=buf, threads=nthreads)
                per_row = (time.perf_counter() - tp) / len(probe_rows)
                # the oracle deals 32-pixel chunks of the rows dynamically: at least 8 chunks per thread, so that no thread's last chunk
                # is the wall time of the sample
                min_rows = -(-8 * nthreads // max(1, -(-W // 32)))
                nrows = int(max(min_rows, 2, min(H, seconds / max(per_row, 1e-6))))
                for attempt in range(2):  # the probe's first rows run cold and overestimate the row time: rescale once if the sample fell short
                    rows = np.unique(np.linspace(0, H - 1, nrows).astype(np.uint32))
                    buf = np.zeros((len(rows), W, 4), np.float32)
                    tp, tt = time.perf_counter(), os.times()
                    _, c = orc.render(pc, cam, W, H, seed=frame, rows=rows, image=buf, threads=nthreads)
                    cpu_s = time.perf_counter() - tp
                    te = os.times()
                    c["cpu_seconds_per_wall_second"] = ((te.user + te.system) - (tt.user + tt.system)) / max(cpu_s, 1e-9)
                    if cpu_s >= 0.6 * seconds or len(rows) >= H:
                        break
                    nrows = int(min(H, len(rows) * seconds / max(cpu_s, 1e-6)))
                return rows, c, cpu_s

            rows, c, cpu_s = sample(threads, 0.8 * args.cpu_seconds)
            rows1, c1, cpu_s1 = sample(1, 0.5 * args.cpu_seconds)
            cpu_rays = c["rays_closest"] + c["rays_shadow"]
            cpu_rays1 = c1["rays_closest"] + c1["rays_shadow"]
            t16 = None
            if threads != 16 and avail >= 16:  # the share of the host a 1-GPU box is sized for (round 1-3 figures were taken on it)
                rows16, c16, cpu_s16 = sample(16, 0.5 * args.cpu_seconds)
                t16 = {"value": (c16["rays_closest"] + c16["rays_shadow"]) / cpu_s16 / 1e6, "unit": "Mrays/s", "cores": 16,
                       "sample": f"{len(rows16)} rows ({c16['rays_closest'] + c16['rays_shadow']} rays, {cpu_s16:.1f} s)"}
            out["cpu_baseline"] = {
                "value": cpu_rays / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"{len(rows)} evenly spaced rows of frame {frame} of the same {W}x{H} workload ({cpu_rays} rays, {cpu_s:.1f} s), "
                          f"full-sweep SAH BVH2 <=4 tris/leaf",
                "single_thread": {"value": cpu_rays1 / cpu_s1 / 1e6, "unit": "Mrays/s", "cores": 1,
                                  "sample": f"{len(rows1)} rows ({cpu_rays1} rays, {cpu_s1:.1f} s)"},
                "cpu_seconds_per_wall_second": round(c["cpu_seconds_per_wall_second"], 1),  # CPUs the sample really kept busy
                "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(), "cpus_available_to_this_process": avail, "cgroup_cpu_quota": quota,
            }
            if t16:
                out["cpu_baseline"]["threads_16"] = t16
            if "roofline" in out and out["roofline"].get("kernel") == "k_wf_traverse":
                tb = (64 * c["nodes_visited"] + 48 * c["tris_tested"]) / cpu_rays
                rl = out["roofline"]
                cg = tb * rl["rays_per_launch"] / (rl["kernel_ms"] * 1e-3) / 1e9
                rl["contract"] = {"bytes_per_ray_traversal": tb, "bytes_per_ray_all": oracle_py.algorithmic_bytes(c, frame_gt0=frame > 0) / cpu_rays,
                                  "achieved": cg, "peak": HBM_PEAK_GBS, "frac": cg / HBM_PEAK_GBS,
                                  "note": "NOT a bound: SURVEY 8d accounting, 64 B per node visit + 48 B per triangle test of the ORACLE's binary tree on the same "
                                          "rays.  Not the kernel's data structure (an 8-wide compressed tree needs ~0.6x the bytes), so this fraction exceeds 1"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()

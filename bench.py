#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing path (BASELINE.json metric: Mrays/s +
achieved HBM GB/s, Sponza 1080p 8-bounce).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one vkrt_pathtrace launch = one progressive frame of the workload: the seeded
procedural Sponza-class atrium (tools/atrium.py; the real Sponza.gltf is not available offline),
1920x1080, 16 spp per frame (PushConstantRay.samples = 16), depth 8, 8 fallback lights, frame
index = step index (frames > 0 jitter and blend into the resident rgba32f image,
raytrace.rgen:44,136-141).  Scene, BVH and image are resident in HBM before the timed region.

Multi-GPU (weak scaling): every rank holds the whole scene and renders its 16-row strips of a
16:9 image with N x 1080p pixels (N=4 is exactly 3840x2160, BASELINE config 4); each step ends with
one RCCL all_gather of the strips.  Mrays/s counts the closest-hit + shadow traceRay calls actually
issued (device counters), summed over ranks.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline     : algorithmic bytes per launch / mean kernel time (HIP events on the launch stream)
                 against the 8 TB/s HBM3E peak; algorithmic bytes = SURVEY.md 8(d) per-ray figure
                 from the instrumented CPU oracle on a bounded pixel sample x rays per launch.
  cpu_baseline : the CPU oracle (scalar C++ restatement, kind "port") timed on the same sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CU_COUNT, CLOCK_GHZ = 256, 2.4  # MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def image_size(n_gpus, base_w, base_h):
    """16:9 image with n_gpus x (base_w x base_h) pixels, width a multiple of 8."""
    if n_gpus <= 1:
        return base_w, base_h
    import math

    w = int(round(base_w * math.sqrt(n_gpus) / 8.0)) * 8
    h = int(round(w * base_h / base_w))
    return w, h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--triangles", type=int, default=262144)
    ap.add_argument("--scene-seed", type=int, default=1)
    ap.add_argument("--build", choices=["sah", "lbvh"], default="sah")
    ap.add_argument("--no-textures", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the oracle sample")
    ap.add_argument("--fixed-size", action="store_true", help="keep --width/--height for every N (strong scaling)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI (default); gloo for rehearsals")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="all ranks use cuda:0 (control-flow rehearsal on a 1-GPU box; needs --backend gloo)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import vkrt_amd  # noqa: F401
    from vkrt_amd.flat_scene import make_push_constants, uniforms_from_matrices
    from vkrt_amd.renderer import Renderer
    from vkrt_amd.sharding import gather_image, make_shard, shard_row_indices
    import atrium
    import camera_np

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
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

    W, H = (args.width, args.height) if args.fixed_size else image_size(world, args.width, args.height)
    flat, info = atrium.build_atrium(args.triangles, seed=args.scene_seed, with_textures=not args.no_textures)
    cam = uniforms_from_matrices(*camera_np.global_uniforms(width=W, height=H, **atrium.DEFAULT_CAMERA))
    lights = int(flat.lights.shape[0])

    r = Renderer(flat, device=local_rank, build=args.build)
    accel = r.accel_info()
    shard = make_shard(W, H, world, rank)
    rows_local = r.shard_rows(shard)
    image = torch.zeros((rows_local, W, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step(frame, flags=0):
        pc = make_push_constants(samples=args.spp, depth=args.depth, frame=frame, lights_count=lights)
        r.pathtrace(pc, cam, W, H, seed=frame, flags=flags, shard=shard, image=image, stream=stream)
        if world > 1:
            # one gather per frame, overlapped with the next frame's kernels (side stream); the timed region ends with a
            # device-wide synchronisation, so the last gathered image is complete inside it
            return gather_image(image if args.backend == "nccl" else image.cpu(), H, world, rank, overlap=args.backend == "nccl")
        return image

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for f in range(args.warmup):
        step(f)
    sync()
    r.reset_counters(stream)
    kernel_ms = []
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
        if world == 1:
            pass
    sync()
    elapsed = time.perf_counter() - t0
    # per-launch kernel time from the HIP events recorded on the launch stream (last launch), and an
    # extra untimed pass that reads every launch's events individually
    cnt = r.counters()
    rays_local = cnt["rays_closest"] + cnt["rays_shadow"]
    from vkrt_amd import abi as _abi

    frame_ms, trav_ms, trav_launches = [], [], []
    for k in range(min(args.steps, 3)):
        step(args.warmup + args.steps + k, flags=_abi.VKRT_TRACE_TIME_KERNELS)
        torch.cuda.synchronize(dev)
        tm = r.last_trace_timing()
        frame_ms.append(tm["total_ms"])
        trav_ms.append(tm["traverse_ms"])
        trav_launches.append(tm["traverse_launches"])
    mode = tm["mode"]
    kernel_ms_mean = float(np.mean(frame_ms))
    cnt_after = r.counters()
    rays_extra = (cnt_after["rays_closest"] + cnt_after["rays_shadow"]) - rays_local
    rays_per_launch_local = rays_extra / max(1, len(frame_ms))
    t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
    rr = torch.tensor([float(rays_local)], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays_total = float(rr.item())

    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        out = {
            "metric": "Mrays/s",
            "value": mrays,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.fixed_size else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"procedural Sponza-class atrium ({info['triangles']} tris, seed {info['seed']}, "
                            f"{'textured' if not args.no_textures else 'untextured'}) {W}x{H}, {args.spp} spp/frame, depth {args.depth}, "
                            f"{lights} fallback lights, progressive frames",
                "width": W, "height": H, "spp": args.spp, "depth": args.depth, "triangles": info["triangles"],
                "bvh": args.build, "bvh_nodes": accel["node_count"], "bvh_depth": accel["max_depth"],
                "parallelism": f"image strips x{world} (16 rows, round-robin) + all_gather per frame (overlapped with the next frame)" if world > 1 else "single GPU",
                "rays_per_step": rays_total / args.steps,
            },
        }
        # ---- CPU oracle on a bounded sample: per-ray algorithmic bytes + reported CPU baseline -----
        bytes_per_ray = None
        if not args.no_cpu_baseline and world == 1:
            import oracle_py

            orc = oracle_py.OracleScene(flat, build_bvh=True, max_leaf=4)
            frame = args.warmup  # the first timed frame
            pc = make_push_constants(samples=args.spp, depth=args.depth, frame=frame, lights_count=lights)
            threads = min(16, os.cpu_count() or 1)  # the 1-GPU box grants 16 CPUs
            # calibrate: one strip-spread row set, then scale the row count to ~cpu-seconds
            probe_rows = np.linspace(0, H - 1, 4).astype(np.uint32)
            buf = np.zeros((len(probe_rows), W, 4), np.float32)
            tp = time.perf_counter()
            orc.render(pc, cam, W, H, seed=frame, rows=probe_rows, image=buf, threads=threads)
            per_row = (time.perf_counter() - tp) / len(probe_rows)
            nrows = int(max(threads, min(H, args.cpu_seconds / max(per_row, 1e-6))))
            rows = np.unique(np.linspace(0, H - 1, nrows).astype(np.uint32))
            buf = np.zeros((len(rows), W, 4), np.float32)
            tp = time.perf_counter()
            _, c = orc.render(pc, cam, W, H, seed=frame, rows=rows, image=buf, threads=threads)
            cpu_s = time.perf_counter() - tp
            cpu_rays = c["rays_closest"] + c["rays_shadow"]
            bytes_per_ray = oracle_py.algorithmic_bytes(c, frame_gt0=frame > 0) / cpu_rays
            trav_bytes_per_ray = (64 * c["nodes_visited"] + 48 * c["tris_tested"]) / cpu_rays
            out["cpu_baseline"] = {
                "value": cpu_rays / cpu_s / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"{len(rows)} evenly spaced rows of frame {frame} of the same {W}x{H} workload ({cpu_rays} rays, {cpu_s:.1f} s), "
                          f"full-sweep SAH BVH2 <=4 tris/leaf",
            }
            out["config"]["algorithmic_bytes_per_ray"] = bytes_per_ray
            out["config"]["algorithmic_bytes_per_ray_traversal_only"] = trav_bytes_per_ray
        if bytes_per_ray is None:
            # committed per-config fixture (tests/golden/algbytes.json) when the oracle leg is skipped
            try:
                fx = json.load(open(os.path.join(ROOT, "tests", "golden", "algbytes.json")))
                bytes_per_ray = float(fx["atrium262k_1080p_16spp_d8"]["bytes_per_ray"])
            except Exception:
                bytes_per_ray = None
        if bytes_per_ray is not None:
            # frame level: all algorithmic bytes of the frame over the frame's GPU time (HIP events on the launch stream)
            alg_bytes_frame = bytes_per_ray * rays_per_launch_local
            frame_gbs = alg_bytes_frame / (kernel_ms_mean * 1e-3) / 1e9
            traffic = None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                traffic = pm.get("hbm_bytes_per_launch")
            except Exception:
                pass
            if mode == "wavefront":
                # dominant kernel = k_wf_traverse: its own algorithmic bytes (nodes + triangles) per
                # launch over its own mean launch duration (HIP events around every launch of it)
                try:
                    tb = trav_bytes_per_ray
                except NameError:
                    tb = float(json.load(open(os.path.join(ROOT, "tests", "golden", "algbytes.json")))
                               ["atrium262k_1080p_16spp_d8"]["traversal_bytes_per_ray"])
                n_l = float(np.mean(trav_launches))
                per_launch_ms = float(np.mean(trav_ms)) / n_l
                alg_launch = tb * rays_per_launch_local / n_l
                achieved = alg_launch / (per_launch_ms * 1e-3) / 1e9
                out["roofline"] = {
                    "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "kernel": "k_wf_traverse", "kernel_ms": per_launch_ms, "launches_per_frame": n_l,
                    "algorithmic_bytes_per_launch": alg_launch, "rays_per_launch": rays_per_launch_local / n_l,
                    "frame": {"ms": kernel_ms_mean, "traverse_ms": float(np.mean(trav_ms)), "algorithmic_bytes": alg_bytes_frame,
                              "achieved_GBs": frame_gbs, "frac": frame_gbs / HBM_PEAK_GBS},
                    "note": "algorithmic bytes follow the SURVEY 8d contract (canonical BVH2: 64 B per node visit + 48 B per triangle test of the "
                            "oracle's tree), not the kernel's own layout; the 8-wide compressed tree moves ~5x less (traffic) and is served from "
                            "L2, so frac can exceed 1 and HBM is not the binding resource -- VALU issue is (see issue). kernel_ms is the mean "
                            "duration of un-overlapped k_wf_traverse launches (timing pass, one sub-frame); value is measured with the "
                            "default two-sub-frame pipeline.",
                }
                try:
                    pi = json.load(open(os.path.join(ROOT, "profiles", "pmc_issue.json")))
                    instr = float(pi["valu_wave_instr_per_launch"])
                    peak = CU_COUNT * CLOCK_GHZ  # one VALU wave-instruction per CU per clock (4 SIMD16 x 4 cycles per wave64 op)
                    got = instr / (per_launch_ms * 1e-3) / 1e9
                    out["roofline"]["issue"] = {"bound": "valu-issue", "achieved": got, "peak": peak, "unit": "G wave-instr/s", "frac": got / peak,
                                                "valu_wave_instr_per_launch": instr, "source": "profiles/pmc_issue.json"}
                except Exception:
                    pass
            else:
                out["roofline"] = {
                    "bound": "hbm", "achieved": frame_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frame_gbs / HBM_PEAK_GBS,
                    "traffic": traffic, "kernel": "k_pathtrace", "kernel_ms": kernel_ms_mean,
                    "algorithmic_bytes_per_launch": alg_bytes_frame, "rays_per_launch": rays_per_launch_local,
                }
            out["config"]["mode"] = mode
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()

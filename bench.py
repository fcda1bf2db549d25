#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the trace loop on the reference's teapot.h scene at 1920x1080.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one render() of the frame: every rank traces its round-robin row
bands on its own MI355X and (N > 1) the 8-bit rows are gathered to rank 0 with
RCCL.  The scene is uploaded before the timed region; nothing crosses PCIe
inside it.  Rank 0 prints one JSON line.

  value        total rays (primary + shadow + secondary, the reference's
               intersect_world call count) of all ranks / wall time, Mrays/s
  roofline     the trace kernel against the FP32 vector peak (the path is
               VALU-bound: the scene lives in SGPRs/caches, 3 bytes per pixel
               are written); algorithmic FLOPs = rays x primitives x 45
  cpu_baseline the reference itself (oracle/_ref, gcc -O2, one core) on the
               same workload, timed on this host
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import raytracer_amd as R  # noqa: E402
from raytracer_amd.dist import BandGather, default_band_rows  # noqa: E402

WIDTH, HEIGHT = 1920, 1080
SCENE = "cube"  # render_some_triangles + the teapot.h mesh (12 triangles), RayTracer.c:6727
FLOP_PER_TRI_TEST = 45  # full Moeller-Trumbore path, RayTracer.c:943-966
FLOP_PER_SPHERE_TEST = 63
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32 peak


def cpu_baseline(rays_expected):
    """The reference (or, failing that, the CPU port) on the same frame, one core of this host."""
    ref = os.path.join(ROOT, "oracle", "_ref", "rt_ref_harness_timing")
    if os.path.exists(ref):
        out = subprocess.run([ref, "--scene", SCENE, "--size", str(WIDTH), str(HEIGHT)], capture_output=True,
                             text=True, timeout=600)
        if out.returncode == 0:
            info = json.loads(out.stdout.strip().splitlines()[-1])
            return {"value": round(rays_expected / info["seconds"] / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                    "kind": "reference", "seconds": round(info["seconds"], 3),
                    "sample": f"whole workload: {SCENE} {WIDTH}x{HEIGHT}, {rays_expected} rays, gcc -O2 build of the "
                              "unmodified reference source"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    rows = 135  # 1/8 of the frame: bands of 1 row, rank 0 of 8
    s = R.stock_scene(SCENE, WIDTH, HEIGHT)
    t0 = time.perf_counter()
    _, st = O.render_rgb(s, n_ranks=8, band_rows=1, rank=0)
    dt = time.perf_counter() - t0
    return {"value": round(st.rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
            "seconds": round(dt, 3), "sample": f"every 8th row ({rows} rows) of {SCENE} {WIDTH}x{HEIGHT}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus}: launch with python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py ...")
        a.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    hip = R.HipRenderer(local_rank)
    scene = R.stock_scene(SCENE, WIDTH, HEIGHT)
    hip.upload(scene)
    band_rows = default_band_rows(HEIGHT, world)
    bg = BandGather(HEIGHT, WIDTH, band_rows, world, rank, dev)
    p = R.make_params(max_depth=5, spp_grid=1, accel=R._ffi.RT_ACCEL_AUTO, band_rows=band_rows, n_ranks=world, rank=rank)
    stream = torch.cuda.current_stream().cuda_stream

    # rays of this rank's rows (counting variant of the kernel, outside the timed region)
    pc = R.make_params(max_depth=5, count_rays=True, band_rows=band_rows, n_ranks=world, rank=rank)
    st = R._ffi.Stats()
    hip.render_device(bg.local.data_ptr(), pc, stream=stream, stats=st)
    my_rays = torch.tensor([st.rays, st.prim_tests], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(my_rays)
    total_rays, _ = [int(x) for x in my_rays.tolist()]
    accel_used = "bvh" if st.accel_used == R._ffi.RT_ACCEL_BVH else "brute"

    def step():
        hip.render_device(bg.local.data_ptr(), p, stream=stream)
        return bg.gather()

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.timer_start(stream)
    for _ in range(a.steps):
        step()
    hip.timer_stop(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = hip.timer_ms()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        n_prim_flops = scene.n_tris * FLOP_PER_TRI_TEST + scene.n_spheres * FLOP_PER_SPHERE_TEST
        # per launch of THIS rank's kernel: its rays x every primitive (brute-force definition)
        flops_per_launch = st.rays * n_prim_flops
        kernel_ms = ev_ms / a.steps  # HIP events on the launch stream around the K launches
        achieved = flops_per_launch / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "Mrays/s (primary+shadow) at 1920x1080 teapot scene; % of FP32 VALU peak",
            "value": round(total_rays * a.steps / dt / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"teapot.h mesh (12 triangles), render_some_triangles camera/light, {WIDTH}x{HEIGHT}, "
                                   "1 spp, depth 5", "rays_per_frame": total_rays, "accel": accel_used,
                       "parallelism": f"row bands of {band_rows} rows round-robin over {world} rank(s)"
                                      + (", one RCCL gather to rank 0 per frame" if world > 1 else "")},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4), "traffic": None,
                         "kernel": "rt_trace_kernel", "kernel_ms": round(kernel_ms, 5),
                         "flops_per_launch": flops_per_launch,
                         "note": "algorithmic FLOPs = rays x 12 triangles x 45; arithmetic is FP64 without FMA "
                                 "(parity with the reference), priced against the FP32 vector peak"},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(total_rays)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    hip.shutdown()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the trace loop on the reference's teapot.h scene at 1920x1080.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one render() of the frame: every rank traces its round-robin row
bands on its own MI355X and (N > 1) the 8-bit rows are exchanged with RCCL so
that frame f is complete on rank f % N (one all-to-all per group of frames).  The scene is uploaded before the timed region; nothing crosses PCIe
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
from raytracer_amd.dist import FrameExchange, default_band_rows  # noqa: E402

WIDTH, HEIGHT = 1920, 1080
SCENE = "cube"  # render_some_triangles + the teapot.h mesh (12 triangles), RayTracer.c:6727
# algorithmic FLOPs per unit of work (the figures the survey fixed for this path)
FLOP_PER_TRI_TEST = 45      # full Moeller-Trumbore path, RayTracer.c:943-966
FLOP_PER_SPHERE_TEST = 63   # two affine mat-vec + quadratic, RayTracer.c:898-940
FLOP_PER_BOX_TEST = 18      # slab test (BVH node or mesh bounds)
FLOP_PER_PRIMARY_RAY = 30   # ray_for_pixel, RayTracer.c:1535
FLOP_PER_SHADED_HIT = 80    # prepare_computations + lighting (+ 1 pow), RayTracer.c:1360, 1273
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32 peak (256 CU x 128 lanes x 2 x 2.4 GHz)
PEAK_FP64_NO_FMA_TFLOPS = 39.3  # FP64 vector: 78.6 with FMA, half of it for a mul/add-only stream


def pmc_traffic_bytes():
    """HBM bytes per launch of the headline kernel, from the committed summary of the PMC passes (a bench run
    cannot collect hardware counters itself); None when no summary is there."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_cube_1080p.json")
    try:
        d = json.load(open(path))
        return int(d["derived_hbm_read_bytes_corrected_x2"] + d["derived_hbm_write_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def algorithmic_flops(st):
    """FLOPs of the work the kernel actually did in one launch, from its own counters (counting build)."""
    tri_tests = st.prim_tests - st.sphere_tests
    return (st.rays_primary * FLOP_PER_PRIMARY_RAY + st.rays_shadow * FLOP_PER_SHADED_HIT
            + tri_tests * FLOP_PER_TRI_TEST + st.sphere_tests * FLOP_PER_SPHERE_TEST + st.node_tests * FLOP_PER_BOX_TEST)


def side_measurements(hip, dev, steps=100):
    """Context for the headline: the other benchmark scenes at 1920x1080 on this GPU, kernel time from HIP events."""
    out = {}
    buf = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.uint8, device=dev)
    for name, scene, tr, accel in (("spheres_1080p", "spheres", 0.0, R._ffi.RT_ACCEL_AUTO),
                                   ("lighthouse_1080p_bvh", "lighthouse", 0.0, R._ffi.RT_ACCEL_BVH),
                                   ("lighthouse_1080p_exhaustive", "lighthouse", 0.0, R._ffi.RT_ACCEL_BRUTE),
                                   ("lighthouse_1080p_reflective_bvh", "lighthouse", 0.3, R._ffi.RT_ACCEL_BVH)):
        hip.upload(R.stock_scene(scene, WIDTH, HEIGHT, tr))
        st = R._ffi.Stats()
        hip.render_device(buf.data_ptr(), R.make_params(accel=accel, count_rays=True), stats=st)
        p = R.make_params(accel=accel)
        n = steps if accel != R._ffi.RT_ACCEL_BRUTE else 10
        for _ in range(3):
            hip.render_device(buf.data_ptr(), p)
        hip.timer_start()
        for _ in range(n):
            hip.render_device(buf.data_ptr(), p)
        hip.timer_stop()
        ms = hip.timer_ms() / n
        executed = algorithmic_flops(st)
        sc = hip.scene_counts()
        exhaustive = st.rays * (sc[1] * FLOP_PER_TRI_TEST + sc[0] * FLOP_PER_SPHERE_TEST)
        fl = exhaustive if st.accel_used == R._ffi.RT_ACCEL_BRUTE else executed
        out[name] = {"rays": st.rays, "kernel_ms": round(ms, 5), "mrays_s": round(st.rays / ms / 1e3, 1),
                     "tflops_algorithmic": round(fl / ms / 1e9, 3), "tflops_executed": round(executed / ms / 1e9, 3),
                     "tri_tests": st.prim_tests - st.sphere_tests,
                     "sphere_tests": st.sphere_tests, "box_tests": st.node_tests}
    return out


def pipelined_measurements(hip, dev, streams=4, steps=300):
    """Frames issued round-robin on several streams, each into an image of its own: the next frame's waves fill
    the CUs while the last waves of the previous one drain.  Reported beside the headline, which is timed on one
    stream so that a launch's duration and the frame time are the same thing."""
    out = {}
    ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    bufs = [torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.uint8, device=dev) for _ in range(streams)]
    for name, scene, tr, accel in (("cube_1080p", SCENE, 0.0, R._ffi.RT_ACCEL_AUTO),
                                   ("lighthouse_1080p_bvh", "lighthouse", 0.0, R._ffi.RT_ACCEL_BVH),
                                   ("lighthouse_1080p_reflective_bvh", "lighthouse", 0.3, R._ffi.RT_ACCEL_BVH)):
        hip.upload(R.stock_scene(scene, WIDTH, HEIGHT, tr))
        st = R._ffi.Stats()
        hip.render_device(bufs[0].data_ptr(), R.make_params(accel=accel, count_rays=True), stats=st)
        p = R.make_params(accel=accel, frames_overlap=True)
        best = None
        for rep in range(3):
            for i in range(2 * streams):
                hip.render_device(bufs[i % streams].data_ptr(), p, stream=ss[i % streams].cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s_ in ss:
                s_.wait_event(e0)
            for i in range(steps):
                hip.render_device(bufs[i % streams].data_ptr(), p, stream=ss[i % streams].cuda_stream)
            for s_ in ss:
                torch.cuda.current_stream().wait_stream(s_)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            best = ms if best is None else min(best, ms)
        same = all(bool(torch.equal(b, bufs[0])) for b in bufs)
        out[name] = {"streams": streams, "ms_per_frame": round(best, 5), "mrays_s": round(st.rays / best / 1e3, 1),
                     "images_identical": same}
    return out


def general_measurements(hip, dev, steps=20):
    """The reference's other scene functions (planes, patterns, cube, cylinder, glass) through the general kernel."""
    out = {}
    buf = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.uint8, device=dev)
    for scene in ("refraction", "world_with_plane", "dual_spheres"):
        hip.upload(R.stock_scene(scene, WIDTH, HEIGHT))
        st = R._ffi.Stats()
        hip.render_device(buf.data_ptr(), R.make_params(count_rays=True), stats=st)
        p = R.make_params()
        for _ in range(2):
            hip.render_device(buf.data_ptr(), p)
        hip.timer_start()
        for _ in range(steps):
            hip.render_device(buf.data_ptr(), p)
        hip.timer_stop()
        ms = hip.timer_ms() / steps
        out[scene + "_1080p"] = {"rays": st.rays, "kernel_ms": round(ms, 4), "mrays_s": round(st.rays / ms / 1e3, 1)}
    return out


def ppm_measurements(hip, dev, steps=50):
    """The step after the path: write_canvas_to_file's P3 text (RayTracer.c:1333) formatted on the device.
    HBM-bound; algorithmic bytes = the image read twice (lengths, then text) + the text written."""
    out = {}
    for (w, h) in ((WIDTH, HEIGHT), (7680, 4320)):
        img = torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device=dev)
        cap = hip.ppm_capacity(w, h)
        text = torch.empty(cap, dtype=torch.uint8, device=dev)
        n = hip.format_ppm_device(img.data_ptr(), w, h, text.data_ptr(), cap)
        for _ in range(3):
            hip.format_ppm_device(img.data_ptr(), w, h, text.data_ptr(), cap, wait=False)
        hip.timer_start()
        for _ in range(steps):
            hip.format_ppm_device(img.data_ptr(), w, h, text.data_ptr(), cap, wait=False)
        hip.timer_stop()
        ms = hip.timer_ms() / steps
        nbytes = 2 * 3 * w * h + n
        out[f"{w}x{h}"] = {"text_bytes": n, "ms": round(ms, 5), "roofline": {
            "bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
            "frac": round(nbytes / ms / 1e6 / 8000.0, 4), "algorithmic_bytes": nbytes}}
        del img, text
    return out


def cpu_baseline_row_split(ref, rays_expected, cores=None):
    """The same reference build as `cores` worker processes, each rendering row bands of the frame (the reference
    itself has no threads: this is what a user gets from running it once per band).  Wall time of the whole job."""
    from concurrent.futures import ThreadPoolExecutor

    cores = cores or int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    n_bands = cores * 4  # several bands per worker: the lit rows are not spread evenly
    edges = [HEIGHT * i // n_bands for i in range(n_bands + 1)]

    def band(i):
        out = subprocess.run([ref, "--scene", SCENE, "--size", str(WIDTH), str(HEIGHT), "--rows", str(edges[i]),
                              str(edges[i + 1])], capture_output=True, text=True, timeout=600)
        return out.returncode

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        codes = list(pool.map(band, range(n_bands)))
    dt = time.perf_counter() - t0
    if any(codes):
        return None
    return {"value": round(rays_expected / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "seconds": round(dt, 3),
            "sample": f"same frame as {n_bands} row bands over {cores} processes (includes process start-up)"}


def cpu_baseline(rays_expected):
    """The reference (or, failing that, the CPU port) on the same frame, one core of this host."""
    ref = os.path.join(ROOT, "oracle", "_ref", "rt_ref_harness_timing")
    if os.path.exists(ref):
        out = subprocess.run([ref, "--scene", SCENE, "--size", str(WIDTH), str(HEIGHT)], capture_output=True,
                             text=True, timeout=600)
        if out.returncode == 0:
            info = json.loads(out.stdout.strip().splitlines()[-1])
            res = {"value": round(rays_expected / info["seconds"] / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                   "kind": "reference", "seconds": round(info["seconds"], 3),
                   "sample": f"whole workload: {SCENE} {WIDTH}x{HEIGHT}, {rays_expected} rays, gcc -O2 build of the "
                             "unmodified reference source (the reference is single-threaded)"}
            res["row_split_processes"] = cpu_baseline_row_split(ref, rays_expected)
            return res
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
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the other-scenes context measurements")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus}: launch with python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py ...")
        a.gpus = world
    # Rehearsal on a one-GPU box: RT_BENCH_REHEARSE=1 puts every rank on device 0 and gathers through gloo
    # (RCCL refuses two ranks on one device).  The driver's real runs use one GPU per rank and RCCL.
    rehearse = os.environ.get("RT_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    hip = R.HipRenderer(device_index)
    scene = R.stock_scene(SCENE, WIDTH, HEIGHT)
    hip.upload(scene)
    band_rows = default_band_rows(HEIGHT, world)
    # Frame f of the run is put together on rank f % world; a group of 4 * world frames moves in one
    # all-to-all (RCCL) while the next group renders into the other buffer set.
    fx = FrameExchange(HEIGHT, WIDTH, band_rows, world, rank, dev, rounds=4, slots=2)
    p = R.make_params(max_depth=5, spp_grid=1, accel=R._ffi.RT_ACCEL_AUTO, band_rows=band_rows, n_ranks=world, rank=rank)
    stream = torch.cuda.current_stream().cuda_stream

    # rays of this rank's rows (counting variant of the kernel, outside the timed region)
    pc = R.make_params(max_depth=5, count_rays=True, band_rows=band_rows, n_ranks=world, rank=rank)
    st = R._ffi.Stats()
    hip.render_device(fx.target(0, 0).data_ptr(), pc, stream=stream, stats=st)
    my_rays = torch.tensor([st.rays, st.prim_tests], dtype=torch.float64, device="cpu" if rehearse else dev)
    if world > 1:
        dist.all_reduce(my_rays)
    total_rays, _ = [int(x) for x in my_rays.tolist()]
    accel_used = "bvh" if st.accel_used == R._ffi.RT_ACCEL_BVH else "brute"

    def run(n_steps):
        """n_steps frames: render, exchange, assemble.  Returns how many complete frames this rank ends up holding."""
        pending = [None, None]
        held = 0
        f, slot = 0, 0
        while f < n_steps:
            if pending[slot] is not None:  # the slot's previous group must have left before it is overwritten
                held += len(fx.finish(pending[slot], slot))
            n = min(fx.group, n_steps - f)
            for j in range(n):
                hip.render_device(fx.target(slot, j).data_ptr(), p, stream=stream)
            pending[slot] = fx.exchange_async(slot, n)
            f += n
            slot ^= 1
        for k in (slot, slot ^ 1):
            if pending[k] is not None:
                held += len(fx.finish(pending[k], k))
        return held

    run(a.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.timer_start(stream)
    held = run(a.steps)
    hip.timer_stop(stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = hip.timer_ms()
    frames_held = torch.tensor([held], dtype=torch.float64, device="cpu" if rehearse else dev)
    if world > 1:
        dist.all_reduce(frames_held)
    assert int(frames_held.item()) == a.steps, "every timed frame must have been put together on some rank"
    if world > 1:
        # the kernel alone (no gather between launches), for the roofline line
        hip.timer_start(stream)
        for _ in range(50):
            hip.render_device(fx.target(0, 0).data_ptr(), p, stream=stream)
        hip.timer_stop(stream)
        ev_ms = hip.timer_ms() / 50 * a.steps
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        # per launch of THIS rank's kernel, from the counting build of the same kernel on the same rows
        flops_per_launch = algorithmic_flops(st)
        exhaustive_flops = st.rays * (scene.n_tris * FLOP_PER_TRI_TEST + scene.n_spheres * FLOP_PER_SPHERE_TEST)
        kernel_ms = ev_ms / a.steps  # HIP events on the launch stream around the K launches
        # Algorithmic FLOPs of a launch.  Exhaustive kernel: rays x primitives x the full-path figure (what the
        # reference's loop over every object costs); BVH kernel: the tests the walk actually made (counted).
        if accel_used == "brute":
            algorithmic = exhaustive_flops
        else:
            algorithmic = flops_per_launch
        achieved = algorithmic / (kernel_ms * 1e-3) / 1e12
        executed = flops_per_launch / (kernel_ms * 1e-3) / 1e12
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
                                      + (f"; frame f is put together on rank f % {world}, one RCCL all-to-all per "
                                         f"{fx.group} frames" if world > 1 else "")},
            "roofline": {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
                         "traffic": pmc_traffic_bytes(), "kernel": "rt_trace_kernel", "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_flops_per_launch": algorithmic,
                         "executed_tflops": round(executed, 3), "executed_flops_per_launch": flops_per_launch,
                         "fp64_no_fma_ceiling_tflops": PEAK_FP64_NO_FMA_TFLOPS,
                         "frac_of_fp64_no_fma_ceiling": round(achieved / PEAK_FP64_NO_FMA_TFLOPS, 4),
                         "counters": {"tri_tests": st.prim_tests - st.sphere_tests, "sphere_tests": st.sphere_tests,
                                      "box_tests": st.node_tests, "primary": st.rays_primary, "shaded": st.rays_shadow},
                         "note": "bound: vector ALU (the scene sits in LDS/scalar cache, 3 B per pixel are written). "
                                 "achieved = algorithmic FLOPs / kernel time: rays x 12 triangles x 45 for this exhaustive "
                                 "kernel; executed = 30/primary ray + 80/shaded hit + 45/triangle test + 18/box test the "
                                 "kernel really ran (a bounds test lets most rays skip the triangle loop). The arithmetic "
                                 "is FP64 mul/add without FMA (bit parity with the reference), whose ceiling is 1/4 of the "
                                 "FP32 vector peak priced here. traffic = HBM bytes per launch from separate rocprofv3 "
                                 "--pmc passes over the same launch (profiles/), FETCH_SIZE x2 + WRITE_SIZE"},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(total_rays)
        if world == 1 and not a.no_side:
            line["other_scenes_1080p"] = side_measurements(hip, dev)
            line["general_kernel_scenes"] = general_measurements(hip, dev)
            line["frames_pipelined_over_streams"] = pipelined_measurements(hip, dev)
            line["ppm_text_on_device_random_image"] = ppm_measurements(hip, dev)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    hip.shutdown()


if __name__ == "__main__":
    main()

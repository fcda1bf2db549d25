#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the trace loop on the reference's teapot.h scene at 1920x1080.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one render() of the frame: every rank traces its round-robin row
bands on its own MI355X and (N > 1) the 8-bit rows are exchanged with RCCL so
that frame f is complete on rank f % N (one all-to-all per group of frames).
The scene is uploaded before the timed region; nothing crosses PCIe inside it.
Rank 0 prints one JSON line.

  value        total rays (primary + shadow + secondary: the reference's
               intersect_world call count) of all ranks / wall time, Mrays/s
  value_cold   the same with only the W warm-up steps in front (no clock ramp)
  roofline     the trace kernel against the FP32 vector peak.  achieved = the
               FLOPs the TIMED kernel executed / kernel time: the counters of a
               counting build that skips what the production kernel skips
               (RT_RENDER_COUNT_AS_RUN: background tiles are filled, not
               traced; camera rays of waves that look past the mesh are not set
               up), priced per test.  The exhaustive figure rays x primitives
               x 45 (what the reference's loop over every object costs) is
               reported beside it as algorithmic_equivalent; it is not what the
               hardware did.
  cpu_baseline the reference itself (oracle/_ref, gcc -O2, one core) on the
               same workload, timed on this host
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import raytracer_amd as R  # noqa: E402
from raytracer_amd.dist import BandGather, FrameExchange, default_band_rows  # noqa: E402

WIDTH, HEIGHT = 1920, 1080
SCENE = "cube"  # render_some_triangles + the teapot.h mesh (12 triangles), RayTracer.c:6727
# FLOPs per unit of work.  Double precision, the reference's statements (no FMA: a multiply and an add are two):
FLOP_TRI_EXACT = 45         # full Moeller-Trumbore path, RayTracer.c:943-966
FLOP_SPHERE_EXACT = 63      # two affine mat-vec + quadratic, RayTracer.c:898-940
FLOP_PER_PRIMARY_RAY = 30   # ray_for_pixel, RayTracer.c:1535
FLOP_PER_SHADED_HIT = 80    # prepare_computations + lighting (+ 1 pow), RayTracer.c:1360, 1273
# Single precision, the conservative first tests (an FMA counts two):
FLOP_TRI_FILTER = 34        # three dot-product chains of 3 + 6 + 6 terms, two subtractions, max3/min3
FLOP_SPHERE_FILTER = 54     # o', d' (two mat-vec), cross product, two squared norms; 36 when o' is the camera's
FLOP_SPHERE_FILTER_PRIMARY = 36
FLOP_PER_BOX_TEST = 18      # slab test (BVH node or mesh bounds)
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X vector FP32 peak (256 CU x 128 lanes x 2 x 2.4 GHz)
PEAK_FP64_NO_FMA_TFLOPS = 39.3  # FP64 vector: 78.6 with FMA, half of it for a mul/add-only stream
PMC_SUMMARY = os.path.join("profiles", "r03_pmc_cube_1080p.json")
LIGHTHOUSE_BIN = os.path.join(ROOT, "raytracer_amd", "data", "lighthouse_tris_f32.bin")
CLI = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer_hip")


def git_commit():
    """The commit of this tree: from git where the tree is a checkout, else from the note build() leaves beside the
    libraries (the tree that runs on the GPU box is a copy without .git)."""
    try:
        out = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10)
        if out.returncode == 0 and out.stdout.strip():
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True,
                                   timeout=10).stdout.strip()
            return out.stdout.strip() + ("+local changes" if dirty else "")
    except (OSError, subprocess.SubprocessError):
        pass
    try:
        return open(os.path.join(ROOT, "raytracer_amd", "lib", "BUILD_COMMIT")).read().strip() or None
    except OSError:
        return None


def pmc_traffic():
    """HBM bytes per launch of the headline kernel and the share of cycles its SIMDs' issue ports were busy with
    vector instructions, from the committed summary of separate rocprofv3 --pmc passes (a bench run cannot collect
    hardware counters itself), with the file and the commit it was taken at."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))
        return (int(d["derived_hbm_read_bytes_corrected_x2"] + d["derived_hbm_write_bytes"]),
                {"file": PMC_SUMMARY, "commit": d.get("_commit")}, d.get("derived_valu_issue_busy"))
    except (OSError, KeyError, ValueError):
        return None, None, None


def counted_as_run(hip, buf, **kw):
    """rt_stats of the counting build told to skip what the production kernel skips (same image, same tests as the
    timed kernel): what the roofline numerator is built from."""
    st = R._ffi.Stats()
    hip.render_device(buf.data_ptr(), R.make_params(count_rays=True, count_as_run=True, **kw), stats=st)
    return st


def executed_flops(st, n_spheres, brute):
    """(single-precision FLOPs, double-precision FLOPs) of the work one launch really did, from the counters of the
    counting build run as the production kernel runs (counted_as_run: rays_primary = camera rays set up).
    Exhaustive kernels run the single-precision test on every triangle and confirm the
    survivors in double precision; the BVH walk tests a leaf's triangles in double precision directly, and its
    camera rays run the single-precision test down their tile's list (of those, the tests that were not followed by
    a double-precision one are counted: a lower bound)."""
    tri_tests = st.prim_tests - st.sphere_tests
    sph_primary = min(st.sphere_tests, st.rays_primary * n_spheres)
    f32 = ((tri_tests if brute else max(0, tri_tests - st.tri_exact)) * FLOP_TRI_FILTER + sph_primary * FLOP_SPHERE_FILTER_PRIMARY
           + (st.sphere_tests - sph_primary) * FLOP_SPHERE_FILTER + st.node_tests * FLOP_PER_BOX_TEST)
    f64 = (st.tri_exact * FLOP_TRI_EXACT + st.sphere_exact * FLOP_SPHERE_EXACT + st.rays_primary * FLOP_PER_PRIMARY_RAY
           + st.rays_shadow * FLOP_PER_SHADED_HIT)
    return f32, f64


def roofline_of(st, n_spheres, n_tris, ms, run=None):
    """The roofline object of one kernel launch that took `ms`.  st: the counters of the reference's loop (every pixel
    traced: ray counts, algorithmic_equivalent); run: the counters of what the timed kernel executes (counted_as_run);
    without it st stands for both."""
    brute = st.accel_used == R._ffi.RT_ACCEL_BRUTE
    run = run or st
    f32, f64 = executed_flops(run, n_spheres, brute)
    total = f32 + f64
    achieved = total / (ms * 1e-3) / 1e12
    # what this mix of instructions could reach at most: each part at its own peak
    ceiling = total / (f32 / PEAK_FP32_VALU_TFLOPS + f64 / PEAK_FP64_NO_FMA_TFLOPS) if total else 0.0
    assert achieved <= ceiling * 1.0001, (achieved, ceiling)
    exhaustive = st.rays * (n_tris * FLOP_TRI_EXACT + n_spheres * FLOP_SPHERE_EXACT)
    return {"bound": "valu", "achieved": round(achieved, 3), "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_FP32_VALU_TFLOPS, 4), "kernel_ms": round(ms, 5),
            "executed_flops_per_launch": {"fp32_fma": int(f32), "fp64_mul_add": int(f64)},
            "ceiling_of_issued_mix_tflops": round(ceiling, 2), "frac_of_ceiling": round(achieved / ceiling, 4) if ceiling else None,
            "algorithmic_equivalent": {"flops_per_launch": int(exhaustive), "tflops": round(exhaustive / (ms * 1e-3) / 1e12, 3),
                                       "what": "rays x primitives x the full-path figure: the reference's loop over every "
                                               "object; not executed here (bounds, single-precision first tests, BVH)"},
            "counters": {"tri_tests": run.prim_tests - run.sphere_tests, "tri_exact": run.tri_exact, "sphere_tests": run.sphere_tests,
                         "sphere_exact": run.sphere_exact, "box_tests": run.node_tests, "camera_rays_set_up": run.rays_primary,
                         "shaded": run.rays_shadow, "secondary": run.rays_secondary,
                         "what": "executed by the timed kernel (background tiles filled, not traced)"}}


def time_launches(hip, buf, p, n, reps=3):
    """Best mean kernel time (ms) of n back-to-back launches on the library's stream, HIP events around the batch."""
    for _ in range(max(2, n // 5)):
        hip.render_device(buf.data_ptr(), p)
    best = None
    for _ in range(reps):
        hip.timer_start()
        for _ in range(n):
            hip.render_device(buf.data_ptr(), p)
        hip.timer_stop()
        ms = hip.timer_ms() / n
        best = ms if best is None else min(best, ms)
    return best


def side_measurements(hip, dev):
    """Context for the headline: the other benchmark scenes at 1920x1080 on this GPU, each with its own roofline."""
    out = {}
    buf = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.uint8, device=dev)
    for name, scene, tr, accel in (("spheres_1080p", "spheres", 0.0, R._ffi.RT_ACCEL_AUTO),
                                   ("lighthouse_1080p_bvh", "lighthouse", 0.0, R._ffi.RT_ACCEL_BVH),
                                   ("lighthouse_1080p_exhaustive", "lighthouse", 0.0, R._ffi.RT_ACCEL_BRUTE),
                                   ("lighthouse_1080p_reflective_bvh", "lighthouse", 0.3, R._ffi.RT_ACCEL_BVH)):
        hip.upload(R.stock_scene(scene, WIDTH, HEIGHT, tr))
        st = R._ffi.Stats()
        hip.render_device(buf.data_ptr(), R.make_params(accel=accel, count_rays=True), stats=st)
        run = counted_as_run(hip, buf, accel=accel)
        ms = time_launches(hip, buf, R.make_params(accel=accel), 100 if accel != R._ffi.RT_ACCEL_BRUTE else 10)
        sc = hip.scene_counts()
        rf = roofline_of(st, sc[0], sc[1], ms, run)
        out[name] = {"rays": st.rays, "kernel_ms": round(ms, 5), "mrays_s": round(st.rays / ms / 1e3, 1), "roofline": rf}
    return out


def full_size_configs(hip, dev, only=None):
    """The two multi-GPU configurations at full size on this one GPU: C4 (4K lighthouse, mesh reflective 0.3, BVH) and
    C5 (8K, 16 spp, cube + spheres, reflective)."""
    out = {}
    for name, scene, w, h, tr, sr, spp, accel, n in (
            ("C4_lighthouse_3840x2160_reflective_bvh", "lighthouse", 3840, 2160, 0.3, 0.0, 1, R._ffi.RT_ACCEL_BVH, 20),
            ("C5_cube_spheres_7680x4320_16spp_reflective", "cube_spheres", 7680, 4320, 0.3, 0.3, 4, R._ffi.RT_ACCEL_AUTO, 3)):
        if only and not name.startswith(only):
            continue
        hip.upload(R.stock_scene(scene, w, h, tr, sr))
        buf = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        st = R._ffi.Stats()
        hip.render_device(buf.data_ptr(), R.make_params(spp_grid=spp, accel=accel, count_rays=True), stats=st)
        run = counted_as_run(hip, buf, spp_grid=spp, accel=accel)
        ms = time_launches(hip, buf, R.make_params(spp_grid=spp, accel=accel), n, reps=2)
        sc = hip.scene_counts()
        out[name] = {"rays": st.rays, "kernel_ms": round(ms, 4), "mrays_s": round(st.rays / ms / 1e3, 1),
                     "roofline": roofline_of(st, sc[0], sc[1], ms, run)}
        del buf
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
        ms = time_launches(hip, buf, R.make_params(), steps, reps=1)
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


def ppm_to_host_measurements(hip, frames=60):
    """render() + write_canvas_to_file()'s text, into host memory, per 1080p frame of the headline scene: one
    synchronous call per frame (rt_hip_render_ppm, pageable buffer) against two frames in flight
    (rt_hip_render_ppm_submit / _collect: pinned buffers, the copy of frame f beside the rendering of f + 1)."""
    import numpy as np

    hip.upload(R.stock_scene(SCENE, WIDTH, HEIGHT))
    text, _ = hip.render_ppm()
    t0 = time.perf_counter()
    for _ in range(10):
        hip.render_ppm()
    sync_ms = (time.perf_counter() - t0) / 10 * 1e3
    p = R.make_params()
    hip.render_ppm_submit(0, p)
    for f in range(1, 4):
        hip.render_ppm_submit(f % 2, p)
        hip.render_ppm_collect((f - 1) % 2, copy=False)
    t0 = time.perf_counter()
    for f in range(4, 4 + frames):
        hip.render_ppm_submit(f % 2, p)
        hip.render_ppm_collect((f - 1) % 2, copy=False)
    piped_ms = (time.perf_counter() - t0) / frames * 1e3
    last, _ = hip.render_ppm_collect((4 + frames - 1) % 2)
    return {"frame": f"{SCENE} {WIDTH}x{HEIGHT}", "text_bytes": len(text), "synchronous_ms_per_frame": round(sync_ms, 4),
            "two_in_flight_ms_per_frame": round(piped_ms, 4), "host_link_gb_s": round(len(text) / piped_ms / 1e6, 1),
            "bytes_identical": last == text}


def run_json(cmd, timeout, cwd=None):
    """Run a child process whose last stdout line starting with '{' is JSON; (dict or None, wall seconds, error text)."""
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=cwd)
    except (OSError, subprocess.SubprocessError) as e:
        return None, time.perf_counter() - t0, repr(e)
    dt = time.perf_counter() - t0
    if out.returncode != 0:
        return None, dt, (out.stderr or out.stdout)[-300:]
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line), dt, None
    return {}, dt, None


def first_call_measurements():
    """What a drop-in user gets: one process, one render, one canvas.ppm (the reference's main(), RayTracer.c:7236-7251).
    (a) the headline frame through the C host in a fresh process: time of rt_hip_init, of the upload and of the first
    rt_hip_render (production kernel), then the steady state.  (b) configuration 1 (lighthouse, 300x300) end to end,
    process start to canvas.ppm written, next to the reference's own code on one host core (same mesh, same file)."""
    out = {}
    if not os.path.exists(CLI):
        return {"error": "raytracer_amd/bin/raytracer_hip is missing"}
    with tempfile.TemporaryDirectory() as tmp:
        info, wall, err = run_json([CLI, SCENE, "--size", str(WIDTH), str(HEIGHT), "--host-ppm", "--no-count", "--frames", "50",
                                    "--json", "--out", os.path.join(tmp, "a.ppm")], 120)
        out["headline_frame_fresh_process"] = info if info else {"error": err}
        if info:
            out["first_call_ms"] = info.get("first_call_ms")
        # the 1080p lighthouse (a frame that walks a BVH: the one whose tile plan can be tuned by measurement), one frame
        # per process as the reference would render it (RT_RENDER_NO_TUNE) and with the plan tuned first
        lh = {}
        for key, extra in (("no_tune", []), ("tuned_first", ["--tune"])):
            info, wall, err = run_json([CLI, "lighthouse", "--tris-f32", LIGHTHOUSE_BIN, "--size", str(WIDTH), str(HEIGHT), "--host-ppm",
                                        "--no-count", "--json", "--out", os.path.join(tmp, "l.ppm")] + extra, 120)
            lh[key] = ({k: info.get(k) for k in ("init_ms", "upload_ms", "tune_ms", "first_call_ms", "first_call_kernel_ms")}
                       if info else {"error": err})
            if info:
                lh[key]["process_wall_s"] = round(wall, 4)
        out["lighthouse_1080p_first_frame"] = lh
        hip_ppm, ref_ppm = os.path.join(tmp, "hip.ppm"), os.path.join(tmp, "ref.ppm")
        info, wall, err = run_json([CLI, "lighthouse", "--tris-f32", LIGHTHOUSE_BIN, "--json", "--out", hip_ppm], 120)
        c1 = {"hip_process_wall_s": round(wall, 4), "hip": info if info else {"error": err}}
        ref = os.path.join(ROOT, "oracle", "_ref", "rt_ref_harness_timing")
        if os.path.exists(ref):
            rinfo, rwall, rerr = run_json([ref, "--scene", "lighthouse", "--tris-f32", LIGHTHOUSE_BIN, "--out-ppm", ref_ppm], 300)
            c1["reference_process_wall_s"] = round(rwall, 4)
            if rinfo:
                c1["reference_pixel_loop_s"] = rinfo.get("seconds")
                c1["speedup_process_wall"] = round(rwall / wall, 2)
                try:
                    sha = [hashlib.sha256(open(f, "rb").read()).hexdigest() for f in (hip_ppm, ref_ppm)]
                    c1["canvas_ppm_identical"] = sha[0] == sha[1]
                    c1["canvas_ppm_sha256"] = sha[0]
                except OSError as e:
                    c1["canvas_ppm_identical"] = repr(e)
            else:
                c1["reference_error"] = rerr
        out["C1_lighthouse_300x300_one_shot"] = c1
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


CLOCK_RAMP_LAUNCHES = 1500  # untimed launches of the frame at the end of the set-up (see main)


def run_frames(render, fx, n_steps):
    """n_steps frames through a FrameExchange: render(target) fills this rank's bands of a frame, groups of frames
    are exchanged while the next group renders.  Returns how many complete frames this rank ends up holding."""
    pending = [None, None]
    held = 0
    f, slot = 0, 0
    while f < n_steps:
        if pending[slot] is not None:  # the slot's previous group must have left before it is overwritten
            held += len(fx.finish(pending[slot], slot))
        n = min(fx.group, n_steps - f)
        for j in range(n):
            render(fx.target(slot, j))
        pending[slot] = fx.exchange_async(slot, n)
        f += n
        slot ^= 1
    for k in (slot, slot ^ 1):
        if pending[k] is not None:
            held += len(fx.finish(pending[k], k))
    return held


def gather_to_root(hip, dev, world, rank, scene_name, w, h, tr, sr, spp, accel, steps, rehearse):
    """One frame at a time, the way a single render() call works over several GPUs: every rank renders its bands,
    ONE gather brings them to rank 0, rank 0 puts the rows in place.  Wall time per frame, max over ranks."""
    hip.upload(R.stock_scene(scene_name, w, h, tr, sr))
    band_rows = default_band_rows(h, world)
    bg = BandGather(h, w, band_rows, world, rank, dev)
    p = R.make_params(spp_grid=spp, accel=accel, band_rows=band_rows, n_ranks=world, rank=rank)
    st = R._ffi.Stats()
    hip.render_device(bg.local.data_ptr(), R.make_params(spp_grid=spp, accel=accel, count_rays=True, band_rows=band_rows,
                                                         n_ranks=world, rank=rank), stats=st)
    stream = torch.cuda.current_stream().cuda_stream
    t_stats = torch.tensor([float(st.rays)], dtype=torch.float64, device="cpu" if rehearse else dev)
    dist.all_reduce(t_stats)
    rays = int(t_stats.item())

    def frame():
        hip.render_device(bg.local.data_ptr(), p, stream=stream)
        return bg.gather()

    for _ in range(2):
        frame()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if rehearse else dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item()) / steps * 1e3
    kernel_ms = time_launches(hip, bg.local, p, max(3, steps // 2), reps=1)
    del bg
    return {"rays": rays, "ms_per_frame": round(ms, 4), "mrays_s": round(rays / ms / 1e3, 1), "rank0_kernel_ms": round(kernel_ms, 4),
            "band_rows": band_rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the context measurements beside the headline")
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
    # the same rows counted as the timed kernel runs them (the roofline's numerator)
    run = R._ffi.Stats()
    hip.render_device(fx.target(0, 0).data_ptr(), R.make_params(max_depth=5, count_rays=True, count_as_run=True, band_rows=band_rows,
                                                              n_ranks=world, rank=rank), stream=stream, stats=run)

    def render(target):
        hip.render_device(target.data_ptr(), p, stream=stream)

    def timed_region():
        """W warm-up steps, then K timed steps between barrier + synchronize on both sides; (wall s, event ms)."""
        run_frames(render, fx, a.warmup)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hip.timer_start(stream)
        held = run_frames(render, fx, a.steps)
        hip.timer_stop(stream)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        frames_held = torch.tensor([held], dtype=torch.float64, device="cpu" if rehearse else dev)
        if world > 1:
            dist.all_reduce(frames_held)
        assert int(frames_held.item()) == a.steps, "every timed frame must have been put together on some rank"
        t = torch.tensor([wall], dtype=torch.float64, device="cpu" if rehearse else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), hip.timer_ms()

    # First as the protocol alone has it: W warm-up steps after the upload, K timed steps (value_cold).
    dt_cold, _ = timed_region()
    # The device's clocks come up over the first thousand-odd launches after any pause (the upload is one): with that
    # ramp counted to the set-up, like the upload, the same region again (value).
    for _ in range(CLOCK_RAMP_LAUNCHES):
        hip.render_device(fx.target(0, 0).data_ptr(), p, stream=stream)
    dt, ev_ms = timed_region()
    if world > 1:
        # the kernel alone (no exchange between launches), for the roofline line
        hip.timer_start(stream)
        for _ in range(50):
            hip.render_device(fx.target(0, 0).data_ptr(), p, stream=stream)
        hip.timer_stop(stream)
        ev_ms = hip.timer_ms() / 50 * a.steps
    line = None
    if rank == 0:
        kernel_ms = ev_ms / a.steps  # HIP events on the launch stream around the K launches
        rf = roofline_of(st, scene.n_spheres, scene.n_tris, kernel_ms, run)
        traffic, source, issue_busy = pmc_traffic()
        rf["traffic"] = traffic
        rf["traffic_source"] = source
        rf["valu_issue_busy"] = round(issue_busy, 4) if issue_busy is not None else None
        rf["kernel"] = "rt_trace_kernel"
        rf["note"] = ("bound: vector ALU (the scene sits in LDS/scalar cache, 3 B per pixel are written). achieved = FLOPs the "
                      "TIMED kernel executed / kernel time: counters of the counting build run as the production kernel runs "
                      "(background tiles filled, not traced; only the camera rays that are set up), priced per test; peak = "
                      "the FP32 vector peak the metric names: 157.3 TFLOP/s = a fused multiply-add in 32 lanes of each of "
                      "1024 SIMDs per cycle (a 64-lane instruction holds its SIMD's issue port for 2 cycles). FP64 runs at "
                      "half that rate, and this build uses no FP64 FMA (bit parity with a reference binary that has "
                      "none), so the executed mix is single-precision FMA (filter and box tests) and double-precision "
                      "mul/add (confirms, ray set-up, shading); ceiling_of_issued_mix prices each part at its own peak "
                      "(157.3 and 39.3 TFLOP/s). This 13 us frame is one launch at the latency of a single wave's "
                      "dependent chain (a rank's share of it takes as long as the whole); the kernels that fill the chip "
                      "are under full_size_configs_one_gpu and other_scenes_1080p. valu_issue_busy = the cycles the "
                      "executed vector instructions hold their SIMD's issue port (2 per instruction, 4 for an FP64 "
                      "add/mul/fma, 8 or 16 for a transcendental; by instruction class from SQ_INSTS_VALU_*) / (1024 "
                      "SIMDs x kernel cycles), and traffic = HBM bytes per launch (FETCH_SIZE "
                      "x2 + WRITE_SIZE), both from separate rocprofv3 --pmc passes")
        line = {
            "metric": "Mrays/s (primary+shadow) at 1920x1080 teapot scene; % of FP32 VALU peak",
            "value": round(total_rays * a.steps / dt / 1e6, 2),
            "value_cold": round(total_rays * a.steps / dt_cold / 1e6, 2),
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
                       "rays_counted": "intersect_world calls: primary + shadow (+ reflected: none in this scene)",
                       "parallelism": f"row bands of {band_rows} rows round-robin over {world} rank(s)"
                                      + (f"; frame f is put together on rank f % {world}, one RCCL all-to-all per "
                                         f"{fx.group} frames" if world > 1 else ""),
                       "commit": git_commit(),
                       "setup_launches": CLOCK_RAMP_LAUNCHES,
                       "note": "after any pause of the device (the scene upload is one) the first thousands of launches run "
                               "up to 12 % slower while the clocks come up (14.7 us per frame for launches 1-26, 14.1 for "
                               "100-300, 13.0 from 1000 on). value_cold is timed as the protocol alone has it (W warm-up "
                               "steps after the upload, then K steps); for value the set-up ends with setup_launches "
                               "untimed launches of the frame before the same W + K steps"},
            "roofline": rf,
        }
    del fx
    if world == 1 and rank == 0:
        if not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(total_rays)
        if not a.no_side:
            for key, fn in (("other_scenes_1080p", lambda: side_measurements(hip, dev)),
                            ("full_size_configs_one_gpu", lambda: full_size_configs(hip, dev)),
                            ("general_kernel_scenes", lambda: general_measurements(hip, dev)),
                            ("frames_pipelined_over_streams", lambda: pipelined_measurements(hip, dev)),
                            ("ppm_text_on_device_random_image", lambda: ppm_measurements(hip, dev)),
                            ("ppm_text_to_host_memory", lambda: ppm_to_host_measurements(hip))):
                try:
                    line[key] = fn()
                except Exception as e:  # a context measurement must not take the headline with it
                    line[key] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 and not a.no_side:
        # What a single render() call over N GPUs costs (one gather to rank 0 per frame), for the headline frame and
        # for the two configurations the row-band design is for.  Every rank takes part; rank 0 reports.
        multi = {}
        for name, args in (("headline_1080p", (SCENE, WIDTH, HEIGHT, 0.0, 0.0, 1, R._ffi.RT_ACCEL_AUTO, 200)),
                           ("C4_lighthouse_3840x2160_reflective_bvh", ("lighthouse", 3840, 2160, 0.3, 0.0, 1, R._ffi.RT_ACCEL_BVH, 50)),
                           ("C5_cube_spheres_7680x4320_16spp_reflective", ("cube_spheres", 7680, 4320, 0.3, 0.3, 4, R._ffi.RT_ACCEL_AUTO, 5))):
            try:
                multi[name] = gather_to_root(hip, dev, world, rank, *args, rehearse)
            except Exception as e:  # noqa: BLE001 - a context measurement must not take the headline with it
                multi[name] = {"error": f"{type(e).__name__}: {e}"}
        # C5 on ONE of these GPUs (rank 0 alone, the whole frame), so that the N-GPU figure above has its own 1-GPU
        # figure from the same node and run beside it
        one_gpu_ms = None
        if rank == 0:
            try:
                one = full_size_configs(hip, dev, only="C5")
                one_gpu_ms = next(iter(one.values()))["kernel_ms"]
            except Exception as e:  # noqa: BLE001 - context only
                multi["C5_one_gpu_error"] = f"{type(e).__name__}: {e}"
        dist.barrier()
        if rank == 0:
            line["single_frame_gather_to_rank0"] = multi
            c5 = multi.get("C5_cube_spheres_7680x4320_16spp_reflective")
            if one_gpu_ms and c5 and "ms_per_frame" in c5:
                line["scaling_c5"] = {"one_gpu_kernel_ms": one_gpu_ms, "n_gpus": world, "n_gpu_ms_per_frame": c5["ms_per_frame"],
                                      "speedup": round(one_gpu_ms / c5["ms_per_frame"], 3),
                                      "what": "8K x 16 spp cube + spheres, reflective: one frame rendered on all ranks and "
                                              "gathered to rank 0 (wall time per frame, exchange included) against the "
                                              "same frame's kernel time on rank 0's GPU alone"}
    hip.shutdown()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not a.no_side and not rehearse:
            if world == 1:
                line["drop_in_first_call"] = first_call_measurements()
            elif os.path.exists(CLI):
                # the C host's own multi-GPU path (one process, N devices, RCCL gather): a child process of rank 0
                # after the ranks are done, so that a failure there cannot take the line above with it
                info, wall, err = run_json([CLI, SCENE, "--size", str(WIDTH), str(HEIGHT), "--gpus", str(world), "--no-count",
                                            "--frames", "100", "--json", "--out", os.path.join(tempfile.gettempdir(), "rt_multi.ppm")], 180)
                line["c_host_render_multi"] = info if info else {"error": err, "wall_s": round(wall, 2)}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

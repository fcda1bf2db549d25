#!/usr/bin/env python3
"""Regenerates tests/golden/* and raytracer_amd/data/lighthouse_tris_f32.bin.

Runs ONLY where the reference is present: it drives oracle/_ref/rt_ref_harness
(the unmodified reference compiled by oracle/Makefile, `make -C oracle ref`)
and stores what it produced - canvases, images, ray counts, scene dumps - as
small fixtures.  Nothing of the reference's source text is stored.

    python tests/golden/make_golden.py [--big]     (--big adds the 1920x1080 images)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refdump  # noqa: E402

HARNESS = os.path.join(ROOT, "oracle", "_ref", "rt_ref_harness")
# the reference's OWN scene functions feeding the harness loop (render() interposed): planes, patterns, glass, ...
CAPTURE = os.path.join(ROOT, "oracle", "_ref", "rt_ref_capture")
OBJ = "/root/reference/RayTracer/lighthouse.obj"

# name -> harness arguments.  f64 canvases are kept for the small cases only.
CASES = {
    # the reference's three in-scope stock scenes at its built-in 300x300 (config 1 and its siblings)
    "lighthouse_300": dict(scene="lighthouse", size=(300, 300), rgb=True, f64=False),
    "cube_300": dict(scene="cube", size=(300, 300), rgb=True, f64=False),
    "spheres_300": dict(scene="spheres", size=(300, 300), rgb=True, f64=False),
    # small double-precision canvases for bit-level comparison
    "lighthouse_96x64": dict(scene="lighthouse", size=(96, 64), f64=True, dump=True),
    "cube_96x64": dict(scene="cube", size=(96, 64), f64=True, dump=True),
    "spheres_96x64": dict(scene="spheres", size=(96, 64), f64=True, dump=True),
    "cube_spheres_96x64": dict(scene="cube_spheres", size=(96, 64), f64=True, dump=True),
    # reflections (config 4 shape) and supersampling (config 5 shape), small
    "lighthouse_refl_96x64": dict(scene="lighthouse", size=(96, 64), tri_refl=0.3, f64=True),
    "cube_spheres_refl_96x64": dict(scene="cube_spheres", size=(96, 64), tri_refl=0.3, sph_refl=0.3, f64=True),
    "cube_spheres_ss4_48x32": dict(scene="cube_spheres", size=(48, 32), spp=4, tri_refl=0.3, sph_refl=0.3, f64=True),
    "spheres_ss2_48x32": dict(scene="spheres", size=(48, 32), spp=2, f64=True),
    # non-square, odd sizes (ragged tiles)
    "cube_101x37": dict(scene="cube", size=(101, 37), f64=True),
    "spheres_33x71": dict(scene="spheres", size=(33, 71), f64=True),
    "lighthouse_refl_320x180": dict(scene="lighthouse", size=(320, 180), tri_refl=0.3, rgb=True, f64=False),
    # the scenes beyond spheres and triangles, built by the reference's own scene functions
    "refraction_300": dict(scene="refraction", size=(300, 300), rgb=True, f64=False, capture=True),
    "world_with_plane_300": dict(scene="world_with_plane", size=(300, 300), rgb=True, f64=False, capture=True),
    "dual_spheres_300": dict(scene="dual_spheres", size=(300, 300), rgb=True, f64=False, capture=True),
    "refraction_96x64": dict(scene="refraction", size=(96, 64), f64=True, dump=True, capture=True),
    "world_with_plane_96x64": dict(scene="world_with_plane", size=(96, 64), f64=True, dump=True, capture=True),
    "dual_spheres_96x64": dict(scene="dual_spheres", size=(96, 64), f64=True, dump=True, capture=True),
    "refraction_ss2_48x32": dict(scene="refraction", size=(48, 32), spp=2, f64=True, capture=True),
    "world_with_plane_77x45": dict(scene="world_with_plane", size=(77, 45), f64=True, capture=True),
}
BIG = {
    "refraction_1080p": dict(scene="refraction", size=(1920, 1080), rgb=True, f64=False, capture=True),
    "world_with_plane_1080p": dict(scene="world_with_plane", size=(1920, 1080), rgb=True, f64=False, capture=True),
    "dual_spheres_1080p": dict(scene="dual_spheres", size=(1920, 1080), rgb=True, f64=False, capture=True),
    "lighthouse_1080p": dict(scene="lighthouse", size=(1920, 1080), rgb=True, f64=False),
    "cube_1080p": dict(scene="cube", size=(1920, 1080), rgb=True, f64=False),
    "spheres_1080p": dict(scene="spheres", size=(1920, 1080), rgb=True, f64=False),
}


def run_case(name, c, tmp):
    w, h = c["size"]
    args = [HARNESS, "--scene", c["scene"], "--size", str(w), str(h), "--obj", OBJ]
    if c.get("capture"):
        args = [CAPTURE, "--capture", "--scene", c["scene"], "--size", str(w), str(h)]
    if c.get("spp", 1) != 1:
        args += ["--spp-grid", str(c["spp"])]
    if c.get("tri_refl"):
        args += ["--tri-reflective", repr(c["tri_refl"])]
    if c.get("sph_refl"):
        args += ["--sph-reflective", repr(c["sph_refl"])]
    f64 = os.path.join(tmp, name + ".f64")
    ppm = os.path.join(tmp, name + ".ppm")
    dump = os.path.join(tmp, name + ".scene")
    args += ["--out-f64", f64, "--out-ppm", ppm]
    if c.get("dump"):
        args += ["--dump-scene", dump]
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
    info = json.loads(out.strip().splitlines()[-1])
    canvas = np.fromfile(f64, dtype="<f8").reshape(h, w, 3)
    with open(ppm, "rb") as f:
        ppm_bytes = f.read()
    # color_convert, as the harness applied it for the ppm
    rgb = np.clip(np.trunc(canvas * 255.0), 0, 255).astype(np.uint8)
    entry = dict(
        scene=c["scene"], width=w, height=h, spp_grid=c.get("spp", 1),
        tri_reflective=c.get("tri_refl", 0.0), sph_reflective=c.get("sph_refl", 0.0),
        rays=info["rays"], max_list=info["max_list"], exact_zero_t=info["exact_zero_t"],
        ppm_sha256=hashlib.sha256(ppm_bytes).hexdigest(), ppm_bytes=len(ppm_bytes),
        hit_pixels=int((canvas.sum(axis=2) != 0).sum()),
        reference_seconds_here=info["seconds"],
    )
    if c.get("f64"):
        np.savez_compressed(os.path.join(HERE, name + ".canvas.npz"), canvas=canvas)
        entry["canvas"] = name + ".canvas.npz"
    if c.get("rgb"):
        np.savez_compressed(os.path.join(HERE, name + ".rgb.npz"), rgb=rgb)
        entry["rgb"] = name + ".rgb.npz"
    if c.get("dump"):
        head, shapes, ext = refdump.load(dump)
        refdump.save_npz(os.path.join(HERE, name + ".scene.npz"), head, shapes, ext)
        entry["scene_dump"] = name + ".scene.npz"
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--only", default=None, help="one case name, or a prefix ending in *")
    a = ap.parse_args()
    if not os.path.exists(HARNESS):
        sys.exit("build the reference first: make -C oracle ref")
    mpath = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(mpath)) if os.path.exists(mpath) else {}
    cases = dict(CASES)
    if a.big:
        cases.update(BIG)
    with tempfile.TemporaryDirectory() as tmp:
        for name, c in cases.items():
            if a.only and not (a.only == name or (a.only.endswith("*") and name.startswith(a.only[:-1]))):
                continue
            manifest[name] = run_case(name, c, tmp)
            print(name, manifest[name]["rays"], manifest[name]["ppm_sha256"][:16], flush=True)
        # the mesh as the reference's loader delivers it: float positions, file order
        dump = os.path.join(tmp, "lh.scene")
        subprocess.run([HARNESS, "--scene", "lighthouse", "--size", "8", "8", "--obj", OBJ, "--no-render",
                        "--dump-scene", dump], check=True, capture_output=True)
        head, shapes, _ = refdump.load(dump)
        tri = shapes[shapes["type"] == refdump.TRIANGLE]
        xyz9 = np.concatenate([tri["p1"][:, :3], tri["p2"][:, :3], tri["p3"][:, :3]], axis=1)
        f32 = xyz9.astype("<f4")
        assert (f32.astype(np.float64) == xyz9).all(), "loader positions are floats widened to double"
        os.makedirs(os.path.join(ROOT, "raytracer_amd", "data"), exist_ok=True)
        f32.tofile(os.path.join(ROOT, "raytracer_amd", "data", "lighthouse_tris_f32.bin"))
        print("lighthouse triangles:", f32.shape)
    with open(mpath, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()

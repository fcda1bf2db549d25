"""The literal drop-in: the reference's own scene functions and PPM writer (compiled unmodified into
oracle/_ref/ where the reference source is present), with render() answered by the HIP library through the
maintainer's binding raytracer_amd/host/rt_reference_binding.c.  canvas.ppm must hash to the reference's file."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "raytracer_dropin")

REFERENCE_DIR = "/root/reference/RayTracer"


@pytest.mark.skipif(not os.path.exists(EXE) or not os.path.exists(os.path.join(REFERENCE_DIR, "lighthouse.obj")),
                    reason="needs oracle/_ref/raytracer_dropin and the reference's lighthouse.obj (build container only)")
def test_reference_lighthouse_world_flattens_to_the_scene_the_gpu_tests_render(tmp_path):
    """Configuration 1 end to end without the OBJ travelling: the reference's OWN render_lighthouse_scene()
    (load_model_file + objpar.h, RayTracer.c:6772-6786, 1621-1705) builds its world, the maintainer's binding
    flattens it (no device involved), and the result is bit for bit the scene that stock_scene("lighthouse") makes
    from raytracer_amd/data/lighthouse_tris_f32.bin: the scene every GPU test and the C host render, whose image
    hashes to the reference's canvas.ppm."""
    import numpy as np

    import raytracer_amd as R

    dump = tmp_path / "scene.bin"
    out = subprocess.run([EXE, "--dump-scene", str(dump), "lighthouse"], cwd=REFERENCE_DIR, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    raw = dump.read_bytes()
    n_tris, n_spheres = np.frombuffer(raw, dtype=np.int32, count=2)
    head = np.frombuffer(raw, dtype=np.float64, count=3 + 16 + 6, offset=8)
    soa = np.frombuffer(raw, dtype=np.float64, count=12 * int(n_tris), offset=8 + 8 * 25).reshape(12, n_tris)
    s = R.stock_scene("lighthouse", 300, 300)
    assert (n_tris, n_spheres) == (s.n_tris, s.n_spheres) == (1336, 0)
    assert head[0] == s.camera.pixel_size and head[1] == s.camera.half_width and head[2] == s.camera.half_height
    assert np.array_equal(head[3:19], np.array(s.camera.inv_view[:]))
    assert np.array_equal(head[19:22], np.array(s.light.position[:])) and np.array_equal(head[22:25], np.array(s.light.intensity[:]))
    mine = np.ctypeslib.as_array(s.tri_soa, shape=(12, s.n_tris))
    assert np.array_equal(soa.view(np.uint64), mine.view(np.uint64))  # every bit, signed zeros included


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref not built (reference sources absent at build time)")
@pytest.mark.parametrize("scene,golden", [("cube", "cube_300"), ("spheres", "spheres_300"),
                                          ("refraction", "refraction_300"), ("world_with_plane", "world_with_plane_300"),
                                          ("dual_spheres", "dual_spheres_300")])
def test_reference_scene_functions_render_through_the_library(tmp_path, manifest, scene, golden):
    out = subprocess.run([EXE, scene], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ppm = (tmp_path / "canvas.ppm").read_bytes()
    assert hashlib.sha256(ppm).hexdigest() == manifest[golden]["ppm_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("scene,golden,extra", [
    ("cube", "cube_300", []),
    ("spheres", "spheres_300", []),
    ("refraction", "refraction_300", []),
    ("world_with_plane", "world_with_plane_300", []),
    ("dual_spheres", "dual_spheres_300", []),
    ("lighthouse", "lighthouse_300", ["--tris-f32", os.path.join(ROOT, "raytracer_amd", "data", "lighthouse_tris_f32.bin")]),
])
def test_c_host_program_writes_the_reference_ppm(tmp_path, manifest, scene, golden, extra):
    """raytracer_hip: the C host (scene builder + library + PPM writer), no Python in the loop."""
    exe = os.path.join(ROOT, "raytracer_amd", "bin", "raytracer_hip")
    out = subprocess.run([exe, scene, *extra], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    ppm = (tmp_path / "canvas.ppm").read_bytes()
    assert hashlib.sha256(ppm).hexdigest() == manifest[golden]["ppm_sha256"]


HARNESS = os.path.join(ROOT, "oracle", "_ref", "rt_ref_harness_timing")


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref not built (reference sources absent at build time)")
def test_reference_harness_renders_the_float_mesh_to_the_reference_file(tmp_path, manifest):
    """The CPU baseline of configuration 1 on a host without lighthouse.obj: the reference's own create_triangle /
    color_at over the 1336 triangles read as floats must write the bytes of the reference's canvas.ppm."""
    ppm = tmp_path / "ref.ppm"
    out = subprocess.run([HARNESS, "--scene", "lighthouse", "--tris-f32",
                          os.path.join(ROOT, "raytracer_amd", "data", "lighthouse_tris_f32.bin"), "--out-ppm", str(ppm)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert hashlib.sha256(ppm.read_bytes()).hexdigest() == manifest["lighthouse_300"]["ppm_sha256"]

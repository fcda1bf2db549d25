"""The literal drop-in: the reference's own scene functions and PPM writer (compiled unmodified into
oracle/_ref/ where the reference source is present), with render() answered by the HIP library through the
maintainer's binding raytracer_amd/host/rt_reference_binding.c.  canvas.ppm must hash to the reference's file."""
import hashlib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "raytracer_dropin")

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref not built (reference sources absent at build time)")
@pytest.mark.parametrize("scene,golden", [("cube", "cube_300"), ("spheres", "spheres_300"),
                                          ("refraction", "refraction_300"), ("world_with_plane", "world_with_plane_300"),
                                          ("dual_spheres", "dual_spheres_300")])
def test_reference_scene_functions_render_through_the_library(tmp_path, manifest, scene, golden):
    out = subprocess.run([EXE, scene], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ppm = (tmp_path / "canvas.ppm").read_bytes()
    assert hashlib.sha256(ppm).hexdigest() == manifest[golden]["ppm_sha256"]


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

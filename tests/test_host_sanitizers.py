"""The host C code (scene flattening, OBJ reader, SAH BVH) under AddressSanitizer and UBSan: every stock scene,
a 5000-triangle mesh, OBJ text with quads / comments / a malformed face, leaf sizes 0-7, finish() twice."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "host_sanitize"
    cmd = ["gcc", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-pthread", "-std=c11", "-D_DEFAULT_SOURCE", "-D_POSIX_C_SOURCE=200809L", "-ffp-contract=off",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "raytracer_amd", "host"),
           os.path.join(ROOT, "tests", "host_sanitize_driver.c"), os.path.join(ROOT, "raytracer_amd", "host", "rt_scene.c"),
           os.path.join(ROOT, "raytracer_amd", "host", "rt_bvh.c"), "-lm", "-o", str(exe)]
    built = subprocess.run(cmd, capture_output=True, text=True)
    if built.returncode != 0 and "sanitize" in built.stderr:
        pytest.skip("this gcc has no sanitizer runtime")
    assert built.returncode == 0, built.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", RT_HOST_THREADS="4")
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]
    assert "lighthouse leaf 7: rc 0 spheres 0 tris 5003" in run.stdout
    assert "threaded build: rc 0 tris 70000" in run.stdout

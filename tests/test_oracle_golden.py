"""The oracle and the host-side flattening against what the REFERENCE ITSELF produced.

tests/golden/ holds canvases, images, ray counts and scene dumps written by the
unmodified reference through oracle/ref_harness.c (tests/golden/make_golden.py).
The oracle must reproduce the double-precision canvas exactly: it is the same
arithmetic in the same order.
"""
import hashlib

import numpy as np
import pytest

import oracle_lib as O
import raytracer_amd as R
import refdump
from conftest import GOLDEN, load_golden, quantise

CANVAS_CASES = [
    "lighthouse_96x64", "cube_96x64", "spheres_96x64", "cube_spheres_96x64", "lighthouse_refl_96x64",
    "cube_spheres_refl_96x64", "cube_spheres_ss4_48x32", "spheres_ss2_48x32", "cube_101x37", "spheres_33x71",
    # planes, patterns, cube, cylinder, glass: worlds built by the reference's own scene functions
    "refraction_96x64", "world_with_plane_96x64", "dual_spheres_96x64", "refraction_ss2_48x32", "world_with_plane_77x45",
]


def scene_for(e, **kw):
    return R.stock_scene(e["scene"], e["width"], e["height"], e["tri_reflective"], e["sph_reflective"], **kw)


@pytest.mark.parametrize("name", CANVAS_CASES)
def test_oracle_canvas_is_bit_identical_to_reference(manifest, name):
    e = manifest[name]
    got, st = O.render_f64(scene_for(e), spp_grid=e["spp_grid"])
    want = load_golden(e, "canvas")
    assert np.array_equal(got, want)
    assert st.rays == e["rays"]
    p = O.probe()
    assert p.max_list == e["max_list"] and p.exact_zero_t == e["exact_zero_t"] and p.list_overflow == 0
    if not name.startswith("refraction"):
        assert p.exact_zero_t == 0  # only that scene puts ray origins exactly on a plane (RayTracer.c:168 quirk)


@pytest.mark.parametrize("name", ["cube_300", "spheres_300", "lighthouse_300", "refraction_300", "world_with_plane_300",
                                  "dual_spheres_300"])
def test_oracle_reproduces_reference_ppm(manifest, name):
    """Config 1: the reference's stock scenes at its built-in 300x300; same image, same PPM bytes."""
    e = manifest[name]
    got, st = O.render_rgb(scene_for(e))
    assert np.array_equal(got, load_golden(e, "rgb"))
    assert st.rays == e["rays"]
    ppm = R.ppm_bytes(got)
    assert len(ppm) == e["ppm_bytes"] and hashlib.sha256(ppm).hexdigest() == e["ppm_sha256"]


def test_golden_hashes_are_the_published_ones(manifest):
    """sha256 of canvas.ppm as written by `gcc -O2 RayTracer.c -lm` and its 1080p siblings."""
    assert manifest["lighthouse_300"]["ppm_sha256"] == "0000c6fc7ecde219aec726b487fcefe31ae5f642a4d751a4bab4067a49be03b0"
    assert manifest["cube_300"]["ppm_sha256"] == "47bfe7d12df6cfba5f627cf7533e72cde0831103c3a9532f8a0fce9ac0fa8d65"
    assert manifest["spheres_300"]["ppm_sha256"] == "3a1fc0b5caf37feace8a1ff8ba5fdeabb5f40a6125b0461e2304c4a543474a1a"
    assert manifest["lighthouse_1080p"]["ppm_sha256"] == "53eff59ff0d5cfc07b410acec5514dcb8c2a9ebdeefa2aafc13050d8c4ee9218"
    assert manifest["cube_1080p"]["ppm_sha256"] == "27cfd36d7d86f36a0a7e96c9f3f11452e5b6981874cbf787089bf1633614b410"
    assert manifest["spheres_1080p"]["ppm_sha256"] == "6e1745f223d2043b8fe0fa3b316a8174c49a1e1ba836d272f83ec5e2ac026cf2"
    for name in ("lighthouse_1080p", "cube_1080p", "spheres_1080p"):
        e = manifest[name]
        rgb = load_golden(e, "rgb")
        assert hashlib.sha256(R.ppm_bytes(rgb)).hexdigest() == e["ppm_sha256"]


@pytest.mark.parametrize("name", ["lighthouse_96x64", "cube_96x64", "spheres_96x64", "cube_spheres_96x64"])
def test_host_flattening_matches_reference_structs(manifest, name):
    """Inverses, triangle edges/normals, camera constants: same bits as the reference's own functions computed."""
    e = manifest[name]
    head, shapes = refdump.load_npz(f"{GOLDEN}/{e['scene_dump']}")
    s = scene_for(e)
    a = R.scene_arrays(s)
    assert s.camera.hsize == head["hsize"] and s.camera.vsize == head["vsize"]
    assert s.camera.pixel_size == head["pixel_size"]
    assert s.camera.half_width == head["half_width"] and s.camera.half_height == head["half_height"]
    assert np.array_equal(a["inv_view"], head["inv_view"])
    assert list(s.light.position) == list(head["light_position"][:3])
    assert list(s.light.intensity) == list(head["light_intensity"][:3])
    sph = np.nonzero(shapes["type"] == refdump.SPHERE)[0]
    tri = np.nonzero(shapes["type"] == refdump.TRIANGLE)[0]
    assert np.array_equal(a["sph_order"], sph) and np.array_equal(a["tri_order"], tri)
    assert np.array_equal(a["sph_inv"], shapes["inverse"][sph])
    for row0, field in ((0, "p1"), (3, "e1"), (6, "e2"), (9, "normal")):
        assert np.array_equal(a["tri_soa"][row0:row0 + 3].T, shapes[field][tri][:, :3]), field
    mats = [s.materials[i] for i in range(s.n_materials)]
    for kind, idxs, midx in (("sph", sph, a["sph_material"]), ("tri", tri, a["tri_material"])):
        for k, i in enumerate(idxs):
            m = mats[midx[k]]
            assert list(m.color) == list(shapes["color"][i][:3])
            assert (m.ambient, m.diffuse, m.specular, m.shininess, m.reflective) == tuple(
                float(shapes[f][i]) for f in ("ambient", "diffuse", "specular", "shininess", "reflective"))


@pytest.mark.parametrize("name", ["refraction_96x64", "world_with_plane_96x64", "dual_spheres_96x64"])
def test_host_flattening_of_general_worlds_matches_reference_structs(manifest, name):
    """Planes, cube, cylinder, patterns, glass: inverses, cylinder bounds, pattern inverses and the first
    object's inverse (the shape lighting() is handed, RayTracer.c:1489) have the reference's bits."""
    e = manifest[name]
    head, shapes, ext = refdump.load_npz_ext(f"{GOLDEN}/{e['scene_dump']}")
    s = scene_for(e)
    a = R.scene_arrays(s)
    assert np.array_equal(a["inv_view"], head["inv_view"])
    assert s.camera.pixel_size == head["pixel_size"]
    assert list(s.light.position) == list(head["light_position"][:3])
    assert s.n_spheres + s.n_tris + s.n_shapes == len(shapes)
    assert np.array_equal(a["first_object_inv"], shapes["inverse"][0])
    sph = np.nonzero(shapes["type"] == refdump.SPHERE)[0]
    assert np.array_equal(a["sph_order"], sph) and np.array_equal(a["sph_inv"], shapes["inverse"][sph])
    mats = [s.materials[i] for i in range(s.n_materials)]
    exts = a["materials_ext"]
    assert exts is not None

    def check_material(midx, i):
        m, x = mats[midx], exts[midx]
        assert list(m.color) == list(shapes["color"][i][:3])
        assert (m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, x.transparency, x.refractive_index) == tuple(
            float(shapes[f][i]) for f in ("ambient", "diffuse", "specular", "shininess", "reflective", "transparency",
                                          "refractive_index"))
        if ext["has_pattern"][i]:
            assert x.pattern_type == ext["pattern_type"][i]
            assert list(x.pattern_from) == list(ext["pattern_from"][i][:3])
            assert list(x.pattern_to) == list(ext["pattern_to"][i][:3])
            assert np.array_equal(np.array(x.pattern_inv[:]).reshape(4, 4), ext["pattern_inverse"][i])
        else:
            assert x.pattern_type == R._ffi.RT_PATTERN_NONE

    for k, i in enumerate(sph):
        check_material(a["sph_material"][k], i)
    for sh in a["shapes"]:
        i = sh.order
        assert sh.type == shapes["type"][i] and sh.type in (refdump.PLANE, refdump.CUBE, refdump.CYLINDER)
        assert np.array_equal(np.array(sh.inv[:]).reshape(4, 4), shapes["inverse"][i])
        if sh.type == refdump.CYLINDER:
            assert (sh.minimum, sh.maximum, sh.closed) == (ext["minimum"][i], ext["maximum"][i], ext["closed"][i])
        check_material(sh.material, i)


def test_inverse_matches_reference_inverse(manifest):
    """rt_mat4_inverse repeats mat4x4_inverse (RayTracer.c:601) operation for operation: the six sphere
    transforms of the stock scene (rotations, anisotropic scales) invert to the reference's bits."""
    head, shapes = refdump.load_npz(f"{GOLDEN}/{manifest['spheres_96x64']['scene_dump']}")
    for sh in shapes:
        assert np.array_equal(R.Mat4.inverse(sh["transform"]), sh["inverse"])
    assert np.array_equal(R.Mat4.inverse(head["view"]), head["inv_view"])
    assert R.Mat4.inverse(np.diag([1e-4, 1e-3, 1.0, 1.0])) is None  # |det| < 1e-6: "not invertible", RayTracer.c:596


def test_row_bands_reassemble_oracle(manifest):
    e = manifest["cube_spheres_refl_96x64"]
    s = scene_for(e)
    full, _ = O.render_rgb(s)
    out = np.zeros_like(full)
    for n_ranks, band in ((3, 5),):
        for rank in range(n_ranks):
            part, st = O.render_rgb(s, n_ranks=n_ranks, band_rows=band, rank=rank)
            ys = [y for y in range(e["height"]) if (y // band) % n_ranks == rank]
            assert part.shape[0] == len(ys) == st.rows_rendered
            out[ys] = part
    assert np.array_equal(out, full)
    assert quantise(load_golden(e, "canvas")).tobytes() == full.tobytes()

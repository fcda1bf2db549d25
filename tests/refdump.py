"""Reader for the scene dumps written by oracle/ref_harness.c (--dump-scene).

Test infrastructure: the dump holds a scene as the REFERENCE holds it (shape
list in world order, transforms, materials) plus the loop invariants computed
by the reference's own mat4x4_inverse.
"""
import numpy as np

SHAPE_DTYPE = np.dtype(
    [
        ("type", "<i4"),
        ("pad", "<i4"),
        ("transform", "<f8", (4, 4)),
        ("inverse", "<f8", (4, 4)),
        ("p1", "<f8", 4),
        ("p2", "<f8", 4),
        ("p3", "<f8", 4),
        ("e1", "<f8", 4),
        ("e2", "<f8", 4),
        ("normal", "<f8", 4),
        ("color", "<f8", 4),
        ("ambient", "<f8"),
        ("diffuse", "<f8"),
        ("specular", "<f8"),
        ("shininess", "<f8"),
        ("reflective", "<f8"),
        ("transparency", "<f8"),
        ("refractive_index", "<f8"),
    ]
)
HEAD_DTYPE = np.dtype(
    [
        ("magic", "S8"),
        ("hsize", "<i4"),
        ("vsize", "<i4"),
        ("fov", "<f8"),
        ("pixel_size", "<f8"),
        ("half_width", "<f8"),
        ("half_height", "<f8"),
        ("view", "<f8", (4, 4)),
        ("inv_view", "<f8", (4, 4)),
        ("light_position", "<f8", 4),
        ("light_intensity", "<f8", 4),
        ("n_shapes", "<i4"),
        ("pad", "<i4"),
    ]
)
# second section of a dump: cylinder bounds, pattern
EXT_DTYPE = np.dtype(
    [
        ("minimum", "<f8"),
        ("maximum", "<f8"),
        ("closed", "<i4"),
        ("has_pattern", "<i4"),
        ("pattern_type", "<i4"),
        ("pad", "<i4"),
        ("pattern_from", "<f8", 4),
        ("pattern_to", "<f8", 4),
        ("pattern_transform", "<f8", (4, 4)),
        ("pattern_inverse", "<f8", (4, 4)),
    ]
)
PLANE, SPHERE, TRIANGLE, CUBE, CYLINDER = 0, 1, 2, 3, 4  # enum shape_type, RayTracer.c:69


def parse(raw):
    head = np.frombuffer(raw, dtype=HEAD_DTYPE, count=1)[0]
    assert head["magic"] == b"RTSCENE1", head["magic"]
    n = int(head["n_shapes"])
    shapes = np.frombuffer(raw, dtype=SHAPE_DTYPE, count=n, offset=HEAD_DTYPE.itemsize)
    end = HEAD_DTYPE.itemsize + n * SHAPE_DTYPE.itemsize
    if end == len(raw):
        return head, shapes
    assert raw[end:end + 8] == b"RTEXT001" and end + 8 + n * EXT_DTYPE.itemsize == len(raw)
    return head, shapes, np.frombuffer(raw, dtype=EXT_DTYPE, count=n, offset=end + 8)


def load(path):
    """(head, shapes, ext); ext is None for dumps without the second section."""
    with open(path, "rb") as f:
        r = parse(f.read())
    return r if len(r) == 3 else (r[0], r[1], None)


def save_npz(path, head, shapes, ext=None):
    extra = {} if ext is None else {"ext": np.asarray(ext, dtype=EXT_DTYPE)}
    np.savez_compressed(path, head=np.array([head], dtype=HEAD_DTYPE), shapes=np.asarray(shapes, dtype=SHAPE_DTYPE), **extra)


def load_npz(path):
    z = np.load(path, allow_pickle=False)
    return z["head"][0], z["shapes"]


def load_npz_ext(path):
    z = np.load(path, allow_pickle=False)
    return z["head"][0], z["shapes"], (z["ext"] if "ext" in z.files else None)

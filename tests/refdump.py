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
SPHERE, TRIANGLE = 1, 2  # enum shape_type, RayTracer.c:69


def parse(raw):
    head = np.frombuffer(raw, dtype=HEAD_DTYPE, count=1)[0]
    assert head["magic"] == b"RTSCENE1", head["magic"]
    n = int(head["n_shapes"])
    shapes = np.frombuffer(raw, dtype=SHAPE_DTYPE, count=n, offset=HEAD_DTYPE.itemsize)
    assert HEAD_DTYPE.itemsize + n * SHAPE_DTYPE.itemsize == len(raw)
    return head, shapes


def load(path):
    with open(path, "rb") as f:
        return parse(f.read())


def save_npz(path, head, shapes):
    np.savez_compressed(path, head=np.array([head], dtype=HEAD_DTYPE), shapes=np.asarray(shapes, dtype=SHAPE_DTYPE))


def load_npz(path):
    z = np.load(path, allow_pickle=False)
    return z["head"][0], z["shapes"]

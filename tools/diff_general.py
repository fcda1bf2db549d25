"""Where the general kernel and the CPU oracle disagree on a stock scene (development aid; the oracle is test infrastructure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as R
import oracle_lib as O
name, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 5
s = R.stock_scene(name, w, h)
r = R.HipRenderer(0)
r.upload(s)
got, st = r.render_f64(max_depth=depth, count_rays=True)
want, ost = O.render_f64(s, max_depth=depth)
print("rays", st.rays, ost.rays)
d = np.abs(got - want).max(axis=2)
ys, xs = np.nonzero(d > 1e-13 * np.maximum(1e-3, np.abs(want).max(axis=2)))
print("differing pixels:", len(ys))
for y, x in list(zip(ys, xs))[:10]:
    print(y, x, got[y, x], want[y, x])
r.shutdown()

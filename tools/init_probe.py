"""rt_hip_init on device 0 with the library build named by RT_HIP_LIB (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler
faulthandler.enable()
import raytracer_amd as R
r = R.HipRenderer(0)
print("init ok", os.environ.get("RT_HIP_LIB"), flush=True)
r.upload(R.stock_scene("cube", 320, 200))
img, st = r.render(count_rays=True)
print("render ok", st.rays, flush=True)
r.shutdown()

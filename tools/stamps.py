"""Timeline of one launch from a build with -DRT_STAMPS=1 (make variant TAG=stamps EXTRA_HIPFLAGS=-DRT_STAMPS=1;
RT_HIP_LIB=stamps): when waves start and end relative to the first start, how long heavy waves run, per-XCD spread."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracer_amd as R

scene = sys.argv[1] if len(sys.argv) > 1 else "cube"
accel = {"auto": R._ffi.RT_ACCEL_AUTO, "brute": R._ffi.RT_ACCEL_BRUTE, "bvh": R._ffi.RT_ACCEL_BVH}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
tr = 0.3 if scene.endswith("_refl") else 0.0
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
hip = R.HipRenderer(0)
L = R._ffi.hip_lib()
hip.upload(R.stock_scene(scene.replace("_refl", ""), W, H, tr))
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
p = R.make_params(accel=accel)
hip.tune(p)
for _ in range(50):
    hip.render_device(buf.data_ptr(), p)
torch.cuda.synchronize()
NW = 65536
WORDS = L.rt_hip_debug_stamp_words()
raw = np.zeros((NW, WORDS), dtype=np.uint64)
L.rt_hip_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert L.rt_hip_debug_stamps(raw.ctypes.data, NW) == 0
used = raw[:, 3] > 0
s = raw[used].astype(np.int64)
w0, w1 = s[:, 2], s[:, 3]          # 100 MHz wall clock: 10 ns ticks
t0 = w0.min()
start_us, end_us = (w0 - t0) / 100.0, (w1 - t0) / 100.0
dur_us = end_us - start_us
cyc = (s[:, 1] - s[:, 0])
heavy = (s[:, 5] & 1) == 1
entry = (s[:, 5] >> 8) & 0xffffffff
xcc = (s[:, 4] >> 32) & 0xF
print(f"{scene}: {used.sum()} waves recorded, {heavy.sum()} in heavy workgroups")
print(f"last wave ends {end_us.max():.2f} us after the first starts; starts: median {np.median(start_us):.2f}, 90% {np.percentile(start_us, 90):.2f}, max {start_us.max():.2f} us")
for name, m in (("heavy", heavy), ("light", ~heavy)):
    if m.any():
        d = dur_us[m]
        print(f"{name:5s} waves: duration median {np.median(d):.2f} us, 90% {np.percentile(d, 90):.2f}, max {d.max():.2f}; "
              f"start median {np.median(start_us[m]):.2f}, max {start_us[m].max():.2f}; end max {end_us[m].max():.2f}; "
              f"shader-clock/wall ratio {np.median(cyc[m] / np.maximum(dur_us[m], 0.01)):.0f} ticks/us")
if WORDS > 6:  # -DRT_PHASES=1: shader-clock ticks per phase, summed over the heavy waves
    ph = s[heavy][:, 6:13].astype(np.float64)
    names = ["ray generation", "nearest hit", "hit point, shadow ray", "shadow ray", "lighting", "store", "start-up, staging", "-"]
    tot = ph.sum()
    print("phases of heavy waves (share of their ticks; median ticks per wave):",
          [(names[k], round(float(ph[:, k].sum() / tot), 3), int(np.median(ph[:, k]))) for k in range(7)])
if WORDS > 6:  # the waves that end last: where their ticks went
    order = np.argsort(-end_us)[:12]
    for i in order:
        e = int(entry[i])
        tal = int(s[i, 13])
        print(f"  wave ending at {end_us[i]:.2f} us (started {start_us[i]:.2f}, heavy {int(s[i, 5]) & 1}, tile {e & 0x3fffff} part {(e >> 22) & 63} of 2^{(e >> 28) & 7}): ticks per phase",
              [int(x) for x in s[i, 6:13]], "most per lane: camera-list confirms, shadow-list confirms, walk steps, leaf tests",
              [(tal >> (16 * k)) & 0xffff for k in range(4)])
print(f"sum of wave durations / 2048 slots: heavy {dur_us[heavy].sum() / 2048:.2f} us, light {dur_us[~heavy].sum() / 2048:.2f} us")
if WORDS > 6:
    for i in np.argsort(-dur_us)[:16]:
        e = int(entry[i]); tal = int(s[i, 13])
        print(f"  long wave {dur_us[i]:.2f} us (started {start_us[i]:.2f}, tile {e & 0x3fffff} part {(e >> 22) & 63} of 2^{(e >> 28) & 7}): ticks per phase",
              [int(x) for x in s[i, 6:13]], [(tal >> (16 * k)) & 0xffff for k in range(4)])
    lgs = (entry[heavy] >> 28) & 7
    for k in np.unique(lgs):
        d = dur_us[heavy][lgs == k]
        print(f"  heavy waves of entries split 2^{k}: {len(d)} waves, duration median {np.median(d):.2f}, max {d.max():.2f}, sum/2048 {d.sum() / 2048:.2f} us")
# where the hardware put the workgroups: HW_ID = wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13
hwid = s[:, 4] & 0xffffffff
simd_key = (xcc << 20) | (((hwid >> 13) & 7) << 16) | (((hwid >> 12) & 1) << 12) | (((hwid >> 8) & 15) << 4) | ((hwid >> 4) & 3)
cu_key = simd_key >> 4
blk = np.nonzero(used)[0] // 4
ends = {}
for k, e, d in zip(simd_key[heavy], end_us[heavy], dur_us[heavy]):
    a_ = ends.setdefault(int(k), [0, 0.0, 0.0]); a_[0] += 1; a_[1] = max(a_[1], float(e)); a_[2] += float(d)
last = np.array([v[1] for v in ends.values()]); cnt_ = np.array([v[0] for v in ends.values()])
print(f"SIMDs with heavy waves: {len(ends)}; heavy waves per SIMD min/median/max {cnt_.min()}/{int(np.median(cnt_))}/{cnt_.max()}; "
      f"last heavy end per SIMD: min {last.min():.2f}, 10% {np.percentile(last, 10):.2f}, median {np.median(last):.2f}, 90% {np.percentile(last, 90):.2f}, max {last.max():.2f} us")
if os.environ.get("RT_STAMPS_PLACEMENT"):
    for c in np.unique(cu_key)[:6]:
        m = cu_key == c
        print(f"  CU {int(c):#x}: workgroups {sorted(set(int(x) for x in blk[m]))[:24]}")
print("per XCD: waves, last end (us):", [(int(x), int((xcc == x).sum()), round(float(end_us[xcc == x].max()), 1)) for x in np.unique(xcc)])
hist, edges = np.histogram(end_us, bins=12)
print("ends histogram (us):", [(round(float(edges[i]), 1), int(hist[i])) for i in range(len(hist))])
busy = np.zeros(int(end_us.max() * 2) + 2)
for a_, b_ in zip(start_us, end_us):
    busy[int(a_ * 2):int(b_ * 2) + 1] += 1
print("waves in flight per 0.5 us:", [int(x) for x in busy])
hip.shutdown()

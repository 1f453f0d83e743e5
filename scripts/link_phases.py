"""Phases of k_link on a large table (3840x2160, ~5000 blobs; two-launch link path), from device realtime
stamps.  Needs a stamps build:  make -C ysmr_amd/csrc clean all EXTRA=-DYSMR_STAMPS OUT=libysmr_stamps.so
then YSMR_HIP_LIB=ysmr_amd/csrc/libysmr_stamps.so python scripts/link_phases.py"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
from ysmr_amd import _lib
F, B, H, W = 32, 8, 2160, 3840
frames = torch.from_numpy(SyntheticVideo(H, W, 5000, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=8192, capacity=8192, rows_per_flush=F * 8192)
def step():
    pipe.reset(); pending = None
    for f0 in range(0, F, B):
        nxt = (pipe.detect_async(frames[f0:f0 + B]), f0)
        if pending is not None:
            (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
        pending = nxt
    (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
for _ in range(2): step()
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 8192)(); n = ctypes.c_uint(0)
L.ysmr_debug_read_ring(buf, ctypes.byref(n))
a = np.array(buf[:], dtype=np.uint64).reshape(4096, 2)
cnt = int(n.value); k = min(cnt, 4096)
rec = a[:k]
names = ["init col tables", "claims", "ageing", "compaction", "registration", "bookkeeping"]
per = {}
for tag, t in rec:
    ph, fr = (int(tag) >> 40) - 10, int(tag) & 0xFFFFFFFF
    if 0 <= ph <= 6: per.setdefault(fr, {})[ph] = int(t)
rows = [[d[i + 1] - d[i] for i in range(6)] for f, d in sorted(per.items()) if len(d) == 7 and f >= 4]
rows = np.array(rows) / 100.0
print(f"{len(rows)} frames; per phase: median / mean / max  (us)")
for i, nm in enumerate(names):
    print(f"  {nm:18s} {np.median(rows[:, i]):7.2f} {rows[:, i].mean():7.2f} {rows[:, i].max():7.2f}")
print(f"  {'in-kernel total':18s} {np.median(rows.sum(1)):7.2f} {rows.sum(1).mean():7.2f} {rows.sum(1).max():7.2f}")

"""Unprofiled GPU-side timeline of batch boundaries (build with EXTRA=-DYSMR_STAMPS): device realtime
stamps of k_set_row_base and of the first/last k_frame of each batch."""
import sys, os, ctypes, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
from ysmr_amd import _lib
F, B, H, W = 512, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=2048, capacity=2048, rows_per_flush=F * 2048)
def step():
    pipe.reset(); pending = None
    for f0 in range(0, F, B):
        nxt = (pipe.detect_async(frames[f0:f0 + B]), f0)
        if pending is not None:
            (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
        pending = nxt
    (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("step ms", 1e3 * (time.perf_counter() - t0))
L = _lib.lib()
buf = (ctypes.c_ulonglong * 8192)(); n = ctypes.c_uint(0)
L.ysmr_debug_read_ring(buf, ctypes.byref(n))
a = np.array(buf[:], dtype=np.uint64).reshape(4096, 2)
cnt = int(n.value); k = min(cnt, 4096)
rec = a[:k] if cnt <= 4096 else np.roll(a, -(cnt & 4095), axis=0)
rec = rec[np.argsort(rec[:, 1])]
beg = {int(tag) & 0xFFFFFFFF: int(t) for tag, t in rec if int(tag) >> 40 == 2}
end = {int(tag) & 0xFFFFFFFF: int(t) for tag, t in rec if int(tag) >> 40 == 3}
# frames 192..255 (one steady-state batch): offset of each frame start within the batch, period, in-kernel time
f0 = 192
print("frame: start offset us | period us | block-0 in-kernel us")
for f in range(f0, f0 + 64):
    if f in beg and f + 1 in beg:
        print(f"{f:4d}: {(beg[f]-beg[f0])/100:8.1f} | {(beg[f+1]-beg[f])/100:6.1f} | {(end[f]-beg[f])/100:6.1f}")
# detection kernels (begin stamps) that started within that batch window, merged into the timeline
names = {1: "threshold", 2: "clear", 3: "list_begin", 4: "collect", 5: "union4", 6: "flag", 7: "union8", 8: "flatten",
         9: "rank", 10: "bbox_euler", 11: "holes", 12: "nested", 13: "geometry", 14: "compact"}
dbuf = (ctypes.c_ulonglong * 2048)(); dn = ctypes.c_uint(0)
if hasattr(L, "ysmr_debug_read_det_ring"):
    L.ysmr_debug_read_det_ring(dbuf, ctypes.byref(dn))
    d = np.array(dbuf[:], dtype=np.uint64).reshape(1024, 2)
    d = d[d[:, 1] > 0]
    t_lo, t_hi = beg[f0], beg[f0 + 63]
    ev = sorted((int(t), names.get(int(tag), str(tag))) for tag, t in d if t_lo - 30000 <= int(t) <= t_hi)
    print("detection kernel starts (us relative to frame %d):" % f0)
    print("  " + ", ".join(f"{n}@{(t - t_lo)/100:.0f}" for t, n in ev))
    # slow frames and the detection kernel running when they started
    for f in range(f0, f0 + 64):
        if f in beg and f + 1 in beg and beg[f + 1] - beg[f] > 1600:
            running = [n for t, n in ev if t <= beg[f]]
            print(f"  slow frame {f} (period {(beg[f+1]-beg[f])/100:.1f} us, in-kernel {(end[f]-beg[f])/100:.1f}) started during: {running[-1] if running else '-'}")

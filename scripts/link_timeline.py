"""Timeline of the two-launch link (k_link, k_track) on the 4K configuration from device realtime stamps: where the
time between a k_link's entry and the next one goes.  Needs a stamps build of the library (-DYSMR_STAMPS)."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
from ysmr_amd import _lib
B, H, W = 16, 2160, 3840
F = 2 * B
frames = torch.from_numpy(SyntheticVideo(H, W, 5000, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=8192, capacity=8192, rows_per_flush=4 * F * 8192)
res = [pipe.det[i].detect(frames[i * B:(i + 1) * B]) for i in range(2)]
torch.cuda.synchronize()
for _ in range(2):
    pipe.reset()
    for k in range(4): pipe.trk.run(res[k & 1].det, res[k & 1].det_count, k * B, pipe.rows, pipe.row_count)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_ulonglong * 8192)(); n = ctypes.c_uint(0)
L.ysmr_debug_read_ring(buf, ctypes.byref(n))
a = np.array(buf[:], dtype=np.uint64).reshape(4096, 2)
k = min(int(n.value), 4096)
ev = sorted((int(t), int(tag) >> 40, int(tag) & 0xFFFFFFFF) for tag, t in a[:k])
ev = ev[len(ev) // 2:]                      # second repetition
names = {9: "link entry", 10: "link counters loaded", 16: "link end", 4: "track entry"}
by = {}
for t, ph, fr in ev:
    if ph in names: by.setdefault(fr, {})[ph] = t
rows = []
for fr in sorted(by):
    d, nx = by[fr], by.get(fr + 1)
    if len(d) == 4 and nx and 9 in nx:
        rows.append([d[10] - d[9], d[16] - d[10], d[4] - d[16], nx[9] - d[4], nx[9] - d[9]])
r = np.array(rows) / 100.0
print(f"{len(r)} frames, link alone (no detection running); median / mean us")
for i, nm in enumerate(["link entry -> counters loaded", "link body", "link end -> track entry", "track entry -> next link entry", "frame to frame"]):
    print(f"  {nm:32s} {np.median(r[:, i]):7.2f} {r[:, i].mean():7.2f}")

"""Per-block begin / bookkeeping-done / end stamps of two chosen frames of the pipelined bench loop
(build with EXTRA='-DYSMR_STAMPS -DYSMR_BS_A=8 -DYSMR_BS_B=40'): frame A sits where detection of the
next batch runs (the slow one right after the threshold kernel), frame B in the quiet tail."""
import sys, os, ctypes, numpy as np, torch
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
for _ in range(3): step()
torch.cuda.synchronize()
L = _lib.lib()
NB = 2048
buf = (ctypes.c_ulonglong * (2 * NB * 8))()
L.ysmr_debug_read_block_stamps(buf, 2 * NB * 8)
both = np.array(buf[:], dtype=np.int64).reshape(2, NB, 8)
n_tracks = pipe.trk.info()[0]
nb = (n_tracks + 3) // 4 - 4
for name, a in (("frame A", both[0, :nb]), ("frame B", both[1, :nb])):
    rt0 = a[:, 4].min()
    rb, rm, re = (a[:, 4] - rt0) / 100.0, (a[:, 5] - rt0) / 100.0, (a[:, 6] - rt0) / 100.0      # us since the first block began
    dur = re - rb
    print(f"{name}: {nb} blocks | begin: med {np.median(rb):.1f} p90 {np.percentile(rb, 90):.1f} max {rb.max():.1f} us | "
          f"bookkeeping done: med {np.median(rm):.1f} max {rm.max():.1f} | end: med {np.median(re):.1f} p90 {np.percentile(re, 90):.1f} max {re.max():.1f} | "
          f"per-block duration med {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f} us")
    late = np.argsort(rb)[-6:]
    print("   latest-starting blocks:", late.tolist(), "begin", np.round(rb[late], 1).tolist(), "end", np.round(re[late], 1).tolist())

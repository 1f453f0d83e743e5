"""The frame feed inside the running pipeline: when each batch's read starts, how long it takes, when the consumer gets it."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd import frames as fr
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
F, B = (int(sys.argv[1]) if len(sys.argv) > 1 else 960), (int(sys.argv[2]) if len(sys.argv) > 2 else 64)
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "clip.npy"); np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(F))
v = fr.NpyVideo(path)
log = []
orig = v.read_into
def timed(start, count, out, pool=None):
    t0 = time.perf_counter(); n = orig(start, count, out, pool); log.append((t0, time.perf_counter())); return n
v.read_into = timed
for rep in range(3):
    log.clear()
    pipe = TrackingPipeline(922, 1228, 30.0, default_settings(), batch=B, rows_per_flush=F * 768)
    feed = fr.DeviceFrameFeed(v, B, "cuda:0")
    t0 = time.perf_counter(); got = []; pending = None
    for devt, f0, cnt, slot in feed:
        got.append(time.perf_counter())
        nxt = (pipe.detect_async(devt), f0)
        feed.release(slot, nxt[0][2])
        if pending is not None:
            pipe.link(pending[0][0], pending[0][1], pending[0][2], pending[1])
        pending = nxt
    pipe.link(pending[0][0], pending[0][1], pending[0][2], pending[1])
    torch.cuda.synchronize(); t1 = time.perf_counter()
    feed.close()
    print(f"rep {rep}: {F / (t1 - t0):.0f} frames/s; reads (start, duration ms):", [(round((a - t0) * 1e3, 1), round((b - a) * 1e3, 2)) for a, b in log])
    print("   consumer got batches at ms:", [round((g - t0) * 1e3, 1) for g in got])

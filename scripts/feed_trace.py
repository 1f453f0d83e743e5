"""Where the producer thread of DeviceFrameFeed spends a batch: waiting for its slot, reading, issuing the upload."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd import frames as fr
from ysmr_amd.synth import SyntheticVideo
F, B = 960, 64
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "clip.npy"); np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(64).repeat(F // 64, axis=0))
v = fr.NpyVideo(path)
log = []
orig = v.read_into
def timed(start, count, out, pool=None):
    t0 = time.perf_counter(); n = orig(start, count, out, pool); log.append((t0, time.perf_counter())); return n
v.read_into = timed
for rep in range(2):
    log.clear()
    feed = fr.DeviceFrameFeed(v, B, "cuda:0")
    t0 = time.perf_counter(); got = []
    for devt, f0, cnt, slot in feed:
        got.append(time.perf_counter())
        ev = torch.cuda.Event(); ev.record(); feed.release(slot, ev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    feed.close()
    print(f"rep {rep}: {F / (t1 - t0):.0f} frames/s; reads (start, duration ms):", [(round((a - t0) * 1e3, 2), round((b - a) * 1e3, 2)) for a, b in log])
    print("   consumer got batches at ms:", [round((g - t0) * 1e3, 2) for g in got])

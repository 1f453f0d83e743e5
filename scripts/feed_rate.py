"""Ingest rate of DeviceFrameFeed alone (file -> pinned -> HBM), and the phases of track_bacteria.
usage: python scripts/feed_rate.py [frames=1920] [readers=4] [depth=3]"""
import os, sys, time, tempfile, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd.frames import DeviceFrameFeed, open_video
from ysmr_amd.synth import SyntheticVideo
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
readers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = tempfile.mkdtemp(dir="/tmp")
base = SyntheticVideo(922, 1228, 500, seed=0).frames(64)
path = os.path.join(d, "clip.npy")
np.save(path, np.concatenate([base] * (F // 64)))
torch.cuda.init()
for rd in sorted({1, 2, readers, 8}):
    for rep in range(2):
        v = open_video(path)
        t0 = time.perf_counter()
        feed = DeviceFrameFeed(v, 64, "cuda:0", depth=depth, readers=rd)
        n = 0
        for dev, f0, cnt, slot in feed:
            ev = torch.cuda.Event(); ev.record()
            feed.release(slot, ev)
            n += cnt
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        feed.close(); v.close()
    print(f"readers={rd}: {n} frames in {dt*1e3:.0f} ms -> {n/dt:.0f} frames/s, {n*922*1228/dt/1e9:.1f} GB/s", flush=True)
# raw copies for reference
a = np.load(path, mmap_mode="r")
pin = torch.empty((64, 922, 1228), dtype=torch.uint8, pin_memory=True)
t0 = time.perf_counter()
for f0 in range(0, F, 64): np.copyto(pin.numpy(), a[f0:f0 + 64])
dt = time.perf_counter() - t0
print(f"mmap -> pinned, one thread: {F*922*1228/dt/1e9:.1f} GB/s")
devb = torch.empty_like(pin, device="cuda:0")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(F // 64): devb.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"pinned -> HBM: {F*922*1228/dt/1e9:.1f} GB/s")

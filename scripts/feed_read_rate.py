"""How fast do a video's frames get from the page cache into pinned memory, and from there to the GPU?  (f1: the frame loop
of track_bacteria on a .npy file.)  usage: feed_read_rate.py [frames]"""
import os, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.frames import NpyVideo, DeviceFrameFeed
from ysmr_amd.synth import SyntheticVideo
F = int(sys.argv[1]) if len(sys.argv) > 1 else 960
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "clip.npy"); np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(64).repeat(F // 64, axis=0))
v = NpyVideo(path)
B = 64
pinned = torch.empty((B, 922, 1228), dtype=torch.uint8, pin_memory=True)
pageable = np.empty((B, 922, 1228), np.uint8)
dev = torch.empty((B, 922, 1228), dtype=torch.uint8, device="cuda")
print("cpus usable", len(os.sched_getaffinity(0)), "load", os.getloadavg())
for name, dst in (("pinned", pinned.numpy()), ("pageable", pageable)):
    for nt in (1, 2, 4, 8, 16, 32):
        pool = ThreadPoolExecutor(nt)
        v.read_into(0, B, dst, pool)
        t0 = time.perf_counter()
        for f0 in range(0, F, B):
            v.read_into(f0, B, dst, pool)
        dt = time.perf_counter() - t0
        print(f"{name:9s} {nt:2d} threads: {F * 922 * 1228 / dt / 1e9:6.1f} GB/s  ({dt / (F // B) * 1e3:.2f} ms per batch)")
        pool.shutdown()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(F // B):
        dev.copy_(pinned, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"H2D pinned -> HBM: {F * 922 * 1228 / dt / 1e9:.1f} GB/s")
for readers in (4, 8, 16):
    feed = DeviceFrameFeed(v, B, "cuda:0", readers=readers)
    t0 = time.perf_counter(); n = 0
    for devt, f0, cnt, slot in feed:
        ev = torch.cuda.Event(); ev.record(); feed.release(slot, ev); n += cnt
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    feed.close()
    print(f"DeviceFrameFeed alone, {readers} readers: {n / dt:.0f} frames/s ({n * 922 * 1228 / dt / 1e9:.1f} GB/s)")

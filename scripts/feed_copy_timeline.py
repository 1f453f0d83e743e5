"""The frame feed's two halves on one time axis: every piece's read (page cache -> pinned memory, host clock) and its upload
(pinned -> HBM, events on the copy stream), for a 1920-frame 1228 x 922 file, no consumer: is the bus busy all the time, and at
what rate while the reads run?   argv: frames, frames per batch, pieces per batch, reader threads, 1 = process on the CPUs next
to the GPU first (ysmr_amd.dist.pin_to_gpu)"""
import os, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.frames import NpyVideo
from ysmr_amd.synth import SyntheticVideo
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
B = int(sys.argv[2]) if len(sys.argv) > 2 else 248
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4
T = int(sys.argv[4]) if len(sys.argv) > 4 else 16
if len(sys.argv) > 5 and sys.argv[5] == "1":
    from ysmr_amd import dist
    print("pinned to", len(dist.pin_to_gpu(0) or ()), "CPUs next to the GPU")
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "clip.npy"); np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(248).repeat((F + 247) // 248, axis=0)[:F])
v = NpyVideo(path)
pool = ThreadPoolExecutor(T)
pin = [torch.empty((B, 922, 1228), dtype=torch.uint8, pin_memory=True) for _ in range(3)]
dev = [torch.empty((B, 922, 1228), dtype=torch.uint8, device="cuda") for _ in range(3)]
s = torch.cuda.Stream()
MB = 922 * 1228 / 1e6
for rep in range(3):
    reads, evs = [], []
    up = [None] * 3
    torch.cuda.synchronize()
    origin = torch.cuda.Event(enable_timing=True); origin.record(s)
    t0 = time.perf_counter()
    step = -(-B // P)
    for i, f0 in enumerate(range(0, F, B)):
        slot = i % 3
        if up[slot] is not None:
            up[slot].synchronize()
        for c0 in range(0, B, step):
            want = min(step, B - c0)
            a = time.perf_counter()
            got = v.read_into(f0 + c0, want, pin[slot].numpy()[c0:c0 + want], pool)
            reads.append((a - t0, time.perf_counter() - a, got))
            if got:
                with torch.cuda.stream(s):
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    e0.record(s); dev[slot][c0:c0 + got].copy_(pin[slot][c0:c0 + got], non_blocking=True); e1.record(s)
                evs.append((e0, e1, got))
            if got < want:
                break
        up[slot] = evs[-1][1]
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"rep {rep}: host loop {t_host * 1e3:.1f} ms, all uploaded {t_all * 1e3:.1f} ms = {F * MB / 1e3 / t_all:.1f} GB/s over the run")
    if rep == 2:
        print("  reads  (start ms, ms, GB/s):", [(round(a * 1e3, 1), round(dt * 1e3, 2), round(g * MB / 1e3 / dt, 1)) for a, dt, g in reads])
        print("  copies (start ms, ms, GB/s):", [(round(origin.elapsed_time(e0), 1), round(e0.elapsed_time(e1), 2), round(g * MB / 1e3 / (e0.elapsed_time(e1) * 1e-3), 1)) for e0, e1, g in evs])

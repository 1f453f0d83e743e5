"""Do the page-cache -> pinned reads and the pinned -> HBM copies of the frame feed overlap?  Both at once, measured apart."""
import os, sys, tempfile, time, threading
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.frames import NpyVideo
from ysmr_amd.synth import SyntheticVideo
F, B = 960, 64
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "clip.npy"); np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(64).repeat(F // 64, axis=0))
v = NpyVideo(path)
pin = [torch.empty((B, 922, 1228), dtype=torch.uint8, pin_memory=True) for _ in range(4)]
dev = [torch.empty((B, 922, 1228), dtype=torch.uint8, device="cuda") for _ in range(2)]
GB = B * 922 * 1228 / 1e9
pool = ThreadPoolExecutor(8)
stop = False
def reader(result):
    n = 0; t0 = time.perf_counter()
    while not stop:
        v.read_into((n * B) % F, B, pin[2 + (n & 1)].numpy(), pool); n += 1
    result.append(n * GB / (time.perf_counter() - t0))
for both in (False, True):
    res = []
    stop = False
    th = threading.Thread(target=reader, args=(res,)) if both else None
    if th: th.start()
    s = torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for k in range(60):
            dev[k & 1].copy_(pin[k & 1], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stop = True
    if th: th.join()
    print(f"H2D {60 * GB / dt:.1f} GB/s" + (f" while reads run at {res[0]:.1f} GB/s" if both else " alone"))

import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
F, B, H, W = 64, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=2048, capacity=2048, rows_per_flush=F * 2048)
res = pipe.det[0].detect(frames); torch.cuda.synchronize()
acc = []
for rep in range(5):
    pipe.reset()
    pipe.trk.run(res.det, res.det_count, 0, pipe.rows, pipe.row_count); torch.cuda.synchronize()
    tail = pipe.rows[-160:].cpu().numpy().view(np.uint64)
    acc.append(np.diff(tail[:9].astype(np.int64)))
    sub = tail[[3, 10, 11, 12, 13, 4]].astype(np.int64)
    print("   bookkeeping split: claims+age %d | compaction %d | unused scan %d | set order %d | rest %d" % tuple(np.diff(sub)))
d = np.median(np.array(acc), axis=0)
names = ["n/m loads+exit test", "prefetch issue + phase A loads (order, gone, row_arg/min)", "claim atomics (LDS)", "claims/age/compaction/registration",
         "phase B: claim data", "GSFF", "order/gone/row write", "rowmin next frame"]
for n, v in zip(names, d):
    print(f"{n:60s} {v:8.0f} ticks  ({v/100:.2f} us at 100 MHz)")
print("total", d.sum() / 100, "us")

import ctypes
from ysmr_amd import _lib
L = _lib.lib()
buf = (ctypes.c_ulonglong * 32)()
acc = []
for rep in range(5):
    pipe.reset(); pipe.trk.run(res.det, res.det_count, 0, pipe.rows, pipe.row_count); torch.cuda.synchronize()
    L.ysmr_debug_read_stamps(buf)
    acc.append(np.diff(np.array(buf[:7], dtype=np.int64)))
    fir = np.array(buf[8:11], dtype=np.int64)
    print("   fir: products %d  wave sums %d" % (fir[1] - fir[0], fir[2] - fir[1]))
d = np.median(np.array(acc), axis=0)
for n, v in zip(["state loads", "fresh/grew", "likelihood exp + broadcast", "append + weights + output", "predict FIR", "write back"], d):
    print(f"  gsff: {n:40s} {v:8.0f} ticks")

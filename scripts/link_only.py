"""Link alone (no detection running next to it): us per frame of DeviceTracker.run over detections made beforehand.
    python scripts/link_only.py [H W blobs batch max_det capacity]      (defaults: the 4K configuration)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
a = [int(v) for v in sys.argv[1:]]
H, W, blobs, B, md, cap = (a + [2160, 3840, 5000, 16, 8192, 8192][len(a):])[:6]
F = 4 * B
frames = torch.from_numpy(SyntheticVideo(H, W, blobs, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=md, capacity=cap, rows_per_flush=2 * F * cap)
res = []
for i in range(2):      # two detector slots: keep the results of two different batches
    res.append(pipe.det[i].detect(frames[i * B:(i + 1) * B]))
torch.cuda.synchronize()
for _ in range(3):
    pipe.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(4): pipe.trk.run(res[k & 1].det, res[k & 1].det_count, k * B, pipe.rows, pipe.row_count)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"{W}x{H} ~{blobs} blobs: link alone {1e6 * (t1 - t0) / (4 * B):.1f} us/frame")

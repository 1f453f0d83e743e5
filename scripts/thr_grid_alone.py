"""k_threshold_mfma alone on the chip with the grid it takes beside the batch link (248 workgroups) and with its own (256)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.detect import Detector
from ysmr_amd.synth import SyntheticVideo
H, W, B, F = 922, 1228, 64, 512
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
for beside in (False, True, False, True):
    det = Detector(B, H, W, max_det=2048, beside_batch_link=beside)
    for f0 in range(0, F, B): det.threshold(frames[f0:f0 + B])
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        for f0 in range(0, F, B):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); det.threshold(frames[f0:f0 + B]); e1.record(); ts.append((e0, e1))
    torch.cuda.synchronize()
    ms = np.array([x.elapsed_time(y) for x, y in ts])
    print(f"beside_batch_link={beside}: median {np.median(ms) * 1e3:.1f} us, min {ms.min() * 1e3:.1f} us")

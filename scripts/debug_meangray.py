"""Step through the mean-gray branch stage by stage on small geometries (prints before every sync)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from ysmr_amd.detect import Detector, mean_gray_params

rng = np.random.default_rng(0)
for (h, w) in [(130, 1228), (97, 131), (64, 64), (40, 301), (5, 7), (3, 2), (1, 1)]:
    for ch in (1, 3):
        frames = rng.integers(0, 256, (16, h, w) if ch == 1 else (16, h, w, 3), dtype=np.uint8)
        det = Detector(16, h, w, max_det=1024, params=mean_gray_params(True, 5, 2.0))
        dev = torch.from_numpy(frames).cuda()
        print(h, w, ch, "threshold", flush=True)
        det.threshold(dev)
        torch.cuda.synchronize()
        print(h, w, ch, "levels", det.mean_levels[:4].cpu().tolist(), "components", flush=True)
        res = det.components(16)
        torch.cuda.synchronize()
        print(h, w, ch, "counts", res.det_count[:4].cpu().tolist(), flush=True)
print("done")

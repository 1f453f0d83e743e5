"""How often the headline clip registers tracks and how many columns the CPython set model orders when it does."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.detect import Detector, threshold_params
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.tracker import DeviceTracker
F, B, H, W = 512, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
det = Detector(B, H, W, max_det=2048, params=threshold_params(True, 5, 2.0))
trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=2048, max_det=2048)
nb = torch.zeros(1, dtype=torch.int32, device="cuda"); nn = torch.zeros(1, dtype=torch.int32, device="cuda")
stats = []
for f0 in range(0, F, B):
    res = det.detect(frames[f0:f0 + B])
    d, c = res.det.clone(), res.det_count.cpu().numpy()
    for k in range(B):
        trk.update(d[k], m=int(c[k]), frame=f0 + k, n_before=nb, n_new=nn)
        stats.append((f0 + k, int(c[k]), int(nb.item()), int(nn.item())))
a = np.array(stats)
reg = a[(a[:, 3] > 0) & (a[:, 0] > 0)]
print(f"{len(reg)} of {F - 1} frames register tracks; columns ordered per such frame: median {np.median(reg[:, 3]):.0f}, mean {reg[:, 3].mean():.1f}, "
      f"p90 {np.percentile(reg[:, 3], 90):.0f}, max {reg[:, 3].max()}")
print("frame, m, n before, new:", reg[:12].tolist())
print("frame 200:", a[200].tolist())

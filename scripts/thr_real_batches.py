"""Per-batch time of the threshold kernel on the bench clip (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd.detect import Detector
from ysmr_amd.synth import SyntheticVideo
H, W, B, F = 922, 1228, 64, 512
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
noise = (torch.randn(F, H, W, device="cuda", generator=g) * 2 + 40).round().clamp(0, 255).to(torch.uint8)
det = Detector(B, H, W, max_det=2048)
for name, clip in (("real", frames), ("noise", noise), ("real", frames)):
    for variant in (0, 2, 1):
        for f0 in range(0, F, B): det.threshold(clip[f0:f0 + B], variant=variant)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            row = []
            for f0 in range(0, F, B):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); det.threshold(clip[f0:f0 + B], variant=variant); e1.record(); row.append((e0, e1))
            torch.cuda.synchronize()
            ts.append([a.elapsed_time(b) * 1e3 for a, b in row])
        print(name, "variant", variant, "us per batch (min over 3 reps):", np.round(np.min(np.array(ts), axis=0), 1))

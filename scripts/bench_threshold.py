"""Micro-benchmark of the fused threshold kernel (a1-a3) alone: distinct frames >> L3, HIP events."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd.detect import Detector
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=512); ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--height", type=int, default=922); ap.add_argument("--width", type=int, default=1228)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--real", action="store_true")
ap.add_argument("--bgr", action="store_true", help="3-channel input (gray replicated)")
ap.add_argument("--variant", type=int, default=0, help="ysmr_threshold_batch_variant: 0 = matrix-pipe kernel, 1 = float32-chain strip kernel")
ap.add_argument("--beside", action="store_true", help="the grid the kernel takes beside the batch link (YSMR_BESIDE_BATCH_LINK: 248 workgroups)")
a = ap.parse_args()
H, W, B, F = a.height, a.width, a.batch, a.frames
if a.real:
    from ysmr_amd.synth import SyntheticVideo
    frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
else:  # background-like noise (40 +- 2), enough for timing
    g = torch.Generator(device="cuda").manual_seed(0)
    frames = (torch.randn(F, H, W, device="cuda", generator=g) * 2 + 40).round().clamp(0, 255).to(torch.uint8)
if a.bgr:
    frames = frames[:F // 2].unsqueeze(-1).expand(-1, -1, -1, 3).contiguous(); F = F // 2
det = Detector(B, H, W, max_det=2048, threshold_variant=a.variant, beside_batch_link=a.beside)
for f0 in range(0, F, B): det.threshold(frames[f0:f0 + B])
torch.cuda.synchronize()
ts = []
for _ in range(a.reps):
    for f0 in range(0, F, B):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); det.threshold(frames[f0:f0 + B]); e1.record(); ts.append((e0, e1))
torch.cuda.synchronize()
ms = np.array([x.elapsed_time(y) for x, y in ts])
alg = (4.0 if a.bgr else 2.0) * B * H * W
print(f"threshold {W}x{H} batch {B}: median {np.median(ms)*1e3:.1f} us  min {ms.min()*1e3:.1f} us  "
      f"{alg/np.median(ms)/1e6:.0f} GB/s algorithmic = {alg/np.median(ms)/1e6/8000:.3f} of 8 TB/s")

"""What the small kernels of the labelling chain find on the bench clip: residue pixels listed by k_windows (all frames / how
many frames have any), components per frame, components with holes (k_nested's queue), large boxes.  Reads the workspace
header (detect.hip: WsHeader) and the result tables after one detection call.
    gpurun -- python scripts/chain_counts.py [frames [max_det]]"""
import os
import struct
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from ysmr_amd.detect import Detector
from ysmr_amd.synth import SyntheticVideo

B = int(sys.argv[1]) if len(sys.argv) > 1 else 248
max_det = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
H, W = 922, 1228
dev = torch.device("cuda:0")
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0, fps=30.0).frames(B)).to(dev)
det = Detector(B, H, W, max_det=max_det, device=dev)
for _ in range(2):
    r = det.detect(frames)
torch.cuda.synchronize()
hdr = bytes(det._ws[:256].cpu().numpy())
magic, labels, mask, total, fault, c0, c1, pad, b, h, w, md, dense, n_big = struct.unpack_from("<4Q4I4i2I", hdr)
print(f"header: listed residue pixels {c0} (second list {c1}), dense {dense}, large boxes {n_big}, max_det {md}")
n = r.det_count.cpu()
print(f"components per frame: min {int(n.min())} mean {float(n.float().mean()):.1f} max {int(n.max())}")
lab = r.labels
m = r.mask
on = (m != 0).flatten(1).sum(1).cpu()
print(f"final-mask pixels per frame: mean {float(on.float().mean()):.0f}")
d = r.det.cpu()
k = torch.arange(d.shape[1])[None, :] < n[:, None]
wh = d[..., 2:4][k]
print(f"box sides of the detections (minAreaRect w, h): mean {float(wh.mean()):.1f}, max {float(wh.max()):.1f}; share with a side > 16: {float((wh.max(1).values > 16).float().mean()):.3f}")

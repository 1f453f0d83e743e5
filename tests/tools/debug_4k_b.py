"""Narrow down the 4K claim discrepancy (tracks 184 / 190, detection 194 of frame 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import ysmr_oracle as yo
from ysmr_amd import _lib
from ysmr_amd.detect import threshold_params
from ysmr_amd.synth import S4K
from ysmr_amd.tracker import DeviceTracker
yo.build()
frames = S4K(seed=1).frames(2)
p = threshold_params(True, 5, 2.0)
dets = [yo.detect_frame(fr, p.inv, p.t_low, p.t_high, p.use_high, 8192).det for fr in frames]

def run(d0, d1, cap=8192):
    trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=cap, max_det=cap)
    ot = yo.OracleTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3)
    rows1 = torch.empty(cap * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    claim = torch.empty(cap, dtype=torch.int32, device="cuda"); newc = torch.empty(cap, dtype=torch.int32, device="cuda")
    scal = torch.zeros(4, dtype=torch.int32, device="cuda")
    out = None
    for k, d in enumerate((d0, d1)):
        det = torch.from_numpy(np.ascontiguousarray(d, np.float32)).cuda()
        trk.update(det, m=len(d), frame=k, rows=rows1, n_rows=scal[0:1], claim=claim, n_before=scal[1:2], new_cols=newc, n_new=scal[2:3])
        nb = int(scal[1].item())
        dev_claim = claim[:nb].cpu().numpy()
        ids, xy, info, claims = ot.update(yo.det_to_rects(d))
        ref = -np.ones(nb, int)
        for r, c in claims: ref[r] = c
        out = np.flatnonzero(dev_claim != ref), dev_claim, ref
    return out

full = run(dets[0], dets[1])
print("full tables: differing rows", full[0])
for m in (4774, 4096, 2048, 1024, 512, 300, 195):
    diff, dc, rc = run(dets[0], dets[1][:m])
    print(f"frame-1 detections truncated to {m}: differing rows {diff[:6]}")
for n in (4811, 2048, 1025, 1024, 512, 200):
    diff, dc, rc = run(dets[0][:n], dets[1])
    print(f"frame-0 tracks truncated to {n}: differing rows {diff[:6]}")
diff, dc, rc = run(dets[0][:200], dets[1][:300], cap=512)
print("200 tracks, 300 detections, capacity 512 (fused path):", diff[:6])
print("---- minimal")
a = dets[0][[184, 190]]; b = dets[1][[194]]
print("tracks", a[:, :2], "detection", b[:, :2])
for cap in (512, 8192):
    diff, dc, rc = run(a, b, cap=cap)
    print("two tracks, one detection, capacity", cap, "device claims", dc, "oracle", rc)
diff, dc, rc = run(a[::-1], b, cap=512)
print("same, tracks in the other order: device", dc, "oracle", rc)
a2 = a.copy(); a2[:, :2] -= [2900.0, 2000.0]; b2 = b.copy(); b2[:, :2] -= [2900.0, 2000.0]
diff, dc, rc = run(a2, b2, cap=512)
print("same, shifted towards the origin: device", dc, "oracle", rc)
from ysmr_amd.tracker import CentroidTracker
ct = CentroidTracker(max_disappeared=30, fps=30.0, use_gsff=False, capacity=16, max_det=16)
ct.update(yo.det_to_rects(a)); ct.update(yo.det_to_rects(b)); print("GSFF off: claims", ct.last_claims)
ct = CentroidTracker(max_disappeared=30, fps=30.0, use_gsff=True, capacity=16, max_det=16)
ct.update(yo.det_to_rects(a)); print("positions after frame 0", ct.objects); ct.update(yo.det_to_rects(b)); print("GSFF on: claims", ct.last_claims)

import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ysmr_oracle as yo
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.detect import Detector, threshold_params
from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
from ysmr_amd import _lib
frames = 255 - SyntheticVideo(160, 200, 15, seed=5).frames(48)
p = yo.threshold_params(False, 5, 2.0)
dets = [yo.det_to_rects(yo.detect_frame(f, *p).det) for f in frames]
def run(eps, at):
    tr = yo.OracleTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3); out=[]
    for f, rects in enumerate(dets):
        if f == at: rects = [((x+eps, y), info) for (x,y),info in rects]
        ids, xy, info, _ = tr.update(rects)
        out.append((list(ids), xy.copy(), [t.gone for t in tr.tracks], [(t.gs.mode, None if t.gs.weights is None else t.gs.weights.copy(), None if t.gs.x_hat is None else t.gs.x_hat.copy()) for t in tr.tracks]))
    return out
a = run(0.0, 5); b = run(1e-13, 5)
det = Detector(16, 160, 200, max_det=512, params=threshold_params(False, 5, 2.0))
trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=512, max_det=512)
rows = torch.empty(48 * 512 * 40, dtype=torch.uint8, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
dev = torch.from_numpy(frames.copy()).cuda()
for f0 in range(0, 48, 16):
    res = det.detect(dev[f0:f0+16]); trk.run(res.det, res.det_count, f0, rows, cnt)
torch.cuda.synchronize()
got = rows_to_numpy(rows, int(cnt.item()))
i = 0; worst = {}
for f in range(48):
    ia, xa, ga, sa = a[f]; ib, xb, gb, sb = b[f]
    n = len(ia); g = got[i:i+n]; i += n
    assert list(g["track_id"]) == ia
    dg = np.abs(np.stack([g["x"], g["y"]], 1) - xa).max(axis=1)
    do = np.abs(xa - xb).max(axis=1)
    for k, tid in enumerate(ia):
        if dg[k] > 1e-6 or do[k] > 1e-6:
            print(f"frame {f} id {tid} gone {ga[k]} gpu-vs-oracle {dg[k]:.2e} oracle-vs-perturbed {do[k]:.2e} mode {sa[k][0]} w {np.round(sa[k][1],5)} xhat_x {np.round(sa[k][2][0],3)}")

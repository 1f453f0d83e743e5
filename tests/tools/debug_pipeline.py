import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import ysmr_oracle as yo
from ysmr_amd.detect import Detector, threshold_params
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
from ysmr_amd import _lib
n_frames, h, w = 48, 240, 320
frames = SyntheticVideo(h, w, 40, seed=7, dropout=0.05, speckle=0.05).frames(n_frames)
ref_rows, _ = yo.track_frames(frames, fps=30.0)
det = Detector(16, h, w, max_det=256, params=threshold_params(True, 5, 2.0))
trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=256, max_det=256)
rows = torch.empty(n_frames * 256 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
count = torch.zeros(1, dtype=torch.int64, device="cuda")
dev = torch.from_numpy(frames).cuda()
for f0 in range(0, n_frames, 16):
    res = det.detect(dev[f0:f0 + 16])
    trk.run(res.det, res.det_count, f0, rows, count)
torch.cuda.synchronize()
got = rows_to_numpy(rows, int(count.item()))
ref = np.array(ref_rows)
bad = np.nonzero(~np.isclose(got["x"], ref[:, 2], rtol=1e-9, atol=1e-9) | ~np.isclose(got["y"], ref[:, 3], rtol=1e-9, atol=1e-9))[0]
print("bad rows", len(bad))
for i in bad[:30]:
    print(got[i], ref[i])
ids = sorted(set(got["track_id"][bad])); print("ids", ids)
for tid in ids[:3]:
    m = got["track_id"] == tid
    print("track", tid)
    for g, r in zip(got[m], ref[m]):
        print(int(g["frame"]), g["disappeared"], g["x"], r[2], g["x"] - r[2], g["w"], r[4])

"""Which rows of the 4K configuration differ from the oracle, and what the tracker saw around them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import ysmr_oracle as yo
from ysmr_amd import _lib
from ysmr_amd.detect import Detector, threshold_params
from ysmr_amd.synth import S4K
from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
yo.build()
frames = S4K(seed=1).frames(3)
p = threshold_params(True, 5, 2.0)
det = Detector(3, 2160, 3840, max_det=8192, params=p)
trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=8192, max_det=8192)
rows = torch.empty(3 * 8192 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
count = torch.zeros(1, dtype=torch.int64, device="cuda")
res = det.detect(torch.from_numpy(frames).cuda())
trk.run(res.det, res.det_count, 0, rows, count)
torch.cuda.synchronize()
got = rows_to_numpy(rows, int(count.item()))
ot = yo.OracleTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, shadows=2)
ref = []
dets = []
for k, fr in enumerate(frames):
    fd = yo.detect_frame(fr, p.inv, p.t_low, p.t_high, p.use_high, 8192)
    dets.append(fd.det)
    ids, xy, info, claims = ot.update(yo.det_to_rects(fd.det))
    for i, tid in enumerate(ids):
        ref.append((k, tid, xy[i][0], xy[i][1], *map(float, info[i]), ot.last_sens[i], ot.tracks[i].gone))
    if k:
        print("frame", k, "tracks", len(ids), "dets", len(fd.det), "claims", len(claims))
ref = np.array(ref)
bad = np.flatnonzero((np.abs(got["x"] - ref[:, 2]) > 1e-6) | (np.abs(got["y"] - ref[:, 3]) > 1e-6))
print("rows", len(got), "bad", len(bad))
for i in bad[:10]:
    f, tid = int(ref[i, 0]), int(ref[i, 1])
    print("row", i, "frame", f, "id", tid, "got", got["x"][i], got["y"][i], got["w"][i], got["h"][i], got["disappeared"][i],
          "ref", ref[i, 2:7], "sens", ref[i, 7], "gone", ref[i, 8])
    prev = np.flatnonzero((ref[:, 0] == f - 1) & (ref[:, 1] == tid))
    if len(prev):
        px, py = ref[prev[0], 2], ref[prev[0], 3]
        d = dets[f]
        dist = np.sqrt((d[:, 0].astype(float) - px) ** 2 + (d[:, 1].astype(float) - py) ** 2)
        o = np.argsort(dist)[:4]
        print("   previous output", px, py, "nearest detections", [(int(j), float(d[j, 0]), float(d[j, 1]), float(dist[j])) for j in o])

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

# ---- the same three frames one update at a time, with the claims
print("---- per-frame updates with claims")
trk2 = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=8192, max_det=8192)
ot2 = yo.OracleTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3)
rows1 = torch.empty(8192 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
claim = torch.empty(8192, dtype=torch.int32, device="cuda")
newc = torch.empty(8192, dtype=torch.int32, device="cuda")
scal = torch.zeros(4, dtype=torch.int32, device="cuda")
for k in range(3):
    d = dets[k]
    det64 = torch.from_numpy(np.ascontiguousarray(d, np.float32)).cuda()
    ids_before = [t.tid for t in ot2.tracks]
    trk2.update(det64, m=len(d), frame=k, rows=rows1, n_rows=scal[0:1], claim=claim, n_before=scal[1:2], new_cols=newc, n_new=scal[2:3])
    nb = int(scal[1].item())
    dev_claim = claim[:nb].cpu().numpy()
    ids, xy, info, claims = ot2.update(yo.det_to_rects(d))
    ref_claim = -np.ones(nb, int)
    for r, c in claims:
        ref_claim[r] = c
    diff = np.flatnonzero(dev_claim != ref_claim)
    print("frame", k, "tracks before", nb, "claims differing", len(diff))
    for r in diff[:10]:
        print("   row", r, "id", ids_before[r], "device col", dev_claim[r], "oracle col", ref_claim[r])

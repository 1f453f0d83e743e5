import sys, numpy as np
sys.path.insert(0, '.')
from oracle import ysmr_oracle as yo
from ysmr_amd.tracker import CentroidTracker
rng = np.random.default_rng(0)
base = rng.uniform(100, 400, (12, 2))
ct = CentroidTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=64, max_det=64)
ot = yo.OracleTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3)
for f in range(45):
    det = base.copy()
    if f >= 20: det[:6] += 6.6
    if f >= 21: det = det[6:]
    rects = [((float(x), float(y)), (1.0, 1.0, 0.0)) for x, y in det]
    objs, _ = ct.update(rects)
    ids, xy, info, _ = ot.update(rects)
    g = np.array(list(objs.values()))
    gone = np.array([t.gone for t in ot.tracks]) > 0
    d = np.abs(g - xy).max(axis=1)
    if f >= 18: print(f, list(objs.keys()) == ids, "matched %.1e lost %.1e" % (d[~gone].max(initial=0), d[gone].max(initial=0)))

import sys, numpy as np
sys.path.insert(0, '.')
from oracle import ysmr_oracle as yo
from ysmr_amd.gsff import GaussianSumFIR
rng = np.random.default_rng(1)
n = 60; t = np.arange(n)
stream = np.stack([300 + 25*np.cos(t/9.0), 200 + 25*np.sin(t/9.0)], 1) + rng.normal(0, 0.2, (n, 2))
for lost_from in (15, 20, 21, 25, 35):
    f = GaussianSumFIR(delta_t=1/30.0, n_min=0, n_max=30, n_f=3)
    o = yo.OracleGSFF(delta_t=1/30.0, n_min=0, n_max=30, n_f=3); st = yo.GsffState()
    state = {}; pg = po = None; worst = []
    for k in range(n):
        zg = np.array(stream[k]) if k < lost_from else pg.copy()
        zo = np.array(stream[k]) if k < lost_from else po.copy()
        cg, state = f.correct(measurement=zg, **state); pg, state = f.predict(**state)
        co = o.correct(zo, st); po = o.predict(st)
        worst.append(np.abs(pg - po).max())
    print("lost_from", lost_from, " ".join(f"{w:.1e}" for w in worst[max(0,lost_from-2):lost_from+14]))

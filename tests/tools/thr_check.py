"""Quick look at the matrix-pipe threshold kernel against the oracle (development aid; the tests are in
tests/test_gpu_detect.py): prints where bytes differ, per variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ysmr_oracle as yo
from ysmr_amd.detect import Detector, threshold_params
yo.build()
rng = np.random.default_rng(0)
for (h, w) in [(64, 64), (97, 132), (45, 1228), (33, 1236), (200, 260)]:
    frames = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    frames[1] = rng.normal(40, 2, (h, w)).round().clip(0, 255).astype(np.uint8); frames[1, ::9, ::11] = 200
    p = threshold_params(True, 5, 2.0)
    ref = np.stack([yo.classify(yo.blur3(f), yo.adaptive_mean(yo.blur3(f)), p.inv, p.t_low, p.t_high, p.use_high) for f in frames])
    d = Detector(2, h, w, max_det=64, params=p)
    for variant in (3, 2, 0):
        got = d.threshold(torch.from_numpy(frames).cuda(), variant=variant).cpu().numpy()
        bad = np.argwhere(got != ref)
        print(f"{h}x{w} variant {variant}: {len(bad)} of {got.size} differ", flush=True)
        if len(bad):
            for f in range(2):
                m = (got[f] != ref[f])
                if m.any():
                    ys, xs = np.nonzero(m)
                    print(f"   frame {f}: {m.sum()} bad; rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}; per 16-row band:",
                          [int(m[r:r + 16].sum()) for r in range(0, h, 16)], "per 16-col block:", [int(m[:, c:c + 16].sum()) for c in range(0, min(w, 160), 16)])

"""Quick look at the matrix-pipe threshold kernel against the oracle (development aid; the tests are in
tests/test_gpu_detect.py): prints where bytes differ, per variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ysmr_oracle as yo
from ysmr_amd.detect import Detector, threshold_params
yo.build()
rng = np.random.default_rng(0)
for (h, w) in [(64, 64), (97, 132), (45, 1228), (33, 1236)]:
    frames = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    frames[1] = rng.normal(40, 2, (h, w)).round().clip(0, 255).astype(np.uint8); frames[1, ::9, ::11] = 200
    p = threshold_params(True, 5, 2.0)
    ref = np.stack([yo.classify(yo.blur3(f), yo.adaptive_mean(yo.blur3(f)), p.inv, p.t_low, p.t_high, p.use_high) for f in frames])
    d = Detector(2, h, w, max_det=64, params=p)
    for variant in (1, 3, 2, 0):
        got = d.threshold(torch.from_numpy(frames).cuda(), variant=variant).cpu().numpy()
        bad = np.argwhere(got != ref)
        print(f"{h}x{w} variant {variant}: {len(bad)} of {got.size} differ", flush=True)
        if len(bad):
            ys, xs = bad[:, 1], bad[:, 2]
            print("   rows", np.unique(ys)[:20], "cols", np.unique(xs)[:40])
            print("   first", [(tuple(b), int(got[tuple(b)]), int(ref[tuple(b)])) for b in bad[:6]])

"""Phase split (shader clocks, block 1) of ONE chosen frame of the pipelined loop: build with
EXTRA='-DYSMR_STAMPS -DYSMR_ST_FRAME=200'."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
F, B, H, W = 512, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=2048, capacity=2048, rows_per_flush=F * 2048)
def step():
    pipe.reset(); pending = None
    for f0 in range(0, F, B):
        nxt = (pipe.detect_async(frames[f0:f0 + B]), f0)
        if pending is not None:
            (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
        pending = nxt
    (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
names = ["counters round trip + exit test", "state loads issued, LDS filled", "claim atomics", "claims/ageing/compaction (+registration)",
         "pick up the claim", "GSFF", "row write", "next row minimum"]
for rep in range(3):
    step(); torch.cuda.synchronize()
    tail = pipe.rows[-160:].cpu().numpy().view(np.uint64).astype(np.int64)
    d = np.diff(tail[:9])
    print(" | ".join(f"{n}: {v/2400:.1f} us" for n, v in zip(names, d)))
    sub = tail[[3, 10, 11, 12, 13, 4]]
    n, nid, _ = pipe.trk.info()
    print("   split of the 4th phase: sweep %.1f | - | unused-column scan %.1f | set order (one thread) %.1f | rest %.1f us" % (
        (sub[1] - sub[0]) / 2400, (sub[3] - sub[2]) / 2400 if sub[3] > sub[2] else 0, (sub[4] - sub[3]) / 2400 if sub[4] > sub[3] else 0,
        (sub[5] - max(sub[4], sub[2])) / 2400))

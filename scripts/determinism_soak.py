"""The same work again and again must give the same bytes: (a) one detector fed the two halves of the bench clip alternately, 40
calls -- label map, mask, detections of every call hashed, compared with the first call on that half (the clearing in k_windows and
the writers of a call: one writer per pixel, whatever the order the waves run in); (b) the pipeline on the whole clip, 8 passes --
the rows hashed.  argv: calls, passes"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.detect import Detector
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
PASSES = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, H, W = 248, 922, 1228
clip = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(2 * B)).cuda()
def digest(*tensors):
    h = hashlib.sha1()
    for t in tensors:
        h.update(t.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()
det = Detector(B, H, W, max_det=2048, beside_batch_link=True)
first, bad = {}, 0
for k in range(CALLS):
    half = k & 1 if k % 5 else (k // 5) & 1          # (not strictly alternating: the same half twice in a row now and then)
    r = det.detect(clip[half * B:(half + 1) * B])
    torch.cuda.synchronize()
    n = r.det_count
    counts = n.cpu().numpy()
    d = digest(r.labels, r.mask, n, r.status, *[r.det[f, :int(c)] for f, c in enumerate(counts)], *[r.anchors[f, :int(c)] for f, c in enumerate(counts)])   # (the tables' live parts)
    if half not in first:
        first[half] = d
    elif d != first[half]:
        bad += 1; print(f"call {k} (half {half}): digest differs")
print(f"detector: {CALLS} calls, {bad} differ from the first call on their half")
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=2048, capacity=768, rows_per_flush=2 * B * 768)
ref, bad_rows = None, 0
for p in range(PASSES):
    pipe.reset(); pending = None
    for f0 in range(0, 2 * B, B):
        nxt = (pipe.detect_async(clip[f0:f0 + B], frames_ready=False), f0)
        if pending is not None:
            (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
        pending = nxt
    (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
    torch.cuda.synchronize()
    rows = pipe.take_rows(sort=True)
    d = hashlib.sha1(rows.tobytes()).hexdigest()
    if ref is None:
        ref = d; print(f"pipeline: {len(rows)} rows per pass")
    elif d != ref:
        bad_rows += 1; print(f"pass {p}: rows differ")
print(f"pipeline: {PASSES} passes, {bad_rows} differ from the first")
sys.exit(1 if bad or bad_rows else 0)

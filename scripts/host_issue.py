import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
F, B, H, W = 256, 64, 922, 1228
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
step(); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"issue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms  ({F} frames)")
# link only (detections already there)
res = pipe.det[0].detect(frames[:B]); torch.cuda.synchronize()
for _ in range(3):
    pipe.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(4): pipe.trk.run(res.det, res.det_count, k * B, pipe.rows, pipe.row_count)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"link only: issue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms ({4*B} frames) -> {1e3*(t2-t0)/(4*B)*1e3:.1f} us/frame")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(4): pipe.det[0].detect(frames[k * B:(k + 1) * B])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"detect only: issue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms ({4*B} frames) -> {1e3*(t2-t0)/(4*B)*1e3:.1f} us/frame")

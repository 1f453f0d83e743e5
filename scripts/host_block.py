"""Which host call of the pipelined loop blocks?  Per-iteration host time of detect_async and link."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
F, B, H, W = 512, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=2048, capacity=2048, rows_per_flush=F * 2048)
def step(log):
    pipe.reset(); pending = None
    for f0 in range(0, F, B):
        t0 = time.perf_counter()
        nxt = (pipe.detect_async(frames[f0:f0 + B]), f0)
        t1 = time.perf_counter()
        if pending is not None:
            (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
        t2 = time.perf_counter()
        log.append((1e6 * (t1 - t0), 1e6 * (t2 - t1)))
        pending = nxt
    (slot, res, ready), p0 = pending; pipe.link(slot, res, ready, p0)
step([]); torch.cuda.synchronize()
for _ in range(2):
    log = []; t0 = time.perf_counter(); step(log); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"issue {1e3*(t1-t0):.2f} ms total {1e3*(t2-t0):.2f} ms")
    print("  detect_async us:", [round(a) for a, b in log])
    print("  link us:        ", [round(b) for a, b in log])

"""Host time of the calls around one batch of the pipeline (where does host_enqueue_ms_per_step go?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
F, B, H, W = 256, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=2048, capacity=768, rows_per_flush=F * 768)
def t(fn, n=1):
    t0 = time.perf_counter(); r = fn(); return r, (time.perf_counter() - t0) * 1e6
for rep in range(3):
    pipe.reset(); torch.cuda.synchronize()
    pend = None
    for f0 in range(0, F, B):
        nxt, td = t(lambda: pipe.detect_async(frames[f0:f0 + B], frames_ready=False))
        tl = 0.0
        if pend is not None:
            _, tl = t(lambda: pipe.link(pend[0][0], pend[0][1], pend[0][2], pend[1]))
        pend = (nxt, f0)
        print(f"rep {rep} batch {f0 // B}: detect_async {td:.0f} us, link {tl:.0f} us")
    torch.cuda.synchronize()
res = pipe.det[0].detect(frames[:B]); torch.cuda.synchronize()
side = torch.cuda.Stream()
for name, fn in (("prepare on current stream", lambda: pipe.trk.prepare(res.det, res.det_count, 0)),
                 ("batched property", lambda: pipe.trk.batched)):
    _, us = t(lambda: [fn() for _ in range(50)])
    torch.cuda.synchronize()
    print(name, us / 50, "us per call")
with torch.cuda.stream(side):
    _, us = t(lambda: [pipe.trk.prepare(res.det, res.det_count, 0) for _ in range(50)])
torch.cuda.synchronize()
print("prepare on a side stream", us / 50, "us per call")

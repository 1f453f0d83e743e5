import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
F, B, H, W = 128, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
for gs in (False, True):
    s = default_settings(); s["disable gsff"] = not gs
    pipe = TrackingPipeline(H, W, 30.0, s, batch=B, max_det=2048, capacity=2048, rows_per_flush=4 * F * 2048)
    res = [pipe.det[i].detect(frames[i * B:(i + 1) * B]) for i in range(2)]
    torch.cuda.synchronize()
    for _ in range(3):
        pipe.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(4): pipe.trk.run(res[k & 1].det, res[k & 1].det_count, k * B, pipe.rows, pipe.row_count)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"gsff={gs}: link only {1e6*(t2-t0)/(4*B):.1f} us/frame")

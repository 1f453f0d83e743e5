"""In-kernel phases of the batch link (k_batch), one frame of a batch, per wave: s_memtime stamps of a stamps build
(scripts/build_stamps.sh; YSMR_HIP_LIB=scripts/var_stamps.so).  usage: batch_stamps.py [blobs] [capacity] [max_det]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd import _lib
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
blobs = int(sys.argv[1]) if len(sys.argv) > 1 else 450
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 512
md = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
F, B, H, W = 128, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, blobs, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=md, capacity=cap, rows_per_flush=F * cap)
res0 = pipe.det[0].detect(frames[:B]); res1 = pipe.det[1].detect(frames[B:]); torch.cuda.synchronize()
L = _lib.lib()
NW = int(os.environ.get("BL_WAVES", "16"))
buf = (ctypes.c_ulonglong * (NW * 16))()
names = ["dma issue", "search + key atomic", "barrier A", "clear next tables + id atomic", "barrier B", "claims / ageing / registration",
         "filter bank", "wait vmcnt(0)", "barrier D", "ranks + row"]
acc = []
# BESIDE=1: the stamped launch runs while another stream loops over the matrix-pipe threshold kernel on 248 workgroups, as
# in the pipeline (which phase of a frame the memory system's load stretches)
beside = os.environ.get("BESIDE") == "1"
if beside:
    from ysmr_amd.detect import Detector
    side = torch.cuda.Stream()
    det_b = Detector(B, H, W, max_det=md, beside_batch_link=True)
    many = torch.from_numpy(SyntheticVideo(H, W, blobs, seed=1).frames(256)).cuda()
for rep in range(12 if beside else 6):
    pipe.reset()
    pipe.trk.run(res0.det, res0.det_count, 0, pipe.rows, pipe.row_count)
    torch.cuda.synchronize()
    if beside:
        with torch.cuda.stream(side):
            for i in range(8): det_b.threshold(many[(i % 4) * B:(i % 4) * B + B])
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); pipe.trk.run(res1.det, res1.det_count, B, pipe.rows, pipe.row_count); t1.record()
    torch.cuda.synchronize()
    L.ysmr_debug_read_bstamps(buf)
    full = np.array(buf[:], dtype=np.int64).reshape(NW, 16)
    a = full[:, :11]
    acc.append(a)
    print("   fast path per wave:", (full[:, 11] - full[:, 1]).tolist(), " lanes left to the wave search:", full[:, 12].tolist())
    print("   over the launch's frames: lanes left to the wave search per wave", full[:, 13].tolist(), "= %.2f per frame; frames in which a wave ran it:" % (full[:, 13].sum() / B), full[:, 14].tolist())
    rt0, rt1, mt0, mt1 = int(full[0, 15]), int(full[1, 15]), int(full[2, 15]), int(full[3, 15])
    if rt1 > rt0:
        print(f"   shader clock over frames 8..56: {(mt1 - mt0) / (rt1 - rt0) * 100:.0f} MHz; {(rt1 - rt0) / 100 / 48:.2f} us and {(mt1 - mt0) / 48:.0f} cycles per frame")
    print(f"rep {rep}: launch pair {t0.elapsed_time(t1) * 1e3:.1f} us for {B} frames; frame (wave 0) {a[0, 10] - a[0, 0]} cycles")
a = np.mean(np.array(acc), axis=0) if beside else np.median(np.array(acc), axis=0)
d = np.diff(a, axis=1)
print("%-44s" % "phase (cycles of s_memtime, 100 MHz? no: shader clock)", " ".join(f"w{w:<6d}" for w in range(NW)))
for k, n in enumerate(names):
    print("%-44s" % n, " ".join(f"{d[w, k]:<7.0f}" for w in range(NW)))
print("%-44s" % "frame", " ".join(f"{a[w, 10] - a[w, 0]:<7.0f}" for w in range(NW)))
print(pipe.trk.info())

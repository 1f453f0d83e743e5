"""When the workgroups of one k_track launch (4K configuration, link alone) begin and end: per-block device stamps
of frames 61 and 62.  Needs EXTRA="-DYSMR_BS_A=61 -DYSMR_BS_B=62" scripts/build_stamps.sh, YSMR_HIP_LIB=scripts/var_stamps.so."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
from ysmr_amd import _lib
B, H, W = 16, 2160, 3840
F = 4 * B
frames = torch.from_numpy(SyntheticVideo(H, W, 5000, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=8192, capacity=8192, rows_per_flush=F * 8192)
res = []
for k in range(4):
    r = pipe.det[0].detect(frames[k * B:(k + 1) * B])
    res.append((r.det.clone(), r.det_count.clone()))
torch.cuda.synchronize()
L = _lib.lib()
NB = 2048
buf = (ctypes.c_ulonglong * (2 * NB * 8))()
for rep in range(3):
    pipe.reset()
    for k in range(4): pipe.trk.run(res[k][0], res[k][1], k * B, pipe.rows, pipe.row_count)
    torch.cuda.synchronize()
    L.ysmr_debug_read_block_stamps(buf, 2 * NB * 8)
    both = np.array(buf[:], dtype=np.int64).reshape(2, NB, 8)
    nb = (pipe.trk.info()[0] + 3) // 4 - 40
    a = both[1, :nb]
    t0 = a[:, 4].min()
    us = lambda col: (a[:, col] - t0) / 100.0
    q = lambda v: f"min {v.min():6.2f} p10 {np.percentile(v, 10):6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} p99 {np.percentile(v, 99):6.2f} max {v.max():6.2f}"
    print(f"frame 62, {nb} blocks (us since the first block's entry)")
    print("  entry            ", q(us(4)))
    print("  slot known       ", q(us(5)))
    print("  filters done     ", q(us(6)))
    print("  next minimum     ", q(us(7)))
    print("  entry -> end     ", q(us(7) - us(4)))
    print("  filters -> end   ", q(us(7) - us(6)))
    late = np.argsort(us(7))[-5:]
    print("  last five blocks:", [(int(b), round(float(us(4)[b]), 2), round(float(us(7)[b]), 2)) for b in late])

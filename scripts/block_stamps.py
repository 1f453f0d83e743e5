"""Per-block begin / end-of-bookkeeping / end stamps of the last k_frame launch (build with -DYSMR_STAMPS)."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import TrackingPipeline
from ysmr_amd import _lib
F, B, H, W = 64, 64, 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
pipe = TrackingPipeline(H, W, 30.0, default_settings(), batch=B, max_det=2048, capacity=2048, rows_per_flush=F * 2048)
res = pipe.det[0].detect(frames); torch.cuda.synchronize()
L = _lib.lib()
NB = 2048
buf = (ctypes.c_ulonglong * (2 * NB * 8))()
for rep in range(3):
    pipe.reset(); pipe.trk.run(res.det, res.det_count, 0, pipe.rows, pipe.row_count); torch.cuda.synchronize()
    L.ysmr_debug_read_block_stamps(buf, 2 * NB * 8)
    both = np.array(buf[:], dtype=np.int64).reshape(2, NB, 8)
    n_tracks = pipe.trk.info()[0]
    nb = (n_tracks + 3) // 4 - 2           # blocks certainly live in both of the last two frames
    prev, a = both[0, :nb], both[1, :nb]   # frame 62 (even) and frame 63 (odd)
    book, total = a[:, 1] - a[:, 0], a[:, 2] - a[:, 0]          # shader clocks, per block
    rt0 = a[:, 4].min()
    rb, re = (a[:, 4] - rt0) * 10, (a[:, 6] - rt0) * 10          # ns since the first block began (100 MHz counter)
    print(f"tracks {n_tracks} blocks {nb}: bookkeeping med {np.median(book):.0f} max {book.max()} clk | block total med {np.median(total):.0f} "
          f"p90 {np.percentile(total, 90):.0f} max {total.max()} clk ({total.max()/2400:.2f} us)")
    print(f"   block begin: med {np.median(rb):.0f} p90 {np.percentile(rb, 90):.0f} max {rb.max()} ns | block end: med {np.median(re):.0f} max {re.max()} ns")
    print(f"   previous launch: first begin {(prev[:, 4].min() - rt0) * 10} ns, last end {(prev[:, 6].max() - rt0) * 10} ns  -> gap {(rt0 - prev[:, 6].max()) * 10} ns, "
          f"launch period {(rt0 - prev[:, 4].min()) * 10} ns")

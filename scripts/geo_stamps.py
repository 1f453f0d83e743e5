import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd.detect import Detector
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd import _lib
frames = torch.from_numpy(SyntheticVideo(922, 1228, 500, seed=0).frames(64)).cuda()
det = Detector(64, 922, 1228, max_det=2048)
buf = (ctypes.c_ulonglong * 16)(); acc = []
for _ in range(5):
    det.detect(frames); torch.cuda.synchronize()
    _lib.lib().ysmr_debug_read_geo_stamps(buf); acc.append(np.diff(np.array(buf[:5], dtype=np.int64)))
print("entry/bbox loads, column scan + sync, chains, calipers:", np.median(np.array(acc), axis=0))

"""When the detection kernels BEGIN (device clock, block 0's first instruction; stamps build: scripts/build_stamps.sh, run with
YSMR_HIP_LIB=scripts/var_stamps.so): the begin-to-begin intervals of the labelling chain on the bench clip, detection alone.
A kernel's interval = its own run + whatever lies between its end and the next kernel's first instruction."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd import _lib
from ysmr_amd.detect import Detector
from ysmr_amd.synth import SyntheticVideo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H, W = 922, 1228
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(2 * B)).cuda()
det = Detector(B, H, W, max_det=2048)
for rep in range(8):
    det.detect(frames[(rep & 1) * B:(rep & 1) * B + B])
torch.cuda.synchronize()
L = _lib.lib()
dbuf = (ctypes.c_ulonglong * 2048)(); dn = ctypes.c_uint(0)
L.ysmr_debug_read_det_ring(dbuf, ctypes.byref(dn))
d = np.array(dbuf[:], dtype=np.uint64).reshape(1024, 2)
d = d[d[:, 1] > 0]
d = d[np.argsort(d[:, 1])]
names = {1: "threshold", 2: "clear", 4: "windows", 5: "residue", 9: "rank", 12: "nested", 13: "geometry", 14: "compact"}
ev = [(int(t), names.get(int(tag), str(int(tag)))) for tag, t in d]
# the last three chains: the first stamp of every kernel (k_rank's grid has several workgroups with blockIdx.x == 0)
starts = [i for i, (t, n) in enumerate(ev) if n == "clear"][-4:]
for a, b in zip(starts[:-1], starts[1:]):
    chain, seen = [], set()
    for t, n in ev[a:b]:
        if n not in seen:
            seen.add(n); chain.append((t, n))
    print("  ".join(f"{n} +{(chain[i + 1][0] - t) / 100:.1f}" if i + 1 < len(chain) else f"{n} (next chain's clear +{(ev[b][0] - t) / 100:.1f})" for i, (t, n) in enumerate(chain)))

"""Phases of k_threshold_mfma's walk in ONE workgroup (block 100), per wave and 16-row step: s_memtime stamps of a stamps
build (scripts/build_stamps.sh; YSMR_HIP_LIB=scripts/var_stamps.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ysmr_amd import _lib
from ysmr_amd.detect import Detector
from ysmr_amd.synth import SyntheticVideo
H, W, B, F = 922, 1228, 64, 128
frames = torch.from_numpy(SyntheticVideo(H, W, 500, seed=0).frames(F)).cuda()
det = Detector(B, H, W, max_det=2048)
for f0 in range(0, F, B): det.threshold(frames[f0:f0 + B])
torch.cuda.synchronize()
L = _lib.lib()
NW, NS, NK = int(os.environ.get("TM_WAVES", "16")), 20, 8
buf = (ctypes.c_ulonglong * (NW * NS * NK))()
acc = []
for rep in range(5):
    det.threshold(frames[(rep % 2) * B:(rep % 2) * B + B]); torch.cuda.synchronize()
    L.ysmr_debug_read_thr_stamps(buf)
    acc.append(np.array(buf[:], dtype=np.int64).reshape(NW, NS, NK))
a = np.median(np.array(acc), axis=0)          # [wave][step][stamp]
names = ["stores of the previous step + DMA issue", "filter (column + row pass, classify)", "wait for the rows of the next step", "(class-map stores: at the step's start since r05)",
         "barrier 1", "blur of the next block", "barrier 2"]
steps = range(3, 13)                            # steady state
d = np.diff(a[:, steps, :], axis=2).mean(axis=1)     # [wave][phase]
print("cycles per 16-row step (s_memtime = 100 MHz x 24?  see the step total), steady-state steps 3..12, one workgroup")
print("%-44s" % "phase", " ".join(f"w{w:<5d}" for w in range(NW)))
for k, n in enumerate(names):
    print("%-44s" % n, " ".join(f"{d[w, k]:<6.0f}" for w in range(NW)))
tot = (a[:, 12, 7] - a[:, 3, 0]) / 10.0
print("%-44s" % "step", " ".join(f"{tot[w]:<6.0f}" for w in range(NW)))
print("first stamp of step 0 to the last of step 14 (wave 0):", a[0, 14, 7] - a[0, 0, 0])
rt0, mt0, rt1, mt1 = a[1, 18, 0], a[1, 18, 1], a[1, 19, 0], a[1, 19, 1]
if rt1 > rt0:
    print(f"shader clock over steps 3..13 of that workgroup: {(mt1 - mt0) / (rt1 - rt0) * 100:.0f} MHz ({mt1 - mt0:.0f} cycles in {(rt1 - rt0) / 100:.2f} us)")

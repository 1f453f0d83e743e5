"""track_bacteria on a 1920-frame 1228x922 .npy clip against the frames per batch (`batch=`): ms per pass, frames/s."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import track_bacteria
from ysmr_amd import track_eval as _te
F = 1920
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "clip.npy"); np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(F))
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
track_bacteria(path, settings=dict(s), result_folder=d)
for b in (32, 64, 128, 256, 64, 128):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); res = track_bacteria(path, settings=dict(s), result_folder=d, batch=b); dt = time.perf_counter() - t0
        best = min(best, dt)
        marks = {k: round(v * 1e3, 1) for k, v in _te.LAST_PASS_MARKS.items()}
    print(f"batch {b}: best of 3 {best*1e3:.0f} ms -> {F/best:.0f} frames/s; last pass marks {marks}")

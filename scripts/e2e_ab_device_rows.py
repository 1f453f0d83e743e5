"""track_bacteria on a 1920-frame 1228 x 922 file with the rows printed on the device (ysmr_rows_format_device, the default since
the end of round 5) against the host path ('hip print rows on device' = False), alternating on one box: best and all of five
warm runs each, and the phases of the last."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd import track_eval as te
d = tempfile.mkdtemp(dir="/tmp"); path = os.path.join(d, "clip.npy")
np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(1920))
base = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False, "log_level": 40})
res = {True: [], False: []}
for rep in range(6):
    for on_dev in (True, False):
        s = dict(base); s["hip print rows on device"] = on_dev
        t0 = time.perf_counter(); out = te.track_bacteria(path, settings=s, result_folder=d); dt = time.perf_counter() - t0
        assert out is not None
        if rep: res[on_dev].append(dt)
        m = dict(te.LAST_PASS_MARKS)
        if rep == 5:
            print(("device" if on_dev else "host  "), "last run marks (ms):", {k: round(v * 1e3, 1) for k, v in m.items()}, "csv bytes", os.path.getsize(out[4]))
for on_dev in (True, False):
    r = res[on_dev]
    print(("rows printed on the device:" if on_dev else "rows printed on the host:  "), f"best {min(r)*1e3:6.1f} ms = {1920/min(r):7.0f} frames/s   all {[round(t*1e3) for t in r]}")

import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd import track_eval as te
from ysmr_amd import helper_file as hf
d = tempfile.mkdtemp(dir="/tmp"); path = os.path.join(d, "clip.npy")
np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(int(sys.argv[1]) if len(sys.argv) > 1 else 496))
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False, "log_level": 40, "minimal frame count": 40})
for rep in range(3):
    t0 = time.perf_counter(); out = te.track_bacteria(path, settings=dict(s), result_folder=d); dt = time.perf_counter() - t0
    print(rep, round(dt*1e3), "ms", {k: round(v*1e3,1) for k,v in hf.LAST_DEVICE_ROWS_MARKS.items()}, {k: round(v*1e3,1) for k,v in te.LAST_PASS_MARKS.items()}, flush=True)

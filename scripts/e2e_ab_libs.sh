#!/bin/bash
# track_bacteria on a 1920-frame file with the libraries given (same box, alternating): best of four warm runs each
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "$@"; do
YSMR_HIP_LIB=$lib python3 - <<PY
import os, sys, time, tempfile
sys.path.insert(0, ".")
import numpy as np
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd import track_eval as te
d = tempfile.mkdtemp(dir="/tmp"); path = os.path.join(d, "clip.npy")
np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(1920))
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False, "log_level": 40})
ts = []
for rep in range(5):
    t0 = time.perf_counter(); res = te.track_bacteria(path, settings=dict(s), result_folder=d); ts.append(time.perf_counter() - t0)
m = te.LAST_PASS_MARKS
print(f"$lib: best {min(ts[1:])*1e3:6.1f} ms = {1920/min(ts[1:]):7.0f} frames/s   all {[round(t*1e3) for t in ts]}   last run: linked at {m.get('last batch linked',0)*1e3:.0f} ms", flush=True)
PY
done
done

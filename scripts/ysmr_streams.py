"""ysmr() over several full-size clips on one GPU: one after the other vs several streams at a time.
usage: python scripts/ysmr_streams.py [clips=6] [frames=640]"""
import os, sys, time, tempfile, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ysmr_amd import ysmr
from ysmr_amd.helper_file import default_settings
from ysmr_amd.main import _OFFLINE_KEYS
from ysmr_amd.synth import SyntheticVideo
n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 6
F = int(sys.argv[2]) if len(sys.argv) > 2 else 640
d = tempfile.mkdtemp(dir="/tmp")
paths = []
for i in range(n_clips):
    p = os.path.join(d, f"clip{i}.npy")
    np.save(p, SyntheticVideo(922, 1228, 500, seed=i).frames(F))
    paths.append(p)
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False,
                        "minimal frame count": 10, **{k: False for k in _OFFLINE_KEYS}})
logging.getLogger("ysmr").setLevel(logging.WARNING)
ysmr(paths[:1], settings=dict(s), result_folder=os.path.join(d, "warm"))
for label, kw in (("one at a time", dict(multiprocess=False)), ("2 streams", dict(multiprocess=True, streams_per_gpu=2)),
                  ("3 streams", dict(multiprocess=True, streams_per_gpu=3))):
    t0 = time.perf_counter()
    done = ysmr(paths, settings=dict(s), result_folder=os.path.join(d, label.replace(" ", "_")), **kw)
    dt = time.perf_counter() - t0
    assert all(r is True for _, r in done)
    print(f"{label:14s}: {n_clips} clips x {F} frames in {dt*1e3:.0f} ms -> {n_clips*F/dt:.0f} frames/s (files, H2D, csv included)", flush=True)

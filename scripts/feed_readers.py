import sys, os, time, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from ysmr_amd import frames as fm, track_eval
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
F = 1920
d = tempfile.mkdtemp(dir="/tmp")
np.save(os.path.join(d, "clip.npy"), SyntheticVideo(922, 1228, 500, seed=0).frames(F))
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
orig = fm.DeviceFrameFeed.__init__
for readers, depth in ((4, 3), (8, 3), (12, 3), (8, 4), (16, 4)):
    def init(self, video, batch, device, depth=depth, readers=readers, _o=orig): _o(self, video, batch, device, depth=depth, readers=readers)
    fm.DeviceFrameFeed.__init__ = init
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = track_eval.track_bacteria(os.path.join(d, "clip.npy"), settings=dict(s), result_folder=d)
        best = min(best, time.perf_counter() - t0)
    print(f"readers={readers} depth={depth}: {best*1e3:.0f} ms -> {F/best:.0f} frames/s", flush=True)

"""End-to-end track_bacteria on a .npy clip (host file -> H2D -> detect+link -> sorted csv + DataFrame)."""
import sys, os, time, tempfile, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import track_bacteria
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
d = tempfile.mkdtemp(dir="/tmp")
t0 = time.perf_counter()
frames = SyntheticVideo(922, 1228, 500, seed=0).frames(F)
if ch == 3: frames = np.repeat(frames[..., None], 3, axis=-1)
path = os.path.join(d, "clip.npy"); np.save(path, frames); del frames
print(f"generated {F} frames in {time.perf_counter()-t0:.1f} s, {os.path.getsize(path)/1e6:.0f} MB")
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
logging.getLogger("ysmr").setLevel(logging.DEBUG)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = track_bacteria(path, settings=dict(s), result_folder=d)
    dt = time.perf_counter() - t0
    df = res[0]
    print(f"run {rep}: {dt*1e3:.0f} ms  -> {F/dt:.0f} frames/s end to end incl. file read, H2D, csv ({len(df)} rows, {os.path.getsize(res[4])/1e6:.1f} MB csv)")

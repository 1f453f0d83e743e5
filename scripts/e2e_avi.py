"""End-to-end track_bacteria on an uncompressed 8-bit AVI (bottom-up DIB frames, what microscope cameras write):
host file -> pinned memory -> H2D -> unpack on the device -> detect+link -> sorted csv + DataFrame, against the
same clip as .npy and against the host unpacking path (YSMR_HOST_UNPACK=1: DeviceFrameFeed without raw_layout)."""
import sys, os, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from avi_tools import write_avi
from ysmr_amd import frames as frames_mod
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import track_bacteria
F = int(sys.argv[1]) if len(sys.argv) > 1 else 960
d = tempfile.mkdtemp(dir="/tmp")
clip = SyntheticVideo(922, 1228, 500, seed=0).frames(F)
np.save(os.path.join(d, "clip.npy"), clip)
write_avi(os.path.join(d, "clip.avi"), clip, 8, fps=(30, 1))
del clip
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
def run(name, label):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = track_bacteria(os.path.join(d, name), settings=dict(s), result_folder=d)
        best = min(best, time.perf_counter() - t0)
    print(f"{label}: {best * 1e3:.0f} ms -> {F / best:.0f} frames/s ({len(res[0])} rows)")
run("clip.npy", ".npy (memory-mapped)")
run("clip.avi", ".avi, unpacked on the device")
frames_mod.AviVideo.raw_layout = property(lambda self: None)
run("clip.avi", ".avi, unpacked by the host reader")

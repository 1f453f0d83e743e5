import sys, os, time, tempfile, logging, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import track_bacteria
F = 640
d = tempfile.mkdtemp(dir="/tmp")
frames = SyntheticVideo(922, 1228, 500, seed=0).frames(F)
path = os.path.join(d, "clip.npy"); np.save(path, frames); del frames
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
logging.getLogger("ysmr").setLevel(logging.WARNING)
track_bacteria(path, settings=dict(s), result_folder=d)
pr = cProfile.Profile(); pr.enable()
track_bacteria(path, settings=dict(s), result_folder=d)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

"""Where the time of track_bacteria on a file goes (f1): cProfile of one warm run on a 1228x922 .npy clip,
the functions with the largest own and cumulative host time."""
import sys, os, time, tempfile, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import track_bacteria
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "clip.npy"); np.save(path, SyntheticVideo(922, 1228, 500, seed=0).frames(F))
s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
for rep in range(3):
    t0 = time.perf_counter(); res = track_bacteria(path, settings=dict(s), result_folder=d); dt = time.perf_counter() - t0
    print(f"run {rep}: {dt*1e3:.0f} ms -> {F/dt:.0f} frames/s ({len(res[0])} rows)")
    from ysmr_amd import track_eval as _te
    print("   marks (ms since the pass began):", {k: round(v * 1e3, 1) for k, v in _te.LAST_PASS_MARKS.items()})
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); res = track_bacteria(path, settings=dict(s), result_folder=d); dt = time.perf_counter() - t0
pr.disable()
print(f"profiled run: {dt*1e3:.0f} ms")
for key in ("cumulative", "tottime"):
    out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats(key).print_stats(18)
    print("\n".join(l[:150] for l in out.getvalue().splitlines() if l.strip() and "Ordered by" not in l and "function calls" not in l))

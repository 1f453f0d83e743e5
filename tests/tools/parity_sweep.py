"""One-off large parity run: the bench clip (1228x922, ~500 blobs) through track_bacteria vs the CPU oracle
on the same frames, for gray and BGR input and both threshold branches.
usage: python tests/tools/parity_sweep.py [frames=200]"""
import os, sys, time, tempfile, logging
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import compare_rows
from oracle import ysmr_oracle as yo
from ysmr_amd import _lib
from ysmr_amd.helper_file import default_settings
from ysmr_amd.synth import SyntheticVideo
from ysmr_amd.track_eval import track_bacteria

F = int(sys.argv[1]) if len(sys.argv) > 1 else 200
yo.build()
logging.getLogger("ysmr").setLevel(logging.WARNING)
frames = SyntheticVideo(922, 1228, 500, seed=0, fps=30.0).frames(F)
d = tempfile.mkdtemp(dir="/tmp")
for name, adt, clip in (("gray adaptive", 2.0, frames), ("gray mean-level", -1.0, frames),
                        ("bgr adaptive", 2.0, np.repeat(frames[:F // 2, ..., None], 3, axis=-1))):
    path = os.path.join(d, name.replace(" ", "_") + ".npy")
    np.save(path, clip)
    s = default_settings(**{"user input": False, "select files": False, "display video analysis": False,
                            "log to file": False, "minimal frame count": 10, "adaptive double threshold": adt})
    t0 = time.perf_counter()
    res = track_bacteria(path, settings=dict(s), result_folder=d)
    t1 = time.perf_counter()
    ref_rows, _ = yo.track_frames(clip, fps=30.0, adt=adt, shadows=2)
    t2 = time.perf_counter()
    df = res[0].sort_values(["POSITION_T", "TRACK_ID"]).reset_index(drop=True)
    rows = np.zeros(len(df), _lib.ROW_DTYPE)
    rows["frame"], rows["track_id"] = df["POSITION_T"], df["TRACK_ID"]
    rows["x"], rows["y"] = df["POSITION_X"], df["POSITION_Y"]
    rows["w"], rows["h"], rows["angle"] = df["WIDTH"], df["HEIGHT"], df["DEGREES_ANGLE"]
    rows["disappeared"] = ((df["WIDTH"] == 0) & (df["HEIGHT"] == 0) & (df["DEGREES_ANGLE"] == 0)).astype(int)
    loose, worst = compare_rows(rows, ref_rows)
    print(f"{name}: {len(clip)} frames, {len(rows)} rows equal to the oracle's ({loose} rows of recently lost tracks, "
          f"worst {worst:.3g} px); device path {t1 - t0:.2f} s, oracle {t2 - t1:.1f} s", flush=True)

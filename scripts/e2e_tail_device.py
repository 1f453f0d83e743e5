"""The tail of track_bacteria on a 1920-frame table, piece by piece: the device formatter (ysmr_rows_format_device), its copies,
the file, the DataFrame -- against the host path (ysmr_rows_write_csv_columns)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ysmr_amd import _lib
from ysmr_amd import helper_file as hf
n = 971507
rng = np.random.default_rng(0)
rows = np.zeros(n, _lib.ROW_DTYPE)
rows["track_id"] = np.arange(n) // 1500; rows["frame"] = np.arange(n) % 1500
rows["x"] = rng.uniform(0, 1228, n); rows["y"] = rng.uniform(0, 922, n)
rows["w"] = rng.uniform(2, 9, n).astype(np.float32); rows["h"] = rng.uniform(2, 9, n).astype(np.float32); rows["angle"] = rng.uniform(0, 90, n).astype(np.float32)
dev = torch.from_numpy(rows.view(np.uint8).copy()).cuda()
d = tempfile.mkdtemp(dir="/tmp")
for rep in range(4):
    t0 = time.perf_counter(); hf.rows_to_csv_file_and_dataframe(rows, os.path.join(d, "h.csv")); t1 = time.perf_counter()
    made = hf.rows_device_to_csv_file_and_dataframe(dev, n, os.path.join(d, "d.csv")); t2 = time.perf_counter()
    print(f"host path {1e3*(t1-t0):6.1f} ms   device path {1e3*(t2-t1):6.1f} ms   marks {getattr(hf, 'LAST_DEVICE_ROWS_MARKS', None)}", flush=True)
assert open(os.path.join(d, "h.csv"), "rb").read() == open(os.path.join(d, "d.csv"), "rb").read()

"""Time ysmr_evaluate_tracks (the statistics half of evaluate_tracks) on a selected table of the bench clip's size,
device-resident columns, next to the pandas / SciPy oracle on a bounded sample of the same table.
usage: python tests/tools/bench_evaluate.py [--tracks 400] [--max-len 700] [--reps 20] [--cpu-tracks 60]"""
import argparse
import ctypes
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=400)
    ap.add_argument("--max-len", type=int, default=700)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-tracks", type=int, default=60)
    args = ap.parse_args()
    import torch
    from select_tables import make_table, select_settings
    from ysmr_amd import _lib
    from ysmr_amd.evaluate import evaluate_params
    df = make_table(11, n_tracks=args.tracks, height=922, width=1228, max_len=args.max_len)
    size = df.groupby("TRACK_ID")["TRACK_ID"].transform("size")
    df = df[size >= 32].reset_index(drop=True)              # what select_tracks lets through (>= 1 s)
    s = select_settings()
    p = evaluate_params(s, 30.0)
    n = len(df)
    L = _lib.lib()
    dev = torch.device("cuda:0")
    cols = [torch.from_numpy(df[c].to_numpy().astype(np.uint32).view(np.int32)).to(dev) for c in ("TRACK_ID", "POSITION_T")] + \
           [torch.from_numpy(df[c].to_numpy().astype(np.float64)).to(dev) for c in ("POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT")]
    ws = torch.empty(L.ysmr_evaluate_workspace_bytes(n), dtype=torch.uint8, device=dev)
    f64 = lambda: torch.empty(n, dtype=torch.float64, device=dev)   # noqa: E731
    i8 = lambda: torch.empty(n, dtype=torch.int8, device=dev)       # noqa: E731
    outs = [f64(), f64(), torch.empty(n, dtype=torch.int32, device=dev), i8(), i8(), f64(), f64(), i8()]
    stats = torch.empty(n, 12, dtype=torch.float64, device=dev)
    n_tracks = ctypes.c_longlong(0)

    def run():
        rc = L.ysmr_evaluate_tracks(_lib.stream_ptr(), n, *[c.data_ptr() for c in cols], ctypes.byref(p), ws.data_ptr(), ws.numel(),
                                    *[o.data_ptr() for o in outs], stats.data_ptr(), ctypes.byref(n_tracks))
        _lib.check(rc, "ysmr_evaluate_tracks")
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / args.reps
    out = {"rows": n, "tracks": int(n_tracks.value), "gpu_ms_per_call": gpu_s * 1e3, "gpu_rows_per_s": n / gpu_s}
    if args.cpu_tracks > 0:
        from oracle import ysmr_oracle as yo
        ids = df["TRACK_ID"].unique()[: args.cpu_tracks]
        sample = df[df["TRACK_ID"].isin(ids)].reset_index(drop=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            yo.evaluate_tracks_oracle(sample, s, 30.0)
            cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sample) / cpu_s, "unit": "rows/s", "cores": 1, "kind": "port",
                               "sample": f"{len(ids)} tracks ({len(sample)} rows), pandas / SciPy oracle"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

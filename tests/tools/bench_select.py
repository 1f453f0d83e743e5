"""Time ysmr_select_tracks on a table of the bench clip's size (device-resident columns, the call is
synchronous) next to the CPU oracle on a bounded sample of the same table.
usage: python tests/tools/bench_select.py [--tracks 900] [--max-len 600] [--reps 20] [--cpu-tracks 60]"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=900)
    ap.add_argument("--max-len", type=int, default=600)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-tracks", type=int, default=60)
    args = ap.parse_args()
    import torch
    from select_tables import make_table, select_settings
    from ysmr_amd import _lib
    from ysmr_amd.select import select_params
    df = make_table(11, n_tracks=args.tracks, height=922, width=1228, max_len=args.max_len)
    s = select_settings()
    p = select_params(s, 30.0, 922, 1228)
    n = len(df)
    L = _lib.lib()
    dev = torch.device("cuda:0")
    cols = [torch.from_numpy(df[c].to_numpy().view(np.int32)).to(dev) for c in ("TRACK_ID", "POSITION_T")] + \
           [torch.from_numpy(df[c].to_numpy()).to(dev) for c in ("POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT")]
    ws = torch.empty(L.ysmr_select_workspace_bytes(n, p.max_recursion), dtype=torch.uint8, device=dev)
    sel_row, sel_index = torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev)
    summ = _lib.SelectSummary()

    def run():
        rc = L.ysmr_select_tracks(_lib.stream_ptr(), n, *[c.data_ptr() for c in cols], ctypes.byref(p), ws.data_ptr(),
                                  ws.numel(), sel_row.data_ptr(), sel_index.data_ptr(), ctypes.byref(summ))
        _lib.check(rc, "ysmr_select_tracks")
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    torch.cuda.synchronize()
    gpu_s = (time.perf_counter() - t0) / args.reps
    out = {"rows": n, "tracks": int(summ.tracks_before), "rows_after_cleanup": int(summ.rows_after),
           "good_tracks": int(summ.good_tracks), "rows_selected": int(summ.rows_selected),
           "gpu_ms_per_call": gpu_s * 1e3, "gpu_rows_per_s": n / gpu_s}
    if args.cpu_tracks > 0:
        from oracle import ysmr_oracle as yo
        sample = df[df["TRACK_ID"] < args.cpu_tracks].reset_index(drop=True)
        t0 = time.perf_counter()
        yo.select_tracks_oracle(sample, s, 30.0, 922, 1228)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sample) / cpu_s, "unit": "rows/s", "cores": 1, "kind": "port",
                               "sample": f"the first {args.cpu_tracks} tracks ({len(sample)} rows), pandas oracle"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

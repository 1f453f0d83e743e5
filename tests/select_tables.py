"""Synthetic (TRACK_ID, POSITION_T)-ordered tables for the select_tracks tests: tracks with holes,
jumps, lost ('disappeared') rows, odd shapes and sizes, positions near and beyond the frame."""
import numpy as np
import pandas as pd


def make_table(seed, n_tracks=60, height=400, width=600, max_len=260):
    rng = np.random.default_rng(seed)
    parts = []
    for tid in range(n_tracks):
        n = int(rng.integers(5, max_len))
        t0 = int(rng.integers(0, 200))
        frames = t0 + np.arange(n)
        keep = rng.random(n) > rng.choice([0.0, 0.0, 0.0, 0.02, 0.15])          # missing frames
        if rng.random() < 0.3 and n > 40:                             # one long gap
            g = int(rng.integers(10, n - 20))
            keep[g:g + int(rng.integers(3, 12))] = False
        keep[0] = True
        frames = frames[keep]
        n = len(frames)
        kind = rng.choice(["rod", "rod", "rod", "rod", "big", "round", "edge", "outside", "excursion"])
        cx, cy = rng.uniform(0.1, 0.9) * width, rng.uniform(0.1, 0.9) * height
        if kind == "edge":
            cx = rng.uniform(0.0, 0.05) * width
        if kind == "outside":
            cx = -3.0
        step = rng.normal(0, rng.choice([0.3, 1.0, 2.0]), (n, 2))
        jumps = rng.random(n) < rng.choice([0.0, 0.0, 0.01, 0.05])
        step[jumps] += rng.normal(0, 40, (int(jumps.sum()), 2))
        xy = np.cumsum(step, axis=0) + (cx, cy)
        if kind == "excursion":                                       # mean inside the frame, one row outside
            xy[:, 0] = np.clip(xy[:, 0], 0.06 * width, None)
            xy[n // 2, 0] = -0.5
        w = np.float32(rng.normal(6.0, 0.6, n)).astype(np.float64)
        h = np.float32(rng.normal(2.0, 0.25, n)).astype(np.float64)
        if kind == "big":
            w *= 4
            h *= 3
        if kind == "round":
            w = np.float32(rng.normal(4.2, 0.3, n)).astype(np.float64)
            h = w * np.float32(0.9)
        swap = rng.random(n) < 0.5
        w, h = np.where(swap, h, w), np.where(swap, w, h)
        lost = rng.random(n) < rng.choice([0.0, 0.0, 0.03, 0.2])
        w[lost] = 0.0
        h[lost] = 0.0
        blow = rng.random(n) < 0.02
        w[blow] *= 3.0
        deg = np.float32(rng.uniform(-90, 0, n)).astype(np.float64)
        deg[lost] = 0.0
        parts.append(pd.DataFrame({"TRACK_ID": np.full(n, tid, np.uint32), "POSITION_T": frames.astype(np.uint32),
                                   "POSITION_X": xy[:, 0], "POSITION_Y": xy[:, 1], "WIDTH": w, "HEIGHT": h,
                                   "DEGREES_ANGLE": deg}))
    return pd.concat(parts, ignore_index=True)


def select_settings(**kw):
    from ysmr_amd.helper_file import default_settings
    s = default_settings(**{"user input": False, "select files": False, "display video analysis": False,
                            "log to file": False, "minimal length in seconds": 1.0,
                            "limit track length to x seconds": 3.0, "store processed .csv file": False})
    s.update(kw)
    return s

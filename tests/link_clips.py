"""Detection sequences for the batch link's tests at the states where its code paths change:
all twelve waves of k_batch seated (more than 704 live tracks), frames of more than 512 / 600 detections (the wave search
without its float pre-pass, 48 cells per side).  Pure numpy: the CPU suite checks with the oracle that a clip reaches the
state its GPU test is about (tests/test_oracle_link.py), the GPU suite runs it through ``ysmr_tracker_run``."""
import numpy as np


def _info(rng, n):
    return np.column_stack([rng.uniform(1, 9, n), rng.uniform(1, 9, n), rng.uniform(0, 90, n)])


def _f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


def crowded_clip(n_frames=112, n_blobs=744, seed=21):
    """~720-765 live tracks for the whole clip (capacity 768: every seat of the twelve waves in use), with deaths
    (blobs that leave for good; max_disappeared = 5), births in that state (a frame only registers tracks when it holds
    more detections than there are tracks: frames without dropout that bring new blobs and spurious detections) and
    the usual dropout in between.  Returns [(xy (m,2), info (m,3)) per frame]."""
    rng = np.random.default_rng(seed)
    pos = np.column_stack([rng.uniform(10, 1218, n_blobs), rng.uniform(10, 912, n_blobs)])
    vel = rng.normal(0, 0.8, (n_blobs, 2))
    alive = np.ones(n_blobs, bool)
    frames = []
    for f in range(n_frames):
        pos = pos + vel + rng.normal(0, 0.15, pos.shape)
        if f in (18, 40, 47, 66, 90):                   # twelve blobs leave for good: their tracks die six frames later
            gone = rng.choice(np.nonzero(alive)[0], 12, replace=False)
            alive[gone] = False
        full = f in (0, 30, 31, 60, 61, 80, 81, 104)    # frames that show everything, and something new
        keep = alive & (np.ones(len(pos), bool) if full else rng.random(len(pos)) > 0.03)
        xy = pos[keep]
        if full and f:
            fresh = np.column_stack([rng.uniform(10, 1218, 8), rng.uniform(10, 912, 8)])
            pos = np.vstack([pos, fresh]); vel = np.vstack([vel, rng.normal(0, 0.8, (8, 2))])
            alive = np.concatenate([alive, np.ones(8, bool)])
            xy = np.vstack([xy, fresh, np.column_stack([rng.uniform(1300, 1500, 3), rng.uniform(0, 900, 3)])])
        xy = _f32(xy)
        frames.append((xy, _f32(_info(rng, len(xy)))))
    return frames


def dense_detection_clip(n_frames=40, n_blobs=760, seed=22, stationary=False):
    """~760 tracks and 620-760 detections per frame: more than 512 (bl_search_wave scans all of them exactly, without
    its float pre-pass) and more than 600 (48 cells per side, other LDS offsets).  One blob far from everything is seen
    in the first frame only: its lost track asks the wave search in every frame of its life.  ``stationary``: blobs that
    do not move and GSFF off, so that the planted configurations are EXACT ties -- two detections equidistant from a
    track whose own blob is missing (lowest column wins), two tracks equidistant from one detection (lowest id wins)."""
    rng = np.random.default_rng(seed)
    pos = np.column_stack([rng.uniform(40, 1200, n_blobs), rng.uniform(40, 900, n_blobs)])
    if stationary:
        pos = np.rint(pos)
        pos[0] = (300.0, 300.0)                       # track 0: its blob goes missing, (303, 304) and (297, 304) appear
        pos[1] = (500.0, 500.0); pos[2] = (506.0, 508.0)      # tracks 1, 2: both missing, one detection at (503, 504)
        far = np.linalg.norm(pos[3:, None, :] - pos[None, :3, :], axis=2).min(1) > 40
        pos = np.vstack([pos[:3], pos[3:][far]])
    n = len(pos)
    vel = np.zeros((n, 2)) if stationary else rng.normal(0, 0.7, (n, 2))
    frames = []
    for f in range(n_frames):
        if not stationary:
            pos = pos + vel + rng.normal(0, 0.15, pos.shape)
        p_drop = 0.0 if f == 0 else (0.02, 0.08, 0.15)[f % 3]
        keep = rng.random(n) >= p_drop
        extra = np.zeros((0, 2))
        if stationary and f in (5, 6, 20):
            keep[:3] = False
            extra = np.array([(303.0, 304.0), (297.0, 304.0), (503.0, 504.0)])
        xy = np.vstack([pos[keep], extra])
        if f == 0:
            xy = np.vstack([xy, [(5000.0, 4000.0)]])   # seen once, then lost far away from every detection
        xy = _f32(xy)
        frames.append((xy, _f32(_info(rng, len(xy)))))
    return frames


def oracle_rows(oracle, frames, **kw):
    """The clip through OracleTracker: rows [(frame, id, x, y, w, h, deg[, sens])], live tracks per frame, the tracker."""
    ot = oracle.OracleTracker(**kw)
    rows, live = [], []
    for f, (d, info) in enumerate(frames):
        rects = oracle.det_to_rects(np.column_stack([d, info]).astype(np.float32)) if len(d) else []
        ids, xy, inf, _ = ot.update(rects)
        live.append(len(ids))
        for i, tid in enumerate(ids):
            row = (f, tid, float(xy[i][0]), float(xy[i][1]), *map(float, inf[i]))
            rows.append(row + (float(ot.last_sens[i]),) if ot.shadow_gsff else row)
    return rows, np.array(live), ot

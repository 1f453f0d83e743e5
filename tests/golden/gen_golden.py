#!/usr/bin/env python3
"""Generate the committed golden fixtures in tests/golden/ (run in the BUILD container only).

What is pinned, and by what:
  * gsff_*.npz, tracker_*.npz -- outputs of the REFERENCE's own ``ysmr/gsff.py`` and
    ``ysmr/tracker.py`` (imported from /root/reference without executing the package __init__,
    which needs cv2).  Only inputs and outputs are stored -- no reference source.
  * propagation.npz           -- outputs of ``scipy.ndimage.binary_propagation`` (the function the
    reference calls at ysmr/track_eval.py:211-214) on small marker/mask pairs.
The image-side cv2 stages (a1/a2/a3/a5/a6) cannot be pinned: cv2 is not installed and the
reference has no test vectors ("parity unpinned", see oracle/ysmr_oracle.c).

/root/reference does not exist on the GPU box: tests read only the .npz files written here.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ysmr"


def import_reference():
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("ysmr")
    pkg.__path__ = [REF]
    sys.modules["ysmr"] = pkg
    import ysmr.gsff  # noqa: E402
    import ysmr.tracker  # noqa: E402
    return sys.modules["ysmr.gsff"], sys.modules["ysmr.tracker"]


# ------------------------------------------------------------------------------------------------
def gsff_streams(rng):
    """Measurement streams: smooth run, jump + recovery, and a 'lost track' that is fed its own
    predictions (what tracker.py:219-225 does for unmatched tracks)."""
    n = 90
    t = np.arange(n)
    smooth = np.stack([100 + 1.7 * t + rng.normal(0, 0.3, n), 50 - 0.9 * t + rng.normal(0, 0.3, n)], 1)
    jump = smooth.copy()
    jump[40:] += np.array([35.0, -20.0])
    turn = np.stack([300 + 12 * np.cos(t / 7.0), 200 + 12 * np.sin(t / 7.0)], 1) + rng.normal(0, 0.2, (n, 2))
    return {"smooth": smooth, "jump": jump, "turn": turn}


def run_gsff(gsff_mod, stream, fps, n_min, n_max, n_f, lost_from=None):
    g = gsff_mod.GaussianSumFIR(delta_t=1 / fps, n_min=n_min, n_max=n_max if n_max else fps, n_f=n_f,
                                likelihood_minimum=10 ** -20, inv_cov=np.linalg.inv(np.eye(2)),
                                x_hat_array_length=2)
    state = {}
    corr, pred, modes, weights, liks = [], [], [], [], []
    fed = []
    z = None
    for k in range(len(stream)):
        if lost_from is not None and k >= lost_from:
            z = pred[-1].copy()
        else:
            z = np.array(stream[k], dtype=float)
        fed.append(z.copy())
        c, state = g.correct(measurement=z, **state)
        p, state = g.predict(**state)
        corr.append(np.array(c))
        pred.append(np.array(p))
        modes.append(state["mode"])
        w = np.zeros(n_f)
        w[: state["mode"]] = state["weight_array"]
        weights.append(w)
        lk = np.zeros(n_f)
        lk[: state["mode"]] = state["likelihood_array"]
        liks.append(lk)
    gains = {f"gain{i}": gn for i, gn in enumerate(g.gains)}
    return dict(fed=np.array(fed), correct=np.array(corr), predict=np.array(pred), mode=np.array(modes),
                weights=np.array(weights), likelihoods=np.array(liks), n_i=np.array(g.n_i), **gains)


# ------------------------------------------------------------------------------------------------
def tracker_scenario(rng, n_obj, n_frames, area, dropout=0.03, birth=0.02, speckle=1, gap=None, grid=True):
    """Random-walk detections with dropout, spurious detections and births; detections are
    shuffled every frame (the tracker must not depend on detection order beyond ids)."""
    pos = rng.uniform(10, area - 10, (n_obj, 2))
    vel = rng.normal(0, 1.5, (n_obj, 2))
    frames = []
    for f in range(n_frames):
        vel += rng.normal(0, 0.3, vel.shape)
        pos += vel
        if rng.random() < birth * 10:
            pos = np.vstack([pos, rng.uniform(10, area - 10, (1, 2))])
            vel = np.vstack([vel, rng.normal(0, 1.5, (1, 2))])
        keep = rng.random(len(pos)) >= dropout
        det = pos[keep] + rng.normal(0, 0.2, (int(keep.sum()), 2))
        if speckle:
            det = np.vstack([det, rng.uniform(0, area, (rng.integers(0, speckle + 1), 2))])
        if gap is not None and gap[0] <= f < gap[1]:
            det = det[:0]
        order = rng.permutation(len(det))
        det = det[order]
        if grid:
            det = np.round(det * 2) / 2  # half-pixel grid like minAreaRect centres of small blobs
        info = np.stack([rng.uniform(1, 7, len(det)), rng.uniform(1, 7, len(det)),
                         rng.uniform(-90, 0, len(det))], 1).astype(np.float32).astype(np.float64)
        frames.append((det, info))
    return frames


def run_tracker(tracker_mod, frames, fps, use_gsff, n_min=0, n_max=30, n_f=3, max_disappeared=None):
    ct = tracker_mod.CentroidTracker(max_disappeared=fps if max_disappeared is None else max_disappeared,
                                     fps=fps, n_min=n_min, n_max=n_max, n_f=n_f, use_gsff=use_gsff)
    out = {}
    det_all, info_all, det_off = [], [], [0]
    ids_all, xy_all, info_out, gone_all, off = [], [], [], [], [0]
    claims_all, claim_off, next_id = [], [0], []
    for f, (det, info) in enumerate(frames):
        rects = [((float(d[0]), float(d[1])), (float(i[0]), float(i[1]), float(i[2]))) for d, i in zip(det, info)]
        col_of = {id(r[1]): c for c, r in enumerate(rects)}
        before = list(ct.objects.keys())
        if before and len(det):
            # the reference's argsort is not stable: a fixture is only well defined when no two
            # tracks tie (same nearest detection, same distance).  Refuse to write one that is not.
            from scipy.spatial.distance import cdist
            dm = cdist(np.array(list(ct.objects.values())), det.reshape(-1, 2))
            key = np.stack([dm.argmin(1), dm.min(1)], 1)
            assert len(np.unique(key, axis=0)) == len(key), f"tie in frame {f}: fixture would be ill-defined"
        objs, infos = ct.update(rects)
        ids = list(objs.keys())
        # claims: tracks that existed before and now carry one of this frame's info tuples
        claims = []
        for row, tid in enumerate(before):
            if tid in infos and id(infos[tid]) in col_of:
                claims.append((row, col_of[id(infos[tid])]))
        det_all.append(det.reshape(-1, 2)); info_all.append(info.reshape(-1, 3)); det_off.append(det_off[-1] + len(det))
        ids_all.append(np.array(ids, dtype=np.int64))
        xy_all.append(np.array([objs[i] for i in ids], dtype=float).reshape(-1, 2))
        info_out.append(np.array([list(infos[i]) for i in ids], dtype=float).reshape(-1, 3))
        gone_all.append(np.array([ct.disappeared[i] for i in ids], dtype=np.int64))
        off.append(off[-1] + len(ids))
        claims_all.append(np.array(claims, dtype=np.int64).reshape(-1, 2)); claim_off.append(claim_off[-1] + len(claims))
        next_id.append(ct.nextObjectID)
    out.update(det=np.concatenate(det_all), det_info=np.concatenate(info_all), det_off=np.array(det_off),
               ids=np.concatenate(ids_all), xy=np.concatenate(xy_all), info=np.concatenate(info_out),
               disappeared=np.concatenate(gone_all), off=np.array(off),
               claims=np.concatenate(claims_all), claim_off=np.array(claim_off), next_id=np.array(next_id),
               fps=np.float64(fps), use_gsff=np.bool_(use_gsff), n_min=np.int64(n_min),
               n_max=np.int64(-1 if n_max is None else n_max), n_f=np.int64(n_f))
    return out


# ------------------------------------------------------------------------------------------------
def propagation_cases(rng):
    from scipy.ndimage import binary_propagation
    cases = {}
    # diagonal-only contact must NOT propagate (4-connectivity)
    mask = np.zeros((8, 8), np.uint8); mask[1:3, 1:3] = 255; mask[3:5, 3:5] = 255; mask[6, 0:3] = 255
    mark = np.zeros_like(mask); mark[1, 1] = 255
    cases["diag"] = (mark, mask)
    # markers outside the mask are kept and seed their 4-neighbours (dark-on-bright quirk)
    mask = np.zeros((9, 9), np.uint8); mask[4, 2:7] = 255; mask[0, 0] = 255
    mark = np.zeros_like(mask); mark[3, 4] = 255; mark[8, 8] = 255
    cases["outside"] = (mark, mask)
    for i in range(4):
        mask = (rng.random((40, 56)) < 0.45).astype(np.uint8) * 255
        mark = ((rng.random((40, 56)) < 0.03) & (mask > 0)).astype(np.uint8) * 255
        cases[f"rand{i}"] = (mark, mask)
    mask = (rng.random((33, 47)) < 0.3).astype(np.uint8) * 255
    mark = np.maximum(mask, (rng.random((33, 47)) < 0.2).astype(np.uint8) * 255)  # superset (INV)
    cases["superset"] = (mark, mask)
    out = {}
    for k, (mark, mask) in cases.items():
        out[f"{k}_markers"] = mark
        out[f"{k}_mask"] = mask
        out[f"{k}_out"] = binary_propagation(mark, mask=mask).astype(np.uint8) * 255
    return out


def main():
    gsff_mod, tracker_mod = import_reference()
    rng = np.random.default_rng(20241223)
    streams = gsff_streams(rng)
    for name, s in streams.items():
        np.savez_compressed(os.path.join(HERE, f"gsff_{name}_default.npz"),
                            **run_gsff(gsff_mod, s, 30.0, 0, 30, 3))
    np.savez_compressed(os.path.join(HERE, "gsff_smooth_2997.npz"),
                        **run_gsff(gsff_mod, streams["smooth"], 29.97, 0, None, 3))
    np.savez_compressed(os.path.join(HERE, "gsff_lost_default.npz"),
                        **run_gsff(gsff_mod, streams["turn"], 30.0, 0, 30, 3, lost_from=45))
    np.savez_compressed(os.path.join(HERE, "gsff_jump_nf4.npz"),
                        **run_gsff(gsff_mod, streams["jump"], 25.0, 4, 24, 4))

    sc = tracker_scenario(rng, 40, 80, 400.0)
    np.savez_compressed(os.path.join(HERE, "tracker_small_gsff.npz"), **run_tracker(tracker_mod, sc, 30.0, True))
    sc_free = tracker_scenario(rng, 40, 80, 400.0, grid=False)  # GSFF off + grid => exact ties
    np.savez_compressed(os.path.join(HERE, "tracker_small_nogsff.npz"), **run_tracker(tracker_mod, sc_free, 30.0, False))
    sc = tracker_scenario(rng, 250, 50, 1200.0)
    np.savez_compressed(os.path.join(HERE, "tracker_mid_gsff.npz"), **run_tracker(tracker_mod, sc, 30.0, True))
    # empty-detection gap longer than max_disappeared -> every track is deregistered, ids restart
    sc = tracker_scenario(rng, 12, 40, 200.0, gap=(15, 24))
    np.savez_compressed(os.path.join(HERE, "tracker_gap.npz"),
                        **run_tracker(tracker_mod, sc, 30.0, True, max_disappeared=5))
    sc = tracker_scenario(rng, 30, 45, 300.0, dropout=0.15)
    np.savez_compressed(os.path.join(HERE, "tracker_2997.npz"),
                        **run_tracker(tracker_mod, sc, 29.97, True, n_max=None, max_disappeared=6))
    np.savez_compressed(os.path.join(HERE, "propagation.npz"), **propagation_cases(rng))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()

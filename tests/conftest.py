import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
#: the round whose name the parity records of this run carry (gpurun_out/<ROUND>_parity_*.json[l], copied to profiles/)
ROUND = "r05"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def tracker_frames(g):
    """Yield (det (M,2), info (M,3)) per frame from a tracker_*.npz fixture."""
    off = g["det_off"]
    for f in range(len(off) - 1):
        yield g["det"][off[f]:off[f + 1]], g["det_info"][off[f]:off[f + 1]]


def rects_of(det, info):
    return [((float(d[0]), float(d[1])), (float(i[0]), float(i[1]), float(i[2]))) for d, i in zip(det, info)]


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """libysmr_hip.so is not tracked: build it (hipcc cross-compiles without a GPU) so that any single test
    file can be run on its own, and re-run make every session so that an edit to csrc/*.hip is never tested
    against a stale binary (a no-op when the library is current; a box without hipcc keeps the prebuilt one)."""
    import shutil
    import subprocess
    from ysmr_amd import _lib
    here = os.path.dirname(_lib.LIB_PATH)
    if not _lib.LIB_PATH.startswith(ROOT):
        return                                    # YSMR_HIP_LIB points at a tuning build
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    elif shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.run(["make", "-s", "-j8", "-C", here], check=True)


@pytest.fixture(scope="session")
def oracle():
    from oracle import ysmr_oracle
    ysmr_oracle.build()
    return ysmr_oracle


#: How far an ill-conditioned row may move, as a multiple of what the oracle's own one-ulp shadows moved it
#: (compare_rows).  Set from measurement, not from taste: 4 x the largest ratio deviation / sens that any test of the
#: suite produced on the final code of the round (every comparison that meets marked rows appends its ratio to
#: gpurun_out/<ROUND>_parity_ratios.jsonl; profiles/r04_parity_ratios.jsonl, profiles/r05_parity_ratios.jsonl are the rounds'
#: copies): 2.4 on a 48-frame dark-on-bright clip (its marked rows sit in the exponential phase of a weight tie, where a rounding
#: difference of a few ulps -- the batch link's window sums against the oracle's dot products -- leads the one-ulp shadows by
#: a frame or two), 0.53 at the bench configuration.  Round 3 allowed 1000 x and 50 px without recording how much was used.
AMPLIFICATION = 10.0
#: ... and in pixels: 4 x the largest deviation of a marked row seen in the suite (1.3 px, same test)
ABS_LIMIT_PX = 5.0


def parity_report(got, ref_rows):
    """What compare_rows checks, as numbers (rows must already agree in frame / id): per table the count of rows the
    oracle's shadow filters mark ill-conditioned, how far the device is from the oracle on them relative to the
    shadows' own spread, and how many rows lie beyond north_star's 1e-5 relative -- split into those whose shadows
    (in this row or earlier in the same lost phase of the track) are themselves more than 1e-5 apart, i.e. rows that
    the reference's arithmetic does not determine to 1e-5, and the rest (must be none)."""
    from oracle.ysmr_oracle import OracleTracker
    ref = np.array(ref_rows, dtype=float)
    sens = ref[:, 7] if ref.shape[1] > 7 else np.zeros(len(ref))
    tied = np.isinf(sens)
    dev = np.maximum(np.abs(got["x"] - ref[:, 2]) / np.maximum(1, np.abs(ref[:, 2])),
                     np.abs(got["y"] - ref[:, 3]) / np.maximum(1, np.abs(ref[:, 3])))
    dev_px = np.maximum(np.abs(got["x"] - ref[:, 2]), np.abs(got["y"] - ref[:, 3]))
    well = ~tied & (sens <= OracleTracker.ILL_CONDITIONED)
    loose = ~tied & ~well
    ratio = dev[loose] / sens[loose] if loose.any() else np.zeros(0)
    beyond = loose & (dev > 1e-5)
    # running maximum of sens along each track (rows are frame-major: group by id, keep frame order)
    order = np.lexsort((ref[:, 0], ref[:, 1]))
    run = np.zeros(len(ref))
    last_id, cur = None, 0.0
    for k in order:
        tid = ref[k, 1]
        if tid != last_id or sens[k] <= OracleTracker.ILL_CONDITIONED:
            cur = 0.0                 # a new track, or the track is well determined again (found its blob)
        last_id = tid
        if np.isfinite(sens[k]):
            cur = max(cur, sens[k])
        run[k] = cur
    return {"rows": int(len(got)), "rows_of_lost_tracks": int((got["disappeared"] > 0).sum()),
            "ill_conditioned_rows": int(loose.sum()), "ill_conditioned_fraction": float(loose.sum() / max(1, len(got))),
            "worst_ill_conditioned_px": float(dev_px[loose].max()) if loose.any() else 0.0,
            "worst_ill_conditioned_relative": float(dev[loose].max()) if loose.any() else 0.0,
            "worst_deviation_over_sens": float(ratio.max()) if len(ratio) else 0.0,
            "median_deviation_over_sens": float(np.median(ratio)) if len(ratio) else 0.0,
            "ill_conditioned_rows_beyond_1e-5_relative": int(beyond.sum()),
            "of_which_shadows_beyond_1e-5_in_this_row": int((beyond & (sens > 1e-5)).sum()),
            "of_which_shadows_beyond_1e-5_in_this_lost_phase": int((beyond & (run > 1e-5)).sum()),
            "beyond_1e-5_with_shadows_within_1e-5": int((beyond & ~(run > 1e-5)).sum()),
            "largest_shadow_spread_relative": float(sens[loose].max()) if loose.any() else 0.0,
            "rows_of_tie_assigned_tracks": int(tied.sum()),
            "worst_well_conditioned_relative": float(dev[well].max()) if well.any() else 0.0,
            "amplification_allowed": AMPLIFICATION}


def _note_ratio(key, ratio, px, n):
    """One line per comparison that met ill-conditioned rows -> gpurun_out/<ROUND>_parity_ratios.jsonl: how much of the
    allowed amplification each test uses (the constant above is set from the maximum over the whole suite)."""
    import json
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, ROUND + "_parity_ratios.jsonl"), "a") as fh:
            fh.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", "?"), "coordinate": key,
                                 "worst_deviation_over_sens": ratio, "worst_px": px, "rows": n}) + "\n")
    except OSError:
        pass


def compare_rows(got, ref_rows, amplification=None):
    """Device rows (structured ysmr_row array) vs oracle rows [(frame, id, x, y, w, h, deg[, sens]), ...].

    Integers exact; w, h bit-exact; angle within one f32 ulp (device atan2).  Positions: 1e-9 for every
    row that the reference's own arithmetic determines (north_star asks for 1e-5 relative).

    Which rows those are is decided by the ORACLE, not by a heuristic: run with ``shadows=2`` it carries,
    beside every track's Gaussian-sum filter, two shadow filters whose FIR outputs and likelihoods are
    moved by one ulp (oracle/ysmr_oracle.py: OracleTracker).  ``sens`` = how far that moves the row.
    A 'disappeared' track is fed its own blended prediction (tracker.py:219-225); when that happens
    while its filter weights are tied -- they are reset to exactly uniform whenever another filter
    switches on (gsff.py:291-295) -- and the filters disagree, the state sits on an unstable symmetric
    equilibrium and the replicator update w_i <- lik_i * w_i / sum amplifies the last bit of the
    likelihoods frame by frame until one filter wins.  Those rows have sens > ILL_CONDITIONED (1e-12,
    i.e. a one-ulp change is amplified >= 10^4 times); they are a fraction of a percent of a table
    (the blanket "lost within 31 frames" window of round 1 covered half of it).  They are held to
    ``AMPLIFICATION`` x their own sensitivity and to < ``ABS_LIMIT_PX``.  Rows without a ``sens`` entry (oracle run
    without shadows, GSFF off) are all held to 1e-9.  sens = inf marks tracks assigned by a distance tie.
    Returns (number of ill-conditioned rows, their worst deviation in px).
    """
    amplification = AMPLIFICATION if amplification is None else amplification
    ref = np.array(ref_rows, dtype=float)
    ref = ref.reshape(-1, ref.shape[1] if ref.ndim == 2 else 7)
    assert len(got) == len(ref), (len(got), len(ref))
    np.testing.assert_array_equal(got["frame"], ref[:, 0].astype(int))
    np.testing.assert_array_equal(got["track_id"], ref[:, 1].astype(int))
    sens = ref[:, 7] if ref.shape[1] > 7 else np.zeros(len(ref))
    from oracle.ysmr_oracle import OracleTracker
    # rows of tracks whose assignment once hung on an exact distance tie (sens = inf): which of two equally
    # distant tracks got a detection is decided by the last bits of the reference's LAPACK-computed gains;
    # they are a handful per table and only their (frame, id) structure is compared
    tied = np.isinf(sens)
    assert tied.sum() <= max(8, 0.002 * len(ref)), f"{int(tied.sum())} rows of tie-assigned tracks"
    firm = ~tied
    loose = firm & (sens > OracleTracker.ILL_CONDITIONED)
    exact = firm & ~loose
    worst = 0.0
    for key, col in (("x", 2), ("y", 3)):
        np.testing.assert_allclose(got[key][exact], ref[exact, col], rtol=1e-9, atol=1e-9)
        if loose.any():
            dev = np.abs(got[key][loose] - ref[loose, col])
            scale = np.maximum(1.0, np.abs(ref[loose, col]))
            bound = np.maximum(1e-9, amplification * sens[loose]) * scale
            assert np.all(dev <= bound), f"{key}: ill-conditioned row off by {(dev / bound).max():.3g} x its bound"
            assert dev.max() < ABS_LIMIT_PX, f"{key}: ill-conditioned row off by {dev.max()} px"
            worst = max(worst, float(dev.max()))
            _note_ratio(key, float((dev / (np.maximum(1e-300, sens[loose]) * scale)).max()), float(dev.max()), int(loose.sum()))
    np.testing.assert_array_equal(got["w"][firm], ref[firm, 4].astype(np.float32))
    np.testing.assert_array_equal(got["h"][firm], ref[firm, 5].astype(np.float32))
    a, b = got["angle"][firm], ref[firm, 6].astype(np.float32)
    assert np.all((a == b) | (np.abs(a - b) <= np.spacing(np.maximum(np.abs(a), np.abs(b))))), "angle"
    return int(loose.sum()), worst

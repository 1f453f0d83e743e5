import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


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
    """A fresh checkout has no libysmr_hip.so (it is not tracked): build it once (hipcc cross-compiles
    without a GPU) so that any single test file can be run on its own."""
    from ysmr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import ysmr_oracle
    ysmr_oracle.build()
    return ysmr_oracle


def compare_rows(got, ref_rows, hist=31):
    """Device rows (structured ysmr_row array) vs oracle rows [(frame, id, x, y, w, h, deg), ...].

    Integers exact.  Positions of tracks with an unbroken detection history: 1e-9.

    A 'disappeared' track is fed its own blended prediction (tracker.py:219-225).  When that happens
    while its filter weights are tied -- they are reset to exactly uniform whenever another filter
    switches on (gsff.py:291-295) -- and the filters disagree, the state sits on an UNSTABLE symmetric
    equilibrium: z is the midpoint of the predictions, the likelihoods are equal up to the last bit,
    and the replicator update w_i <- lik_i * w_i / sum amplifies that last bit by about
    Var_w(x_hat) [px^2] per frame until one filter wins.  Which one wins is decided by rounding noise
    (BLAS summation order, exp), in the reference itself as much as here; the extrapolated position
    then differs by up to the filters' disagreement, i.e. pixels (observed: weights [0.5, 0.5],
    predictions 4 px apart, 12 px deviation after 25 lost frames; tests/tools/debug_inv.py).
    Rows of tracks lost within the last `hist` frames are therefore checked statistically: >= 90 %
    within north_star's 1e-5 relative (1e-3 px absolute near zero), none off by more than 50 px.
    Returns (number of loose rows, worst loose deviation in px).
    """
    ref = np.array(ref_rows, dtype=float).reshape(-1, 7)
    assert len(got) == len(ref), (len(got), len(ref))
    np.testing.assert_array_equal(got["frame"], ref[:, 0].astype(int))
    np.testing.assert_array_equal(got["track_id"], ref[:, 1].astype(int))
    last_lost = {}
    loose = np.zeros(len(got), bool)
    for i in range(len(got)):
        tid, f = int(got["track_id"][i]), int(got["frame"][i])
        if got["disappeared"][i] > 0:
            last_lost[tid] = f
        loose[i] = tid in last_lost and f - last_lost[tid] <= hist
    worst = 0.0
    for key, col in (("x", 2), ("y", 3)):
        np.testing.assert_allclose(got[key][~loose], ref[~loose, col], rtol=1e-9, atol=1e-9)
        if loose.any():
            dev = np.abs(got[key][loose] - ref[loose, col])
            ok = dev <= 1e-3 + 1e-5 * np.abs(ref[loose, col])
            assert ok.mean() >= 0.9, f"{key}: only {ok.mean():.1%} of lost-track rows within 1e-5"
            assert dev.max() < 50.0, f"{key}: lost-track row off by {dev.max()} px"
            worst = max(worst, float(dev.max()))
    np.testing.assert_array_equal(got["w"], ref[:, 4].astype(np.float32))
    np.testing.assert_array_equal(got["h"], ref[:, 5].astype(np.float32))
    a, b = got["angle"], ref[:, 6].astype(np.float32)
    assert np.all((a == b) | (np.abs(a - b) <= np.spacing(np.maximum(np.abs(a), np.abs(b))))), "angle"
    return int(loose.sum()), worst

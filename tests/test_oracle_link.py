"""The link-half oracle (oracle/ysmr_oracle.py) pinned against fixtures captured from the
reference's own ysmr/gsff.py and ysmr/tracker.py (tests/golden/gen_golden.py).

The oracle issues the same NumPy/BLAS calls as the reference, so in one environment it is
bit-identical; across machines BLAS/libm may differ in the last bits, hence rtol=1e-12 on floats
and exact equality on every integer quantity (ids, modes, claims, counters).
"""
import numpy as np
import pytest

from conftest import golden, rects_of, tracker_frames

GSFF_CASES = [("gsff_smooth_default.npz", 30.0, 0, 30, 3), ("gsff_jump_default.npz", 30.0, 0, 30, 3),
              ("gsff_turn_default.npz", 30.0, 0, 30, 3), ("gsff_lost_default.npz", 30.0, 0, 30, 3),
              ("gsff_smooth_2997.npz", 29.97, 0, 29.97, 3), ("gsff_jump_nf4.npz", 25.0, 4, 24, 4)]


@pytest.mark.parametrize("name,fps,n_min,n_max,n_f", GSFF_CASES)
def test_gsff_matches_reference(oracle, name, fps, n_min, n_max, n_f):
    g = golden(name)
    f = oracle.OracleGSFF(delta_t=1 / fps, n_min=n_min, n_max=n_max, n_f=n_f)
    assert f.n_i == list(g["n_i"])
    for i, gain in enumerate(f.gains):
        np.testing.assert_allclose(gain, g[f"gain{i}"], rtol=1e-12, atol=1e-15)
    st = oracle.GsffState()
    for k, z in enumerate(g["fed"]):
        c = f.correct(np.array(z), st)
        p = f.predict(st)
        assert st.mode == g["mode"][k]
        np.testing.assert_allclose(c, g["correct"][k], rtol=1e-12)
        np.testing.assert_allclose(p, g["predict"][k], rtol=1e-12)
        np.testing.assert_allclose(st.weights, g["weights"][k][:st.mode], rtol=1e-10, atol=1e-300)
        np.testing.assert_allclose(st.likelihoods, g["likelihoods"][k][:st.mode], rtol=1e-10, atol=1e-300)


def test_gsff_horizons_and_closed_form(oracle):
    assert oracle.horizon_sizes(0, 30, 3) == [10, 20, 30]
    assert oracle.horizon_sizes(0, 29.97, 3) == [9, 19, 29]
    # rows 0/1 of the gain are decoupled and equal the one-step-ahead LS line-fit predictor
    for n in (9, 10, 20, 30):
        gain = oracle.lsf_gain(n, 1 / 30.0)
        t = np.arange(n) - (n - 1) / 2
        coef = 1 / n + t * ((n + 1) / 2) / np.sum(t * t)
        np.testing.assert_allclose(gain[0, 0::2], coef, atol=1e-12)
        np.testing.assert_allclose(gain[1, 1::2], coef, atol=1e-12)
        assert np.all(gain[0, 1::2] == 0) and np.all(gain[1, 0::2] == 0)


TRACKER_CASES = ["tracker_small_gsff.npz", "tracker_small_nogsff.npz", "tracker_mid_gsff.npz",
                 "tracker_gap.npz", "tracker_2997.npz"]


@pytest.mark.parametrize("name", TRACKER_CASES)
def test_tracker_matches_reference(oracle, name):
    g = golden(name)
    fps = float(g["fps"])
    n_max = None if int(g["n_max"]) < 0 else int(g["n_max"])
    max_gone = {"tracker_gap.npz": 5, "tracker_2997.npz": 6}.get(name, fps)
    tr = oracle.OracleTracker(max_disappeared=max_gone, fps=fps, n_min=int(g["n_min"]), n_max=n_max,
                              n_f=int(g["n_f"]), use_gsff=bool(g["use_gsff"]))
    off, coff = g["off"], g["claim_off"]
    for f, (det, info) in enumerate(tracker_frames(g)):
        ids, xy, inf, claims = tr.update(rects_of(det, info))
        sl = slice(off[f], off[f + 1])
        assert ids == list(g["ids"][sl]), f"frame {f}"
        np.testing.assert_allclose(xy, g["xy"][sl], rtol=1e-12, err_msg=f"frame {f}")
        np.testing.assert_array_equal(np.array([list(i) for i in inf], dtype=float).reshape(-1, 3), g["info"][sl])
        assert [t.gone for t in tr.tracks] == list(g["disappeared"][sl])
        assert tr.next_id == g["next_id"][f]
        assert sorted(claims) == sorted(map(tuple, g["claims"][coff[f]:coff[f + 1]].tolist())), f"frame {f}"


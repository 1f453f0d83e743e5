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



def test_batch_link_clips_reach_the_states_their_gpu_tests_are_about(oracle):
    """tests/link_clips.py builds the detection sequences of the batch link's GPU tests at the states where k_batch's code
    paths change: more than 704 live tracks for >= 70 frames with deaths and births in that state (all
    twelve waves seated), frames of more than 600 detections (48 cells per side, the wave search without its float
    pre-pass).  Checked here, on the CPU, so that an edit of a generator cannot quietly move a test off its path."""
    from link_clips import crowded_clip, dense_detection_clip, oracle_rows
    kw = dict(max_disappeared=5.0, fps=30.0, n_min=0, n_max=30, n_f=3)
    frames = crowded_clip()
    rows, live, ot = oracle_rows(oracle, frames, use_gsff=True, **kw)
    assert len(frames) >= 100 and live.min() >= 716 and live.max() <= 768           # seats 705.. of the twelfth wave in use throughout
    assert ot.next_id - live[-1] >= 60                                               # deaths ...
    births = [f for f in range(1, len(live)) if max(r[1] for r in rows if r[0] == f) > max(r[1] for r in rows if r[0] == f - 1)]
    assert len(births) >= 5 and all(live[f - 1] >= 716 for f in births)             # ... and births in that state
    assert any(32 < f < 96 for f in births)
    for stationary in (False, True):
        frames = dense_detection_clip(stationary=stationary)
        rows, live, ot = oracle_rows(oracle, frames, use_gsff=not stationary, **kw)
        m = np.array([len(d) for d, _ in frames])
        assert live.max() <= 768 and live.min() > 704 and m.min() > 600 and m.max() <= 768

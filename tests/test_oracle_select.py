"""select_tracks oracle (oracle/ysmr_oracle.py:select_tracks_oracle): parity unpinned -- the
reference's track_eval.py cannot be imported here (top-level cv2) and ships no vectors -- so the
restatement is checked for the properties the reference's criteria imply."""
import numpy as np
import pytest

from select_tables import make_table, select_settings


@pytest.mark.parametrize("seed", [0, 1])
def test_selected_fragments_obey_the_criteria(oracle, seed):
    df = make_table(seed)
    s = select_settings()
    out, info = oracle.select_tracks_oracle(df, s, 30.0, 400, 600)
    assert out is not None and info["status"] == 0
    assert list(out.columns) == ["index", "TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT",
                                 "DEGREES_ANGLE"]
    assert sum(info["kick_reasons"]) == info["tracks_after"] and info["kick_reasons"][0] >= info["good_tracks"] > 0
    assert out["index"].is_monotonic_increasing and info["rows_selected"] == len(out)
    for _, frag in out.groupby("TRACK_ID"):
        steps = np.diff(frag["POSITION_T"].to_numpy().astype(np.int64))
        assert (np.diff(frag["index"].to_numpy()) == 1).all()                   # one contiguous fragment per track
        assert len(frag) >= 30 or frag["POSITION_T"].iloc[-1] - frag["POSITION_T"].iloc[0] + 1 <= 90
        assert steps.max(initial=1) <= s["maximal consecutive holes"]
        assert frag["POSITION_T"].iloc[-1] - frag["POSITION_T"].iloc[0] + 1 <= 90      # 3 s limit
        assert (frag["WIDTH"] * frag["HEIGHT"] != 0).all()
        assert frag["POSITION_X"].between(0, 600).all() and frag["POSITION_Y"].between(0, 400).all()
    # every selected row is a row of the input
    merged = out.merge(df, on=["TRACK_ID", "POSITION_T"], suffixes=("", "_in"))
    assert len(merged) == len(out) and (merged["POSITION_X"] == merged["POSITION_X_in"]).all()


def test_statuses_and_switches(oracle):
    df = make_table(2, n_tracks=20)
    s = select_settings()
    assert oracle.select_tracks_oracle(df.iloc[:10], s, 30.0, 400, 600)[1]["status"] == 1
    tiny = df.copy()
    tiny["WIDTH"] = 0.0
    assert oracle.select_tracks_oracle(tiny, s, 30.0, 400, 600)[1]["status"] == 2
    far = df.copy()
    far["POSITION_X"] += 5000.0
    assert oracle.select_tracks_oracle(far, s, 30.0, 400, 600)[1]["status"] == 3
    exact = select_settings(**{"limit track length exactly": True})
    out, _ = oracle.select_tracks_oracle(df, exact, 30.0, 400, 600)
    if out is not None:
        for _, frag in out.groupby("TRACK_ID"):
            assert frag["POSITION_T"].iloc[-1] - frag["POSITION_T"].iloc[0] + 1 == 90
    no_limit = select_settings(**{"limit track length to x seconds": 0.0, "try to omit motility outliers": False,
                                  "percent quantiles excluded area": 0.0, "maximal recursion depth": 0})
    out2, info2 = oracle.select_tracks_oracle(df, no_limit, 30.0, 400, 600)
    assert info2["outliers_used"] == 0 and info2["area_lo"] == -1 and info2["area_hi"] == np.inf

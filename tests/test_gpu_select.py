"""GPU: ysmr_select_tracks (select_tracks / find_good_tracks on the device) vs the CPU oracle.
Selected rows, their index column, every count and every bound must be equal -- the bounds as
doubles, bit for bit (median_linear, numpy's linear percentile and pairwise summation are reproduced)."""
import os

import numpy as np
import pytest

from select_tables import make_table, select_settings

pytestmark = pytest.mark.gpu

_INFO = ("rows_before", "tracks_before", "rows_after", "tracks_after", "area_lo", "area_hi", "q1_dist", "q3_dist",
         "dist_fence", "dist_outliers", "outliers_used", "good_tracks", "rows_selected")


def _compare(oracle, df, settings, fps=30.0, h=400, w=600):
    from ysmr_amd.select import select_params, select_rows
    ref, info = oracle.select_tracks_oracle(df, settings, fps, h, w)
    rows, index, s = select_rows(df, select_params(settings, fps, h, w))
    assert s.status == info["status"]
    if info["status"] in (1, 2):
        return ref, info
    for k in _INFO:
        assert getattr(s, k) == info[k], (k, getattr(s, k), info[k])
    assert list(s.kick_reasons) == info["kick_reasons"]
    if ref is None:
        assert len(rows) == 0
        return ref, info
    np.testing.assert_array_equal(index, ref["index"].to_numpy())
    got = df.iloc[rows]
    for col in ("TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"):
        np.testing.assert_array_equal(got[col].to_numpy(), ref[col].to_numpy(), err_msg=col)
    return ref, info


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_select_matches_oracle(oracle, seed):
    df = make_table(seed)
    variants = [{}, {"try to omit motility outliers": False},
                {"limit track length exactly": True},
                {"limit track length to x seconds": 0.0, "percent quantiles excluded area": 0.0},
                {"maximal recursion depth": 0}, {"maximal recursion depth": 1},
                {"stop excluding motility outliers if total count above percent": 0.001},
                {"exclude measurement when above x times average area": 0.0, "percent of screen edges to exclude": 0.0},
                {"maximal consecutive holes": 1, "maximal empty frames in %": 1.5, "minimal length in seconds": 0.3}]
    for kw in variants:
        ref, info = _compare(oracle, df, select_settings(**kw))
        assert info["status"] in (0, 3)


def test_select_long_tracks_use_the_pairwise_tree(oracle):
    """Tracks of thousands of rows: segments span many 128-value leaf blocks of numpy's pairwise sum."""
    rng = np.random.default_rng(5)
    import pandas as pd
    parts = []
    for tid, n in enumerate([129, 257, 1000, 5000, 12345, 130, 8, 136]):
        xy = np.cumsum(rng.normal(0, 0.7, (n, 2)), axis=0) + (300, 200)
        w = np.float32(rng.normal(6, 0.5, n)).astype(np.float64)
        h = np.float32(rng.normal(2, 0.2, n)).astype(np.float64)
        parts.append(pd.DataFrame({"TRACK_ID": np.full(n, tid, np.uint32), "POSITION_T": np.arange(n, dtype=np.uint32),
                                   "POSITION_X": xy[:, 0], "POSITION_Y": xy[:, 1], "WIDTH": w, "HEIGHT": h,
                                   "DEGREES_ANGLE": np.zeros(n)}))
    df = pd.concat(parts, ignore_index=True)
    s = select_settings(**{"limit track length to x seconds": 0.0, "try to omit motility outliers": False,
                           "percent of screen edges to exclude": 0.0})
    ref, info = _compare(oracle, df, s, h=2000, w=2000)
    assert info["good_tracks"] >= 5


def test_select_statuses(oracle):
    df = make_table(2, n_tracks=20)
    s = select_settings()
    _compare(oracle, df.iloc[:10].reset_index(drop=True), s)
    tiny = df.copy()
    tiny["WIDTH"] = 0.0
    _compare(oracle, tiny, s)
    far = df.copy()
    far["POSITION_X"] += 5000.0
    _, info = _compare(oracle, far, s)
    assert info["status"] == 3


def test_analyse_runs_the_selection(tmp_path, oracle):
    """analyse(): video -> table -> select_tracks -> <name>_selected_data.csv, and the same from the csv."""
    import pandas as pd
    from ysmr_amd import analyse
    from ysmr_amd.helper_file import get_data
    from ysmr_amd.synth import SyntheticVideo
    frames = SyntheticVideo(200, 260, 14, seed=3, dropout=0.01).frames(90)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    from ysmr_amd.main import _EVALUATE_KEYS
    # the selection is the last stage here (with any evaluation key set, analyse() goes on to evaluate_tracks
    # and returns ITS result, as upstream: tests/test_gpu_evaluate.py)
    s = select_settings(**{"minimal frame count": 40, "store processed .csv file": True, **{k: False for k in _EVALUATE_KEYS}})
    out_dir = tmp_path / "res"
    from ysmr_amd.track_eval import track_bacteria
    # what select_tracks is handed in the reference: the DataFrame track_bacteria returns (values parsed
    # ONCE from the tracker's text; re-reading the final csv parses pandas' own output a second time, and
    # pandas' float parser is not an exact inverse of its writer)
    os.makedirs(tmp_path / "first")
    in_memory = track_bacteria(str(path), settings=dict(s), result_folder=str(tmp_path / "first"))[0]
    ref, info = oracle.select_tracks_oracle(in_memory, s, 30.0, 200, 260)
    assert ref is not None and info["good_tracks"] >= 3
    df = analyse(str(path), settings=dict(s), result_folder=str(out_dir), return_df=True)
    assert df is not None
    pd.testing.assert_frame_equal(df, ref, check_dtype=False, check_exact=True)
    assert (out_dir / "clip_selected_data.csv").read_text() == ref.to_csv(index=False)
    # second entry: the list csv of an earlier run (fps and frame size from clip_meta.json)
    ref2, _ = oracle.select_tracks_oracle(get_data(str(out_dir / "clip_list.csv")), s, 30.0, 200, 260)
    df2 = analyse(str(out_dir / "clip_list.csv"), settings=dict(s), result_folder=str(out_dir), return_df=True)
    pd.testing.assert_frame_equal(df2, ref2, check_dtype=False, check_exact=True)
    assert (out_dir / "clip_list_selected_data.csv").read_text() == ref2.to_csv(index=False)


def test_select_tracks_wrapper_messages_and_file(tmp_path, caplog, oracle):
    """The host mirror: the reference's log lines for every exit, and <name>_selected_data.csv."""
    import logging
    from ysmr_amd.select import select_tracks
    caplog.set_level(logging.DEBUG, logger="ysmr")
    df = make_table(1)
    s = select_settings(**{"store processed .csv file": True})
    kw = dict(path_to_file=str(tmp_path / "t_list.csv"), results_directory=str(tmp_path), fps=30.0, frame_height=400,
              frame_width=600)
    out = select_tracks(df=df.copy(), settings=s, **kw)
    ref, _ = oracle.select_tracks_oracle(df, s, 30.0, 400, 600)
    assert out is not None and out.equals(ref)
    assert (tmp_path / "t_list_selected_data.csv").read_text() == ref.to_csv(index=False)
    assert "Tracks before initial cleanup" in caplog.text and "Area quartiles" in caplog.text and "passed:" in caplog.text
    assert "Low amount of accepted tracks" in caplog.text          # a few of ~50 tracks pass
    assert select_tracks(df=df.iloc[:10].reset_index(drop=True), settings=s, **kw) is None
    assert "insufficient length before initial clean-up" in caplog.text
    flat = df.copy(); flat["HEIGHT"] = 0.0
    assert select_tracks(df=flat, settings=s, **kw) is None
    assert "insufficient length after initial clean-up" in caplog.text
    far = df.copy(); far["POSITION_Y"] -= 9000.0
    assert select_tracks(df=far, settings=s, **kw) is None
    assert "has no acceptable tracks" in caplog.text
    noisy = select_settings(**{"stop excluding motility outliers if total count above percent": 0.0001})
    select_tracks(df=df.copy(), settings=noisy, **kw)
    assert "Distance outlier exclusion switched off" in caplog.text

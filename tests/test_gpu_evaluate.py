"""GPU: evaluate_tracks' statistics (csrc/evaluate.hip) against the pandas / SciPy oracle, column by column."""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _long_enough(df, rows):
    """Tracks of at least `rows` rows.  (A track shorter than the second median filter's kernel, about one second,
    makes scipy.signal.medfilt read past its zero padding: SciPy 1.15 returns uninitialised int8 values for it.
    Upstream never meets such tracks: the selection only passes tracks of 'minimal length in seconds'.)"""
    size = df.groupby("TRACK_ID")["TRACK_ID"].transform("size")
    return df[size >= rows].reset_index(drop=True)


def _compare(oracle, df, settings, fps):
    from ysmr_amd.evaluate import ROW_COLUMNS, STATS_COLUMNS, evaluate_columns, evaluate_params
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_rows, ref_stats = oracle.evaluate_tracks_oracle(df, settings, fps)
    rows, stats = evaluate_columns(df.reset_index(drop=True), evaluate_params(settings, fps))
    assert list(ref_rows.columns) == ROW_COLUMNS and list(ref_stats.columns) == STATS_COLUMNS
    assert stats.shape == (len(ref_stats), 12)
    # the heading goes through atan2, which the device library does not round correctly: a change of heading that
    # lies within 1e-9 of a whole number of degrees may truncate to the neighbouring integer (never seen; counted)
    ang_ref = ref_rows["angle_diff"].to_numpy()
    off = rows["angle_diff"] != ang_ref
    assert off.sum() == 0, f"{off.sum()} of {len(off)} angle_diff values differ"
    for name in ("WIDTH", "HEIGHT", "travelled_dist"):
        np.testing.assert_array_equal(rows[name], ref_rows[name].to_numpy(), err_msg=name)          # bit for bit
    for name in ("moving", "turn_points", "motility_phenotype"):
        np.testing.assert_array_equal(rows[name], ref_rows[name].to_numpy().astype(np.int8), err_msg=name)
    np.testing.assert_array_equal(rows["tp_of_tracks"], ref_rows["tp_of_tracks"].to_numpy(), err_msg="tp_of_tracks")   # NaN == NaN
    for k, name in enumerate(STATS_COLUMNS):
        want = ref_stats[name].to_numpy().astype(np.float64)
        np.testing.assert_array_equal(stats[:, k], want, err_msg=name)                                # as doubles, bit for bit
    return ref_rows, ref_stats


def test_statistics_match_the_oracle_bit_for_bit(oracle):
    from select_tables import make_table, select_settings
    seen = set()
    for seed, kw, fps in ((1, {}, 30.0), (2, {"compare angle between n frames": 3, "minimal angle in degrees for turning point": 12.5}, 29.97),
                          (3, {"minimal length in seconds": 30.0, "limit track length to x seconds": 0.0, "pixel per micrometre": 0.7}, 12.0),
                          (4, {}, 25.0)):
        df = _long_enough(make_table(seed, n_tracks=40, max_len=700), 32)
        ref_rows, ref_stats = _compare(oracle, df, select_settings(**kw), fps)
        seen |= set(ref_stats["Motility Phenotype"].unique())
        assert ref_rows["turn_points"].sum() > len(ref_stats)            # real turning points, not only track starts
        assert (ref_rows["moving"] == 0).any() and np.isnan(ref_rows["tp_of_tracks"]).any()
    assert seen == {0, 1, 2}


def test_statistics_on_awkward_tables(oracle):
    """One long track, tracks shorter than every window (medfilt kernel, turning-point order, reach lag), a table
    whose last row is a turning point, standing tracks (no path at all)."""
    import pandas as pd
    from select_tables import make_table, select_settings
    s = select_settings()
    rng = np.random.default_rng(9)
    n = 5000
    xy = np.cumsum(rng.normal(0, 1.5, (n, 2)), axis=0) + 300
    one = pd.DataFrame({"TRACK_ID": np.zeros(n, np.uint32), "POSITION_T": np.arange(n, dtype=np.uint32), "POSITION_X": xy[:, 0],
                        "POSITION_Y": xy[:, 1], "WIDTH": rng.uniform(4, 8, n), "HEIGHT": rng.uniform(1, 3, n),
                        "DEGREES_ANGLE": rng.uniform(0, 90, n)})
    _compare(oracle, one, s, 30.0)
    short = _long_enough(make_table(6, n_tracks=60, max_len=14), 6)       # at 4 fps the kernel is 5 rows
    assert short["TRACK_ID"].nunique() > 10
    _compare(oracle, short, s, 4.0)
    still = _long_enough(make_table(7, n_tracks=10, max_len=300), 32)
    still.loc[still["TRACK_ID"] < 4, ["POSITION_X", "POSITION_Y"]] = [10.0, 20.0]
    _compare(oracle, still, s, 30.0)
    zig = one.iloc[:400].copy()
    zig["POSITION_X"] = np.where(np.arange(400) // 25 % 2 == 0, np.arange(400) * 2.0, 800 - np.arange(400) * 2.0)
    zig["POSITION_Y"] = np.arange(400) * 1.0
    _compare(oracle, zig, s, 30.0)


def test_evaluate_tracks_entry_point_and_files(tmp_path, oracle):
    """evaluate_tracks() as the reference exposes it: (df, df_stats), the two csv files written by the reference's
    own to_csv call, analyse() running it after the selection."""
    import pandas as pd
    from select_tables import make_table, select_settings
    from ysmr_amd import evaluate_tracks
    from ysmr_amd.evaluate import ROW_COLUMNS, STATS_COLUMNS
    df = _long_enough(make_table(5, n_tracks=25, max_len=500), 32)
    s = select_settings(**{"store generated statistical .csv file": True, "store final analysed .csv file": True})
    res = evaluate_tracks(str(tmp_path / "clip_selected_data.csv"), str(tmp_path), df=df, settings=s, fps=30.0)
    assert res is not None
    out, stats = res
    assert list(out.columns) == ROW_COLUMNS and list(stats.columns) == STATS_COLUMNS + ["Categories (Perc. Motile)"]
    assert stats.index.name == "TRACK_ID" and (stats["Categories (Perc. Motile)"] == "All").all()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_rows, ref_stats = oracle.evaluate_tracks_oracle(df, s, 30.0)
    pd.testing.assert_frame_equal(out, ref_rows, check_exact=True)
    pd.testing.assert_frame_equal(stats[STATS_COLUMNS], ref_stats, check_exact=True, check_names=False)
    ref_stats.to_csv(tmp_path / "ref_stats.csv", index=False, encoding="utf-8")
    ref_rows.to_csv(tmp_path / "ref_rows.csv", index=False, encoding="utf-8")
    assert (tmp_path / "clip_selected_data_statistics.csv").read_bytes() == (tmp_path / "ref_stats.csv").read_bytes()
    assert (tmp_path / "clip_selected_data_analysed.csv").read_bytes() == (tmp_path / "ref_rows.csv").read_bytes()
    assert evaluate_tracks(str(tmp_path / "x.csv"), str(tmp_path), df=df, settings=dict(s, **{"frames per second": 0.0}), fps=None) is None


def test_analyse_goes_on_to_the_statistics(tmp_path, oracle):
    """analyse() with an evaluation key set: video -> table -> selection -> evaluate_tracks, whose (df, df_stats)
    is what it returns (main.py:131-139), next to <name>_statistics.csv."""
    import pandas as pd
    from select_tables import select_settings
    from ysmr_amd import analyse
    from ysmr_amd.evaluate import STATS_COLUMNS
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(200, 260, 14, seed=3, dropout=0.01).frames(120)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    s = select_settings(**{"minimal frame count": 40, "store generated statistical .csv file": True})
    os.makedirs(tmp_path / "first")
    table = track_bacteria(str(path), settings=dict(s), result_folder=str(tmp_path / "first"))[0]
    selected, info = oracle.select_tracks_oracle(table, s, 30.0, 200, 260)
    assert selected is not None and info["good_tracks"] >= 3
    res = analyse(str(path), settings=dict(s), result_folder=str(tmp_path / "res"), return_df=True)
    assert isinstance(res, tuple) and len(res) == 2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_rows, ref_stats = oracle.evaluate_tracks_oracle(selected, s, 30.0)
    pd.testing.assert_frame_equal(res[0], ref_rows, check_exact=True)
    pd.testing.assert_frame_equal(res[1][STATS_COLUMNS], ref_stats, check_exact=True, check_names=False)
    assert (tmp_path / "res" / "clip_statistics.csv").exists()
    assert analyse(str(path), settings=dict(s), result_folder=str(tmp_path / "res2")) is True

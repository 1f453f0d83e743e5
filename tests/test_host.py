"""CPU tests of the host-side mirror of the reference interface: tracking.ini surface, the
*_list.csv wire format, frame sources, stream sharding (world_size 2 over gloo)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_default_settings_surface():
    from ysmr_amd.helper_file import TRACKING_INI, default_settings
    s = default_settings()
    assert len(TRACKING_INI) == 11 and sum(len(v) for v in TRACKING_INI.values()) == 85
    # the keys that reach the hot path and their upstream defaults
    assert s["white bacteria on dark background"] is True
    assert s["threshold offset for detection"] == 5 and s["adaptive double threshold"] == 2.0
    assert s["color filter"] == 6 and s["minimal frame count"] == 600 and s["list save length interval"] == 10000
    assert (s["disable gsff"], s["number of LSFFs"], s["minimum horizon size"], s["maximum horizon size"]) == (False, 3, 0, 30)
    assert s["user input"] and s["select files"] and s["display video analysis"]   # interactive upstream defaults
    assert s["maximal empty frames in %"] == pytest.approx(1.05) and s["percent quantiles excluded area"] == pytest.approx(0.1)


def test_get_configs_roundtrip_and_regeneration(tmp_path):
    from ysmr_amd.helper_file import default_settings, get_configs
    ini = tmp_path / "tracking.ini"
    assert get_configs(str(ini)) is None and ini.exists()      # missing file: regenerated, None returned
    s = get_configs(str(ini))
    d = default_settings()
    assert {k: v for k, v in s.items() if k != "tracking_ini_filepath"} == {k: v for k, v in d.items() if k != "tracking_ini_filepath"}
    passthrough = {"x": 1}
    assert get_configs(passthrough) is passthrough             # dicts pass through untouched
    text = ini.read_text().replace("maximum horizon size = 30", "maximum horizon size = none")
    ini.write_text(text)
    assert get_configs(str(ini))["maximum horizon size"] is None
    ini.write_text("[BASIC RECORDING SETTINGS]\nframes per second = 30\n")
    assert get_configs(str(ini)) is None                       # broken file: regenerated again


def test_csv_wire_format(tmp_path):
    from ysmr_amd import _lib
    from ysmr_amd.helper_file import CSV_HEADER, get_data, rows_to_csv_text, save_list, sort_list
    video = tmp_path / "clip.npy"
    video.write_bytes(b"")
    old, csv_path = save_list(path=str(video), result_folder=str(tmp_path), first_call=True)
    assert old is False and csv_path.endswith("clip_list.csv") and open(csv_path).read() == CSV_HEADER
    rows = np.zeros(3, _lib.ROW_DTYPE)
    rows[0] = (0, 1, 120.25001525878906, 7.5, np.float32(5.824352264404297), np.float32(2.5), np.float32(-45.0), 0)
    rows[1] = (1, 1, 121.0, 8.0, 0, 0, 0, 2)            # disappeared: the reference writes integer zeros
    rows[2] = (1, 0, 3.0, 4.0, np.float32(0.0), np.float32(0.0), np.float32(0.0), 0)   # 1-pixel blob: float zeros
    text = rows_to_csv_text(rows)
    # identical to what the reference's save_list formats from python objects (helper_file.py:1455-1475)
    ref = "".join("{0},{1},{2},{3},{4},{5},{6}\n".format(*t) for t in [
        (1, 0, np.float64(120.25001525878906), np.float64(7.5), float(np.float32(5.824352264404297)), 2.5, -45.0),
        (1, 1, np.float64(121.0), np.float64(8.0), 0, 0, 0),
        (0, 1, np.float64(3.0), np.float64(4.0), 0.0, 0.0, 0.0)])
    assert text == ref
    with open(csv_path, "a", newline="") as fh:
        fh.write(text)
    df = sort_list(file_path=csv_path, save_file=True)
    assert list(df["TRACK_ID"]) == [0, 1, 1] and list(df["POSITION_T"]) == [1, 0, 1]
    assert df.dtypes["TRACK_ID"] == np.uint32 and df.dtypes["POSITION_X"] == np.float64
    again = get_data(csv_path)
    np.testing.assert_array_equal(again.to_numpy(), df.to_numpy())
    # save_list keeps the reference's coords interface too
    save_list(path=csv_path, coords=[(2, 5, np.array([1.5, 2.5]), (1.0, 2.0, 3.0))])
    assert open(csv_path).read().endswith("5,2,1.5,2.5,1.0,2.0,3.0\n")


def test_previous_list_is_removed_small_at_once_large_beside_the_run(tmp_path):
    """save_list(first_call) without 'rename previous result .csv' (helper_file.py:1434-1445: os.remove): the name holds a
    header-only file when it returns, whatever the old list's size; a large one is unlinked by a thread that
    wait_for_removals (track_bacteria's last step) joins."""
    import os
    from ysmr_amd import helper_file as hf
    video = tmp_path / "clip.npy"
    video.write_bytes(b"")
    csv_path = str(tmp_path / "clip_list.csv")
    for size in (100, hf._REMOVE_ASIDE_FROM + 1):
        with open(csv_path, "wb") as fh:
            fh.write(b"x" * size)
        old, got = hf.save_list(path=str(video), result_folder=str(tmp_path), first_call=True, rename_old_list=False)
        assert old is False and got == csv_path and open(csv_path).read() == hf.CSV_HEADER
        hf.wait_for_removals()
        assert sorted(os.listdir(tmp_path)) == ["clip.npy", "clip_list.csv"] and not hf._REMOVALS
    # 'rename previous result .csv': the old list stays, under a dated name (helper_file.py:1437-1442)
    with open(csv_path, "wb") as fh:
        fh.write(b"y" * (hf._REMOVE_ASIDE_FROM + 1))
    old, got = hf.save_list(path=str(video), result_folder=str(tmp_path), first_call=True, rename_old_list=True)
    assert old and os.path.getsize(old) == hf._REMOVE_ASIDE_FROM + 1 and open(csv_path).read() == hf.CSV_HEADER


def test_reshape_result():
    from ysmr_amd.helper_file import reshape_result
    assert reshape_result(((1.0, 2.0), (3.0, 4.0), -45.0)) == ((1.0, 2.0), (3.0, 4.0, -45.0))
    assert reshape_result(((1.0, 2.0), (3.0, 4.0), -45.0), 0.5) == ((1.0, 2.0, 0.5), (3.0, 4.0, -45.0))


def test_frame_sources(tmp_path):
    from ysmr_amd.frames import open_video
    rng = np.random.default_rng(0)
    clip = rng.integers(0, 256, (7, 12, 16), dtype=np.uint8)
    p = tmp_path / "a.npy"
    np.save(p, clip)
    (tmp_path / "a_meta.json").write_text('{"fps": 29.97}')
    v = open_video(str(p))
    assert (v.frame_count, v.height, v.width, v.channels, v.fps) == (7, 12, 16, 1, 29.97)
    np.testing.assert_array_equal(v.read(5, 4), clip[5:7])
    from concurrent.futures import ThreadPoolExecutor
    big = rng.integers(0, 256, (40, 12, 16), dtype=np.uint8)
    np.save(tmp_path / "big.npy", big)
    vb, out = open_video(str(tmp_path / "big.npy")), np.zeros((32, 12, 16), np.uint8)
    with ThreadPoolExecutor(3) as pool:
        assert vb.read_into(4, 32, out, pool) == 32 and np.array_equal(out, big[4:36])
        assert vb.read_into(30, 32, out, pool) == 10 and np.array_equal(out[:10], big[30:])
    # y4m: luma plane only, 4:2:0
    y4 = tmp_path / "b.y4m"
    with open(y4, "wb") as fh:
        fh.write(b"YUV4MPEG2 W16 H12 F30000:1001 Ip A1:1 C420jpeg\n")
        for f in clip:
            fh.write(b"FRAME\n" + f.tobytes() + bytes(2 * 8 * 6))
    v = open_video(str(y4))
    assert (v.frame_count, v.height, v.width) == (7, 12, 16) and abs(v.fps - 29.97) < 1e-2
    np.testing.assert_array_equal(v.read(0, 7), clip)
    out = np.zeros((4, 12, 16), np.uint8)
    assert v.read_into(5, 4, out) == 2 and np.array_equal(out[:2], clip[5:])
    with pytest.raises(OSError):
        open_video(str(tmp_path / "c.avi"))      # no OpenCV in this image


def test_track_bacteria_error_conventions(tmp_path, caplog):
    """Failures are logged and signalled by None, never raised (track_eval.py:50-77)."""
    from ysmr_amd.helper_file import default_settings
    from ysmr_amd.track_eval import track_bacteria
    s = default_settings(**{"user input": False, "select files": False, "display video analysis": False,
                            "log to file": False})
    assert track_bacteria(str(tmp_path / "missing.npy"), settings=s, result_folder=str(tmp_path)) is None
    short = tmp_path / "short.npy"
    np.save(short, np.zeros((10, 8, 8), np.uint8))
    assert track_bacteria(str(short), settings=s, result_folder=str(tmp_path)) is None   # < minimal frame count
    s2 = dict(s); s2["include luminosity in tracking calculation"] = True
    s2["minimal frame count"] = 5
    assert track_bacteria(str(short), settings=s2, result_folder=str(tmp_path)) is None  # unsupported option


def _gloo_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    from ysmr_amd import dist
    info = dist.rank_info()
    dist.init(info, backend="gloo")
    streams = [f"video{i}" for i in range(5)]
    mine = dist.shard(streams, info.rank, info.world)
    dist.barrier(info)
    elapsed = dist.max_over_ranks(1.0 + info.rank, info)      # slowest rank defines the job time
    rates = dist.gather_over_ranks(100.0 * (1 + info.rank), info)     # every rank's own rate, in rank order, on every rank
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as fh:
        fh.write(f"{','.join(mine)};{elapsed};{rates}")
    dist.finish(info)


def test_stream_sharding_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "r0.txt").read_text().split(";")
    r1 = (tmp_path / "r1.txt").read_text().split(";")
    assert r0[0] == "video0,video2,video4" and r1[0] == "video1,video3"
    assert float(r0[1]) == float(r1[1]) == 2.0
    assert r0[2] == r1[2] == "[100.0, 200.0]"


# ---- a19 / f2: the csv the reference ends up with ------------------------------------------------------

def _random_rows(n, seed):
    from ysmr_amd import _lib
    rng = np.random.default_rng(seed)
    rows = np.zeros(n, dtype=_lib.ROW_DTYPE)
    rows["frame"] = rng.integers(0, 100000, n)
    rows["track_id"] = rng.integers(0, 5000, n)
    rows["x"] = rng.uniform(-5, 1300, n)
    rows["y"] = rng.uniform(-5, 1000, n)
    rows["w"] = rng.uniform(0, 40, n).astype(np.float32)
    rows["h"] = rng.uniform(0, 40, n).astype(np.float32)
    rows["angle"] = rng.uniform(-90, 0, n).astype(np.float32)
    gone = rng.random(n) < 0.2
    rows["disappeared"] = gone * rng.integers(1, 30, n)
    for k in ("w", "h", "angle"):
        rows[k][gone] = 0
    return rows


def _pandas_csv(rows, tmp_path):
    """DataFrame.to_csv exactly as save_df_to_csv calls it (helper_file.py:1392-1394), on the exact values."""
    from ysmr_amd.helper_file import rows_to_dataframe
    path = tmp_path / "ref.csv"
    with open(path, "w+", newline="\n") as fh:
        rows_to_dataframe(rows, via_pandas=False).to_csv(fh, index=False, encoding="utf-8")
    return path.read_bytes()


_SPECIAL = [0.0, -0.0, 1.0, 100000.0, 1e15, 9999999999999998.0, 1e16, 1.5e16, 1e22, 1e-4, 0.00012345, 9.999e-5, 1e-5,
            1.5e-7, 123456.789, 0.1, 1 / 3, 2 / 3, 5e-324, 1.7976931348623157e308, 922.0000000000001, 613.5,
            float(np.float32(3.6)), float(np.float32(-89.98)), -1234.5678, 1e100, 1.2345678901234567e-100,
            0.00012345678901234567, 1234567890123456.7, 4.35e-320]


def _with_special_values(rows):
    # values that exercise every branch of the float layout: exponent form on both sides, integral
    # values, leading zeros, 17 significant digits, negative zero, subnormals, float32 widened to float64
    k = len(_SPECIAL)
    rows["x"][:k] = _SPECIAL
    rows["y"][:k] = _SPECIAL[::-1]
    rows["w"][:6] = np.float32([0.1, 1e-8, 3.4e38, 16777216.0, 1e10, 2.5])
    rows["disappeared"][:6] = 0      # (a disappeared track always carries zeros: tracker.py:101, 205)
    return rows


def test_native_csv_matches_pandas_bytes(tmp_path):
    """ysmr_rows_format_csv (a host function of the library) against DataFrame.to_csv, byte for byte."""
    from ysmr_amd.helper_file import rows_to_csv_bytes
    rows = _with_special_values(_random_rows(20000, 3))
    for threads in (1, 4):
        got = rows_to_csv_bytes(rows, header=True, via_pandas=False, threads=threads)
        assert got == _pandas_csv(rows, tmp_path)
    assert rows_to_csv_bytes(rows[:0]) == b"TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE\n"
    assert rows_to_csv_bytes(rows[:3], header=False, via_pandas=False) == \
        b"".join(_pandas_csv(rows[:3], tmp_path).split(b"\n", 1)[1:])


def test_output_equals_the_reference_text_pandas_detour(tmp_path):
    """a19 end to end.  The reference appends Python-formatted rows (track_eval.py:313-316, 340-346;
    integer zeros for disappeared tracks, tracker.py:101, 205), re-reads the file with
    pandas.read_csv, sorts by (TRACK_ID, POSITION_T) and rewrites it with to_csv (sort_list,
    helper_file.py:1538-1574).  pandas' float parser is not correctly rounded, so that detour is not
    the identity; the native path restates it.  DataFrame and file must come out identical."""
    import pandas as pd
    from ysmr_amd.helper_file import get_data, rows_to_csv_bytes, rows_to_csv_text, rows_to_dataframe, sort_list
    rows = _with_special_values(_random_rows(30000, 5))
    rows["x"][40:] = np.random.default_rng(6).uniform(0, 1300, len(rows) - 40)
    rows["frame"] = np.arange(len(rows)) // 50          # (track_id, frame) must be unique
    rows["track_id"] = np.arange(len(rows)) % 50 * 7
    # --- the reference's way
    ref_path = tmp_path / "ref_list.csv"
    with open(ref_path, "w", newline="") as fh:
        fh.write("TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE\n")
        fh.write(rows_to_csv_text(rows))
    df_ref = sort_list(file_path=str(ref_path), save_file=True)
    # --- the native way
    order = np.lexsort((rows["frame"], rows["track_id"]))
    srt = rows[order]
    df = rows_to_dataframe(srt)
    assert (df_ref["POSITION_X"].to_numpy() != srt["x"]).any(), "expected pandas' parser to perturb some values"
    pd.testing.assert_frame_equal(df, df_ref, check_exact=True)
    assert rows_to_csv_bytes(srt) == ref_path.read_bytes()


def test_native_csv_round_trips_through_get_data(tmp_path):
    """The file written natively reads back (get_data, helper_file.py:860-905) to the frame handed over."""
    from ysmr_amd.helper_file import get_data, rows_to_csv_bytes, rows_to_dataframe
    rows = _random_rows(5000, 4)
    order = np.lexsort((rows["frame"], rows["track_id"]))
    rows = rows[order]
    path = tmp_path / "x_list.csv"
    path.write_bytes(rows_to_csv_bytes(rows))
    df = get_data(str(path), check_sorted=False)
    ref = rows_to_dataframe(rows)
    assert list(df.columns) == list(ref.columns)
    for c in ref.columns:
        assert df[c].dtype == ref[c].dtype
    # reading the final file once more goes through the inexact parser again: equal to 1 ulp
    for c in ("POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"):
        np.testing.assert_allclose(df[c].to_numpy(), ref[c].to_numpy(), rtol=4e-16, atol=0)
    for c in ("TRACK_ID", "POSITION_T"):
        np.testing.assert_array_equal(df[c].to_numpy(), ref[c].to_numpy())


def test_uncompressed_avi_reader(tmp_path):
    from avi_tools import write_avi as _write_avi
    from ysmr_amd.frames import AviVideo, open_video
    rng = np.random.default_rng(4)
    gray = rng.integers(0, 256, (9, 10, 13), dtype=np.uint8)              # width 13: rows padded to 16 bytes
    _write_avi(tmp_path / "g.avi", gray, 8)
    v = open_video(str(tmp_path / "g.avi"))
    assert isinstance(v, AviVideo) and (v.frame_count, v.height, v.width, v.channels) == (9, 10, 13, 1)
    assert abs(v.fps - 29.97) < 1e-2
    np.testing.assert_array_equal(v.read(0, 9), gray)
    np.testing.assert_array_equal(v.read(7, 5), gray[7:])
    out = np.zeros((4, 10, 13), np.uint8)
    assert v.read_into(2, 4, out) == 4 and np.array_equal(out, gray[2:6])
    _write_avi(tmp_path / "t.avi", gray, 8, top_down=True, split=4)      # top-down, continued in RIFF AVIX
    v = open_video(str(tmp_path / "t.avi"))
    assert v.frame_count == 9
    np.testing.assert_array_equal(v.read(0, 9), gray)
    bgr = rng.integers(0, 256, (5, 6, 7, 3), dtype=np.uint8)
    _write_avi(tmp_path / "c.avi", bgr, 24)
    v = open_video(str(tmp_path / "c.avi"))
    assert v.channels == 3
    np.testing.assert_array_equal(v.read(0, 5), bgr)
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)                   # a colour palette: frames come out as BGR
    _write_avi(tmp_path / "p.avi", gray, 8, palette=pal)
    v = open_video(str(tmp_path / "p.avi"))
    assert v.channels == 3
    np.testing.assert_array_equal(v.read(0, 9), pal[gray])
    (tmp_path / "bad.avi").write_bytes(b"RIFF" + bytes(60))
    with pytest.raises(OSError):                                          # not readable natively and no OpenCV here
        open_video(str(tmp_path / "bad.avi"))
    # an empty chunk is a dropped frame: the previous frame again, still one frame slot (frame numbers keep
    # their alignment with what cv2.VideoCapture delivers); a short chunk is an error, not a skipped frame
    _write_avi(tmp_path / "d.avi", gray, 8, dropped=(3, 4, 8))
    v = open_video(str(tmp_path / "d.avi"))
    assert v.frame_count == 9
    np.testing.assert_array_equal(v.read(0, 9), gray[[0, 1, 2, 2, 2, 5, 6, 7, 7]])
    _write_avi(tmp_path / "s.avi", gray, 8, truncated=5)
    with pytest.raises(ValueError, match="chunk 5"):
        AviVideo(str(tmp_path / "s.avi"))
    import gc, warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error", ResourceWarning)                   # the failed constructor closed its file
        gc.collect()


def test_select_tracks_argument_checks(tmp_path, caplog):
    """select_tracks logs and returns None before any device work (track_eval.py:560-611)."""
    import logging
    from select_tables import make_table, select_settings
    from ysmr_amd.select import select_params, select_tracks
    df = make_table(0, n_tracks=3)
    s = select_settings()
    caplog.set_level(logging.DEBUG, logger="ysmr")
    assert select_tracks(path_to_file=None, df=df, settings=s) is None
    bad = dict(s); bad["frames per second"] = 0
    assert select_tracks(path_to_file=str(tmp_path / "a_list.csv"), df=df, results_directory=str(tmp_path), fps=0, settings=bad) is None
    bad = dict(s); bad["extreme area outliers lower end in px*px"] = 50
    assert select_tracks(path_to_file=str(tmp_path / "a_list.csv"), df=df, results_directory=str(tmp_path), fps=30.0, settings=bad) is None
    assert select_tracks(path_to_file=str(tmp_path / "a_list.csv"), df=df, results_directory=str(tmp_path), fps=30.0,
                         frame_height=0, frame_width=10, settings=s) is None
    bad = dict(s); bad["pixel per micrometre"] = 0.0
    assert select_tracks(path_to_file=str(tmp_path / "a_list.csv"), df=df, results_directory=str(tmp_path), fps=30.0, settings=bad) is None
    assert select_tracks(path_to_file=str(tmp_path / "missing_list.csv"), results_directory=str(tmp_path), fps=30.0, settings=s) is None
    text = caplog.text
    for needle in ("needs path_to_file", "fps value is negative or zero", "Minimal area exclusion", "Frame width or frame height",
                   "pixel per micrometre", "Error reading data frame"):
        assert needle in text, needle
    p = select_params(s, 29.97, 922, 1228)       # lengths use round(fps): 30 frames per second
    assert (p.min_length_frames, p.limit_frames, p.frame_height, p.frame_width) == (30, 90, 922, 1228)
    assert abs(p.max_empty_ratio - 1.05) < 1e-12 and abs(p.q_area - 0.1) < 1e-12 and p.max_recursion == 960


def test_motion_jpeg_avi_reader(tmp_path):
    """Motion-JPEG AVI through Pillow (optional): frames equal Pillow's own decode of the same JPEGs."""
    Image = pytest.importorskip("PIL.Image")
    import io
    from concurrent.futures import ThreadPoolExecutor
    from avi_tools import write_avi
    from ysmr_amd.frames import open_video
    rng = np.random.default_rng(2)
    smooth = (np.add.outer(np.arange(48), np.arange(64)) % 200 + 20).astype(np.uint8)
    gray = np.stack([np.roll(smooth, 3 * k, axis=1) for k in range(6)])
    for mode, frames in (("L", gray), ("RGB", np.stack([gray, gray[:, ::-1], 255 - gray], axis=-1))):
        blobs, decoded = [], []
        for f in frames:
            buf = io.BytesIO()
            Image.fromarray(f, mode).save(buf, format="JPEG", quality=92)
            blobs.append(buf.getvalue())
            with Image.open(io.BytesIO(blobs[-1])) as im:
                decoded.append(np.asarray(im.convert(mode)))
        path = tmp_path / f"m{mode}.avi"
        write_avi(path, frames[..., 0] if mode == "RGB" else frames, 24, fps=(25, 1), jpeg=blobs)
        v = open_video(str(path))
        assert (v.frame_count, v.height, v.width, v.fps) == (6, 48, 64, 25.0)
        assert v.channels == (1 if mode == "L" else 3)
        ref = np.stack(decoded) if mode == "L" else np.stack(decoded)[..., ::-1]      # BGR like cv2
        np.testing.assert_array_equal(v.read(0, 6), ref)
        out = np.zeros_like(ref[:4])
        with ThreadPoolExecutor(3) as pool:
            assert v.read_into(2, 4, out, pool) == 4
        np.testing.assert_array_equal(out, ref[2:6])
        assert np.abs(v.read(0, 1)[0].astype(int) - (frames[0] if mode == "L" else frames[0][..., ::-1]).astype(int)).mean() < 6


def test_bench_argument_presets(monkeypatch):
    """bench.py's command line: the driver's contract flags and the BASELINE configuration presets."""
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert a.batch is None and a.frames is None          # (parse() does not load the library: main() builds it first)
    a = bench.resolve_batch(a)
    # (248 frames per batch: one per workgroup of the threshold kernel beside the batch link; two batches per step)
    assert (a.gpus, a.steps, a.warmup, a.height, a.width, a.blobs, a.batch, a.frames, a.detect_only, a.adt) == \
        (1, 5, 1, 922, 1228, 500, 248, 496, False, 2.0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "2", "--config", "4"])
    a = bench.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 3, 2)
    assert (a.height, a.width, a.blobs, a.frames, a.batch, a.max_det, a.capacity) == (2160, 3840, 5000, 64, 16, 8192, 8192)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "4", "--batch", "8"])
    assert bench.parse().batch == 8
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "1"])
    a = bench.resolve_batch(bench.parse())
    assert a.detect_only and (a.batch, a.frames) == (256, 512)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--config", "0"])
    assert bench.parse().blobs == 50


def test_frames_per_batch_follow_the_frame_size():
    """track_bacteria's batch when neither the call nor the settings name one: ~300 MB of frames, 16 ... 256."""
    from ysmr_amd.track_eval import auto_batch
    # (248 = ysmr_threshold_workgroups beside the batch link: a frame per workgroup, not 256 frames on 248 workgroups)
    assert auto_batch(922, 1228) == 248 and auto_batch(2160, 3840) == 32 and auto_batch(2160, 3840, 3) == 16
    assert auto_batch(200, 260) == 248 and auto_batch(8000, 8000) == 16 and auto_batch(1080, 1920) == 144


def test_ysmr_rejects_missing_paths_without_a_gpu(tmp_path, caplog):
    """ysmr(): settings come first, a missing file is reported per path and does not stop the others
    (main.py:292-313); nothing here touches the device."""
    from ysmr_amd import ysmr
    from ysmr_amd.helper_file import default_settings
    s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
    done = ysmr([str(tmp_path / "a.npy"), str(tmp_path / "b.npy")], settings=s, result_folder=str(tmp_path / "out"))
    assert [r for _, r in done] == [None, None] and "Failed to analyse 2 of 2" in caplog.text
    assert ysmr([], settings=dict(s, **{"select files": True}), result_folder=str(tmp_path / "out")) is None


def _mgpu_settings():
    from ysmr_amd.helper_file import default_settings
    return default_settings(**{"user input": False, "select files": False, "display video analysis": False,
                               "log to file": False})


def test_ysmr_multiprocess_one_worker_process_per_gpu(tmp_path, monkeypatch):
    """The branch of ysmr(multiprocess=True) that starts one worker PROCESS per GPU (main.py:281-288
    upstream: one per file) -- on a host that reports two GPUs, with the per-GPU worker replaced by a stub:
    paths are dealt round-robin, every GPU's share runs in its own spawned process, results come back in
    the order of the paths."""
    import torch
    import mgpu_stub
    from ysmr_amd import main
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: False)
    monkeypatch.setattr(main, "_gpu_worker", mgpu_stub.fake_gpu_worker)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,6")        # this process's cuda:0 / cuda:1 are the host's GPUs 4 and 6
    paths = [str(tmp_path / f"v{i}.npy") for i in range(5)]
    paths[3] = paths[1]                                     # a path given twice is analysed twice, results by position
    done = main.ysmr(paths, settings=_mgpu_settings(), result_folder=str(tmp_path), multiprocess=True, streams_per_gpu=3)
    assert [p for p, _ in done] == paths
    assert [r["device"] for _, r in done] == ["cuda:0", "cuda:1", "cuda:0", "cuda:1", "cuda:0"]
    pids = {r["device"]: r["pid"] for _, r in done}
    assert len(set(pids.values())) == 2 and os.getpid() not in pids.values()     # two workers, neither is this process
    assert all(r["pid"] == pids[r["device"]] and r["streams"] == 3 for _, r in done)
    # every worker process is told which GPU of the HOST is its own (it makes that its HIP_VISIBLE_DEVICES)
    assert {r["device"]: r["physical"] for _, r in done} == {"cuda:0": "4", "cuda:1": "6"}


def test_gpu_worker_sees_only_its_gpu_and_sits_next_to_it(tmp_path, monkeypatch):
    """The placement a per-GPU worker process gives itself (ysmr_amd/dist.py): HIP_VISIBLE_DEVICES = its GPU, its jobs
    renamed to cuda:0, CPU affinity = the GPU's NUMA-local CPUs that the process may use -- against a made-up sysfs tree;
    nothing is pinned when sysfs does not know, when too few CPUs would be left, or when the GPU cannot be opened."""
    from ysmr_amd import dist, main
    sys_root = tmp_path / "sys"
    dev = sys_root / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "local_cpulist").write_text("48-55,144-147\n")
    want = set(range(48, 56)) | set(range(144, 148))
    assert dist.local_cpus("0000:c1:00.0", str(sys_root)) == want
    other = sys_root / "bus" / "pci" / "devices" / "0000:05:00.0"
    other.mkdir(parents=True)
    (other / "numa_node").write_text("1\n")
    node = sys_root / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127\n")
    assert dist.local_cpus("0000:05:00.0", str(sys_root)) == set(range(64, 128))
    (other / "numa_node").write_text("-1\n")
    assert dist.local_cpus("0000:05:00.0", str(sys_root)) is None and dist.local_cpus(None, str(sys_root)) is None
    pinned = []
    assert dist.pin_to_gpu(0, str(sys_root), bus_id="0000:c1:00.0", allowed=range(0, 100), setter=pinned.append) == set(range(48, 56))
    assert pinned == [set(range(48, 56))]
    assert dist.pin_to_gpu(0, str(sys_root), bus_id="0000:c1:00.0", allowed=range(53, 60), setter=pinned.append) is None   # 3 CPUs: too few
    assert dist.pin_to_gpu(0, str(sys_root), bus_id="0000:ff:00.0", allowed=range(0, 100), setter=pinned.append) is None
    assert len(pinned) == 1
    assert dist.physical_device(1, {"HIP_VISIBLE_DEVICES": "2, 5"}) == "5" and dist.physical_device(3, {}) == "3"
    # the worker itself: environment, job names, and the fold when the parent counted a GPU this process cannot open
    calls = []
    monkeypatch.setattr(main, "_worker", lambda job: calls.append(job[3]) or (job[0], True))
    monkeypatch.setattr(dist, "pin_to_gpu", lambda index=0, **kw: {1, 2, 3, 4})
    monkeypatch.setattr(dist, "usable_gpus", lambda dev_root="/dev/dri": 8)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    jobs = [("a.npy", {}, "out", "cuda:5"), ("b.npy", {}, "out", "cuda:5")]
    assert main._gpu_worker((jobs, 1, "5")) == [("a.npy", True), ("b.npy", True)]
    assert os.environ["HIP_VISIBLE_DEVICES"] == "5" and calls == ["cuda:0", "cuda:0"]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.setattr(dist, "usable_gpus", lambda dev_root="/dev/dri": 1)
    calls.clear()
    main._gpu_worker((jobs, 1, "5"))
    assert "HIP_VISIBLE_DEVICES" not in os.environ and calls == ["cuda:5", "cuda:5"]
    # (in-process workers -- the parent holds a GPU context -- change neither the environment nor the names)
    calls.clear()
    main._gpu_worker((jobs, 1, None))
    assert "HIP_VISIBLE_DEVICES" not in os.environ and calls == ["cuda:5", "cuda:5"]


def test_eight_ranks_get_eight_disjoint_cpu_sets_and_all_streams(tmp_path):
    """An 8-GPU node as sysfs would describe it (two sockets, four GPUs each, every GPU with a NUMA-local range of its
    own): ``dist.pin_to_gpu`` gives the eight ranks eight disjoint CPU sets, ``dist.shard`` deals sixteen streams two to a
    rank with none left over, and ``physical_device`` names eight different GPUs."""
    from ysmr_amd import dist
    sys_root = tmp_path / "sys"
    buses = []
    for g in range(8):
        bus = "0000:%02x:00.0" % (0x05 + 0x10 * g)
        dev = sys_root / "bus" / "pci" / "devices" / bus
        dev.mkdir(parents=True)
        lo = 16 * g + (0 if g < 4 else 64)             # socket 0: cpus 0-63 (+ SMT 128-191), socket 1: 128.. shifted
        (dev / "local_cpulist").write_text(f"{lo}-{lo + 15},{lo + 256}-{lo + 271}\n")
        buses.append(bus)
    sets = []
    for rank in range(8):
        got = dist.pin_to_gpu(rank, str(sys_root), bus_id=buses[rank], allowed=range(0, 512), setter=lambda cpus: None)
        assert got and len(got) == 32
        sets.append(got)
    for a in range(8):
        for b in range(a + 1, 8):
            assert not (sets[a] & sets[b]), (a, b)
    streams = list(range(16))
    dealt = [dist.shard(streams, r, 8) for r in range(8)]
    assert all(len(d) == 2 for d in dealt) and sorted(sum(dealt, [])) == streams
    assert sorted({dist.physical_device(r, {}) for r in range(8)}) == [str(r) for r in range(8)]
    assert [dist.physical_device(r, {"HIP_VISIBLE_DEVICES": "7,6,5,4,3,2,1,0"}) for r in range(8)] == list("76543210")


def test_ysmr_multiprocess_stays_in_process_once_the_gpu_is_initialised(tmp_path, monkeypatch, caplog):
    """A process that already holds a GPU context must not start workers from itself (the worker would be
    forked/exec'ed out of a GPU-initialised parent): the per-GPU workers then run as threads."""
    import logging
    import torch
    import mgpu_stub
    from ysmr_amd import main
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    monkeypatch.setattr(main, "_gpu_worker", mgpu_stub.fake_gpu_worker)
    paths = [str(tmp_path / f"v{i}.npy") for i in range(4)]
    with caplog.at_level(logging.WARNING, logger="ysmr"):
        done = main.ysmr(paths, settings=_mgpu_settings(), result_folder=str(tmp_path), multiprocess=True)
    assert [p for p, _ in done] == paths
    assert all(r["pid"] == os.getpid() and r["thread"].startswith("ysmr-gpu") for _, r in done)
    assert {r["device"] for _, r in done} == {"cuda:0", "cuda:1"}
    assert any("already initialised the GPU" in rec.getMessage() for rec in caplog.records)


def test_worker_selects_the_device_of_its_job(monkeypatch):
    """_worker makes its job's GPU current before anything is allocated (a fresh worker's current device
    is cuda:0 whatever GPU the job was dealt to)."""
    from ysmr_amd import _lib, main
    seen = []

    class Ctx:
        def __init__(self, dev): self.dev = dev
        def __enter__(self): seen.append(("enter", self.dev))
        def __exit__(self, *a): seen.append(("exit", self.dev))
    monkeypatch.setattr(_lib, "on", lambda dev: Ctx(dev))
    monkeypatch.setattr(main, "analyse", lambda path, **kw: seen.append(("analyse", kw["device"])) or True)
    assert main._worker(("x.npy", {}, "out", "cuda:5")) == ("x.npy", True)
    assert seen == [("enter", "cuda:5"), ("analyse", "cuda:5"), ("exit", "cuda:5")]


def test_csv_written_by_the_formatting_threads_equals_the_one_buffer_form(tmp_path):
    """``ysmr_rows_write_csv`` (every thread formats its range and writes its own piece at its place in the file) leaves
    the bytes of ``ysmr_rows_format_csv``, with and without header, for tables below and above the size at which the
    work is split, and for the thread counts that do not divide the rows."""
    from ysmr_amd import _lib
    from ysmr_amd.helper_file import rows_to_csv_bytes, rows_to_csv_file
    rng = np.random.default_rng(3)
    for n, threads in ((0, 0), (7, 0), (5000, 3), (40001, 7), (40001, 0)):
        rows = np.zeros(n, _lib.ROW_DTYPE)
        rows["track_id"] = np.sort(rng.integers(0, 900, n))
        rows["frame"] = rng.integers(0, 100000, n)
        rows["x"], rows["y"] = rng.uniform(-5, 4000, n), rng.uniform(0, 1e-3, n)
        rows["w"], rows["h"], rows["angle"] = rng.uniform(0, 30, n), rng.uniform(0, 30, n), rng.uniform(-90, 90, n)
        rows["w"][::7] = 0
        for header in (True, False):
            for via in (True, False):
                path = tmp_path / f"t_{n}_{threads}_{header}_{via}.csv"
                path.write_bytes(b"stale content that is longer than a short table's text " * 3)
                length = rows_to_csv_file(rows, str(path), header=header, via_pandas=via, threads=threads)
                want = rows_to_csv_bytes(rows, header=header, via_pandas=via)
                assert length == len(want) and path.read_bytes() == want
                # ... and the one-pass form leaves the same file and the same DataFrame as the two separate calls
                from ysmr_amd.helper_file import rows_to_csv_file_and_dataframe, rows_to_dataframe
                path2 = tmp_path / f"u_{n}_{threads}_{header}_{via}.csv"
                length2, df2 = rows_to_csv_file_and_dataframe(rows, str(path2), header=header, via_pandas=via, threads=threads)
                assert length2 == len(want) and path2.read_bytes() == want
                ref = rows_to_dataframe(rows, via_pandas=via)
                assert df2.equals(ref) and list(df2.dtypes) == list(ref.dtypes)


def test_frame_feed_threads_run_next_to_the_gpu_and_the_caller_stays_where_it_was(monkeypatch):
    """DeviceFrameFeed(near_gpu=True): the CPUs of the GPU's NUMA node as far as this thread may use them (frames.cpus_near_gpu),
    taken on for the allocation of the staging buffers only (frames._ThreadOn restores the caller's mask)."""
    import os
    import threading
    from ysmr_amd import dist, frames
    mine = sorted(os.sched_getaffinity(0))
    monkeypatch.setattr(dist, "pci_bus_id", lambda index=0: "0000:c1:00.0")
    frames._NODE_CPUS.clear()
    monkeypatch.setattr(dist, "local_cpus", lambda bus_id, sysfs_root="/sys": set(mine[:4]) | {100000} if len(mine) >= 4 else None)
    if len(mine) >= 4:
        assert frames.cpus_near_gpu("cuda:0") == set(mine[:4])
        seen = {}
        with frames._ThreadOn(set(mine[:4])):
            seen["inside"] = os.sched_getaffinity(0)
            th = threading.Thread(target=lambda: seen.__setitem__("child", os.sched_getaffinity(0)))
            th.start(); th.join()
        assert seen["inside"] == set(mine[:4]) and seen["child"] == set(mine[:4])      # (threads inherit the mask)
        assert os.sched_getaffinity(0) == set(mine)
    frames._NODE_CPUS.clear()
    monkeypatch.setattr(dist, "local_cpus", lambda bus_id, sysfs_root="/sys": set(mine[:3]))
    assert frames.cpus_near_gpu("cuda:0") is None                                       # fewer than four: left alone
    frames._NODE_CPUS.clear()
    monkeypatch.setattr(dist, "local_cpus", lambda bus_id, sysfs_root="/sys": None)
    assert frames.cpus_near_gpu("cuda:0") is None
    with frames._ThreadOn(None):
        assert os.sched_getaffinity(0) == set(mine)
    frames._NODE_CPUS.clear()


def test_gc_freeze_is_counted_across_overlapping_passes():
    """track_bacteria's frame loops hold the garbage collector's view of the heap frozen (a full collection is a 40-ms
    stall of the loop that feeds the GPU).  Two stream threads per GPU worker overlap their passes: the first to finish
    must not thaw the heap under the other, and a freeze of the caller's own is never undone."""
    import gc
    import threading
    from ysmr_amd import track_eval as te
    assert gc.get_freeze_count() == 0 and te._GC_HOLDERS == 0
    first_in, second_in, first_out = threading.Event(), threading.Event(), threading.Event()
    seen = {}

    def first():
        te._gc_hold()
        first_in.set()
        second_in.wait(10)
        te._gc_release()                       # the other pass is still inside
        seen["after_first_left"] = gc.get_freeze_count()
        first_out.set()

    def second():
        first_in.wait(10)
        te._gc_hold()
        second_in.set()
        first_out.wait(10)
        seen["second_still_inside"] = gc.get_freeze_count()
        te._gc_release()
        seen["after_both"] = gc.get_freeze_count()

    threads = [threading.Thread(target=first), threading.Thread(target=second)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(20)
    assert seen["after_first_left"] > 0 and seen["second_still_inside"] > 0
    assert seen["after_both"] == 0 and te._GC_HOLDERS == 0
    # a caller's own freeze stays the caller's
    gc.freeze()
    try:
        te._gc_hold()
        te._gc_release()
        assert gc.get_freeze_count() > 0
    finally:
        gc.unfreeze()


def test_design_quotes_the_parity_record_it_names():
    """DESIGN.md section 2 quotes the bench configuration's parity record; rounds 3 and 4 each shipped a quote that the
    tracked file contradicted.  The quote now carries a machine-readable line naming its file: every
    figure in it must be the file's, to the digits quoted."""
    import json
    import re
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- parity-record file=(\S+) (.*?) -->", text)
    assert m, "DESIGN.md lost its parity-record line"
    rec = json.load(open(os.path.join(ROOT, m.group(1))))
    q = dict(kv.split("=") for kv in m.group(2).split())
    assert int(q["rows"]) == rec["rows"] and int(q["lost"]) == rec["rows_of_lost_tracks"]
    assert int(q["marked"]) == rec["ill_conditioned_rows"]
    assert int(q["beyond"]) == rec["ill_conditioned_rows_beyond_1e-5_relative"]
    assert q["worst_px"] == f"{rec['worst_ill_conditioned_px']:.4f}"
    assert float(q["worst_rel"]) == float(f"{rec['worst_ill_conditioned_relative']:.1e}")
    assert q["over_sens"] == f"{rec['worst_deviation_over_sens']:.2f}"
    assert float(q["well"]) == float(f"{rec['worst_well_conditioned_relative']:.1e}")
    # ... and the prose beside it says the same
    para = text[m.start() - 900:m.start()]
    for needle in ("60 of 100 828", "0.0030 px", "5.2e-6", "0.53 x", "4.4e-12"):
        assert needle in para, needle


def _tracker_like_rows(n_frames, n_tracks, seed):
    """Rows as the link emits them: frame-major, ids ascending within a frame; tracks are born and die, ids never return."""
    from ysmr_amd import _lib
    rng = np.random.default_rng(seed)
    born = np.sort(rng.integers(0, max(1, n_frames - 2), n_tracks)); born[: n_tracks // 3] = 0
    born.sort()
    life = rng.integers(1, n_frames, n_tracks)
    out = []
    for f in range(n_frames):
        ids = np.nonzero((born <= f) & (f < born + life))[0]
        r = np.zeros(len(ids), _lib.ROW_DTYPE)
        r["frame"], r["track_id"] = f, ids
        r["x"], r["y"] = rng.uniform(-5, 1300, len(ids)), rng.uniform(0, 1000, len(ids))
        r["w"], r["h"], r["angle"] = rng.uniform(0, 30, len(ids)), rng.uniform(0, 30, len(ids)), rng.uniform(0, 90, len(ids))
        gone = rng.random(len(ids)) < 0.1
        for k in ("w", "h", "angle"):
            r[k][gone] = 0
        r["disappeared"] = gone
        out.append(r)
    return np.concatenate(out)


def test_row_stream_equals_sorting_then_writing(tmp_path):
    """``ysmr_rows_stream_*`` (ABI 13): rows pushed batch by batch in the order the link emits them, formatted while later
    batches arrive, ordered at the end -- the csv and the DataFrame are those of sorting the whole table by
    (TRACK_ID, POSITION_T) and writing it in one go (``rows_to_csv_file_and_dataframe``), byte for byte: for the tracker's
    tables (a row's place from its id and frame alone), for tables that break the tracker's invariants (gaps, duplicate
    (id, frame) pairs, ids that were never issued: the stable-sort fallback), with and without file and header."""
    from ysmr_amd import _lib
    from ysmr_amd.helper_file import RowStream, rows_to_csv_file_and_dataframe
    cases = {"empty": np.zeros(0, _lib.ROW_DTYPE), "small": _tracker_like_rows(9, 5, 1), "large": _tracker_like_rows(160, 900, 2)}
    irregular = _tracker_like_rows(40, 60, 3)
    irregular = np.concatenate([irregular[::2], irregular[5:9], irregular[-1:]])      # gaps, duplicates
    irregular["track_id"][3] = 100000                                                    # an id far beyond the row count
    cases["irregular"] = irregular
    for name, rows in cases.items():
        order = np.lexsort((rows["frame"], rows["track_id"])) if name != "irregular" else np.argsort(
            rows["track_id"].astype(np.int64) << 32 | rows["frame"].astype(np.int64), kind="stable")
        for via in (True, False):
            want_path = tmp_path / f"want_{name}_{via}.csv"
            want_len, want_df = rows_to_csv_file_and_dataframe(rows[order], str(want_path), via_pandas=via)
            for pieces, header, with_file in ((1, True, True), (7, True, True), (7, False, True), (3, True, False)):
                st = RowStream(via_pandas=via, threads=3)
                for part in np.array_split(rows, pieces):
                    st.push(part)
                assert len(st) == len(rows)
                path = tmp_path / f"got_{name}_{via}_{pieces}_{header}.csv"
                path.write_bytes(b"stale " * 50)
                length, df = st.finish(str(path) if with_file else None, header=header)
                st.close()
                assert df.equals(want_df) and list(df.dtypes) == list(want_df.dtypes), (name, via, pieces)
                if with_file:
                    want = want_path.read_bytes()
                    if not header:
                        want = want[want.index(b"\n") + 1:]
                    assert length == len(want) and path.read_bytes() == want, (name, via, pieces, header)


def _devicelike(rows, header, via_pandas):
    import ctypes
    from ysmr_amd import _lib
    L = _lib.lib()
    n = len(rows)
    cap = int(L.ysmr_rows_csv_bound(n, int(header)))
    out = np.empty(cap, np.uint8)
    cols = np.empty((5, n), np.float64)
    length, bad = ctypes.c_size_t(0), ctypes.c_longlong(0)
    _lib.check(L.ysmr_rows_format_csv_devicelike(rows.ctypes.data, n, int(header), int(via_pandas), out.ctypes.data, cap,
                                                 ctypes.byref(length), cols.ctypes.data, ctypes.byref(bad)), "devicelike")
    return out[:length.value].tobytes(), cols, int(bad.value)


def test_device_formatter_arithmetic_equals_the_host_formatter():
    """csrc/fmt.h -- the shortest round-trip digits of Burger & Dybvig's free-format algorithm in 128-bit fixed point, CPython's
    layout, pandas' float converter: what ysmr_rows_format_device runs per thread -- on the HOST against ysmr_rows_format_csv
    (std::to_chars + the pandas restatement, itself pinned against pandas above): the same bytes and the same column values,
    on uniform values, values spread over the served exponents, float32 values widened to float64 (their exact expansions end
    in a 5: real ties, to the even digit), halves and tenths, negative values, zeros, the neighbours of every power of two;
    values outside 2^-20 .. 2^24, NaN and infinities are counted as unserved and their rows left out
    (helper_file.py:1403-1478, 860-905, 1366-1400)."""
    from ysmr_amd.helper_file import rows_to_csv_bytes, rows_to_dataframe
    rng = np.random.default_rng(12)
    n = 150000
    powers = np.ldexp(1.0, np.arange(-20, 24))
    pools = [rng.uniform(0, 1300, n), rng.uniform(0, 1, n) * 10.0 ** rng.integers(-5, 7, n),
             np.ldexp(rng.uniform(0.5, 1, n), rng.integers(-19, 24, n)), rng.uniform(0, 4000, n).astype(np.float32).astype(np.float64),
             rng.integers(0, 4000, n) + rng.choice([0, 0.5, 0.25, 0.125, 0.1, 0.3], n), -rng.uniform(0, 100, n),
             np.resize(np.concatenate([powers, np.nextafter(powers, 0), np.nextafter(powers, np.inf), 1.5 * powers, [0.0, -0.0]]), n)]
    for k, pool in enumerate(pools):
        rows = _random_rows(n, 20 + k)
        x = pool.copy()
        x[(np.abs(x) < 2.0 ** -20) & (x != 0)] = 1.0          # (the served range; the unserved cases are below)
        x[np.abs(x) >= 2.0 ** 24] = 2.0
        rows["x"], rows["y"] = x, x[::-1]
        for via_pandas in (False, True):
            text, cols, bad = _devicelike(rows, True, via_pandas)
            assert bad == 0
            assert text == rows_to_csv_bytes(rows, header=True, via_pandas=via_pandas, threads=4), (k, via_pandas)
            if via_pandas:
                df = rows_to_dataframe(rows)
                for j, name in enumerate(("POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE")):
                    np.testing.assert_array_equal(cols[j].view(np.uint64), df[name].to_numpy().view(np.uint64))
    rows = _random_rows(1000, 3)
    rows["x"][:6] = [np.nan, np.inf, -np.inf, 1e-9, 2.0 ** 24, 5e-324]
    text, _, bad = _devicelike(rows, False, True)
    assert bad == 6 and text == rows_to_csv_bytes(rows[6:], header=False, via_pandas=True)

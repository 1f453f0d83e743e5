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
    # y4m: luma plane only, 4:2:0
    y4 = tmp_path / "b.y4m"
    with open(y4, "wb") as fh:
        fh.write(b"YUV4MPEG2 W16 H12 F30000:1001 Ip A1:1 C420jpeg\n")
        for f in clip:
            fh.write(b"FRAME\n" + f.tobytes() + bytes(2 * 8 * 6))
    v = open_video(str(y4))
    assert (v.frame_count, v.height, v.width) == (7, 12, 16) and abs(v.fps - 29.97) < 1e-2
    np.testing.assert_array_equal(v.read(0, 7), clip)
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
    s2 = dict(s); s2["adaptive double threshold"] = -1.0
    s2["minimal frame count"] = 5
    assert track_bacteria(str(short), settings=s2, result_folder=str(tmp_path)) is None  # mean-gray branch: unsupported


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
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as fh:
        fh.write(f"{','.join(mine)};{elapsed}")
    dist.finish(info)


def test_stream_sharding_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = (tmp_path / "r0.txt").read_text().split(";")
    r1 = (tmp_path / "r1.txt").read_text().split(";")
    assert r0[0] == "video0,video2,video4" and r1[0] == "video1,video3"
    assert float(r0[1]) == float(r1[1]) == 2.0

"""GPU: the reference-shaped entry points (track_bacteria / analyse / ysmr) end to end vs the oracle."""
import os

import numpy as np
import pytest

from conftest import compare_rows

pytestmark = pytest.mark.gpu


def _settings(**kw):
    from ysmr_amd.helper_file import default_settings
    s = default_settings(**{"user input": False, "select files": False, "display video analysis": False,
                            "log to file": False, "minimal frame count": 40})
    s.update(kw)
    return s


def _rows_from_df(df):
    from ysmr_amd import _lib
    df = df.sort_values(["POSITION_T", "TRACK_ID"]).reset_index(drop=True)
    rows = np.zeros(len(df), _lib.ROW_DTYPE)
    rows["frame"], rows["track_id"] = df["POSITION_T"], df["TRACK_ID"]
    rows["x"], rows["y"] = df["POSITION_X"], df["POSITION_Y"]
    rows["w"], rows["h"], rows["angle"] = df["WIDTH"], df["HEIGHT"], df["DEGREES_ANGLE"]
    rows["disappeared"] = ((df["WIDTH"] == 0) & (df["HEIGHT"] == 0) & (df["DEGREES_ANGLE"] == 0)).astype(int)
    return rows


def test_track_bacteria_matches_oracle(tmp_path, oracle):
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(208, 272, 30, seed=21, dropout=0.04, speckle=0.04).frames(70)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16, max_det=256, capacity=256)
    assert res is not None
    df, fps, h, w, csv_path = res
    assert (fps, h, w) == (30.0, 208, 272) and os.path.basename(csv_path) == "clip_list.csv"
    assert list(df.columns) == ["TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"]
    assert df["TRACK_ID"].is_monotonic_increasing
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, shadows=2)
    got = _rows_from_df(df)
    # a matched 1-pixel blob also has w = h = angle = 0; use the oracle's notion of "disappeared"
    ref = np.array(ref_rows)
    compare_rows(got, ref_rows)
    assert open(csv_path).readline().strip() == "TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE"
    assert len(ref) == len(df)


def test_dark_on_bright_and_offset_sign_quirk(tmp_path, oracle):
    """THRESH_BINARY_INV path + the reference's in-place sign flip of the offset (track_eval.py:132)."""
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = 255 - SyntheticVideo(160, 200, 15, seed=5).frames(48)
    path = tmp_path / "dark.npy"
    np.save(path, frames)
    s = _settings(**{"white bacteria on dark background": False})
    res = track_bacteria(str(path), settings=s, result_folder=str(tmp_path), batch=16, max_det=512, capacity=512)
    assert res is not None and s["threshold offset for detection"] == -5     # mutated like upstream
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, white_on_dark=False, offset=5, adt=2.0, shadows=2)
    compare_rows(_rows_from_df(res[0]), ref_rows)


def test_analyse_and_ysmr(tmp_path):
    import json
    from ysmr_amd import analyse, ysmr
    from ysmr_amd.synth import SyntheticVideo
    paths = []
    for i in range(2):
        p = tmp_path / f"v{i}.npy"
        np.save(p, SyntheticVideo(120, 160, 8, seed=i).frames(45))
        paths.append(str(p))
    out = tmp_path / "results"
    from ysmr_amd.main import _OFFLINE_KEYS
    no_selection = {k: False for k in _OFFLINE_KEYS}   # (45 frames cannot hold a 20 s track: select_tracks has its own tests)
    df = analyse(paths[0], settings=_settings(**no_selection), result_folder=str(out), return_df=True, note="x")
    assert df is not None and len(df) > 0
    meta = json.load(open(out / "v0_meta.json"))
    assert meta["fps"] == 30.0 and meta["frame_height"] == 120 and meta["frame_width"] == 160 and meta["note"] == "x"
    done = ysmr(paths + [str(tmp_path / "nope.npy")], settings=_settings(**no_selection), result_folder=str(out))
    assert [p for p, _ in done] == paths + [str(tmp_path / "nope.npy")]
    assert done[0][1] is True and done[1][1] is True and done[2][1] is None
    assert (out / "v1_list.csv").exists()


def test_4k_dense_field_config(oracle):
    """BASELINE configs[4]: 3840x2160, ~5000 blobs.  Exercises large frames, max_det/capacity 8192
    (the two-kernel link path: the fused one needs its tables in LDS) and large-N assignment."""
    import torch
    from ysmr_amd import _lib
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import S4K
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    frames = S4K(seed=1).frames(3)
    p = threshold_params(True, 5, 2.0)
    det = Detector(3, 2160, 3840, max_det=8192, params=p)
    trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=8192, max_det=8192)
    rows = torch.empty(3 * 8192 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    res = det.detect(torch.from_numpy(frames).cuda())
    trk.run(res.det, res.det_count, 0, rows, count)
    torch.cuda.synchronize()
    assert int(res.status.max().item()) == 0 and trk.info()[2] == 0
    fd = oracle.detect_frame(frames[0], p.inv, p.t_low, p.t_high, p.use_high, 8192)
    assert 4500 < fd.count < 5200
    np.testing.assert_array_equal(res.labels[0].cpu().numpy(), fd.labels)
    np.testing.assert_array_equal(res.mask[0].cpu().numpy(), fd.mask)
    assert int(res.det_count[0].item()) == fd.count
    np.testing.assert_array_equal(res.det[0, :fd.count, :4].cpu().numpy(), fd.det[:, :4])
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, max_det=8192, shadows=2)
    compare_rows(rows_to_numpy(rows, int(count.item())), ref_rows)


def test_4k_dense_field_through_every_filter_and_a_tracks_whole_life(oracle):
    """BASELINE configs[4] in depth (the three-frame test above only sees the first filter): 3840x2160, ~5000 blobs,
    32 frames through the two-launch link (k_link + k_track + the detection grid) in batches of 16, as bench.py --config 4
    runs it, against the oracle's tracker with its conditioning probe.  fps = 10 keeps the oracle's Python loop to a
    couple of minutes and still covers a track's whole life: the second and third filters switch on at frames 10 and 20,
    dropped-out blobs (2 % per frame) are carried on their own predictions and deregistered after 10 lost frames, new ids
    are issued throughout, and the ring search around a lost track's prediction falls back to wider rings."""
    import torch
    from ysmr_amd import _lib
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import S4K
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    n_frames, batch, fps = 32, 16, 10.0
    frames = S4K(seed=2).frames(n_frames)
    p = threshold_params(True, 5, 2.0)
    det = Detector(batch, 2160, 3840, max_det=8192, params=p)
    trk = DeviceTracker(max_disappeared=fps, fps=fps, n_min=0, n_max=30, n_f=3, capacity=8192, max_det=8192)
    rows = torch.empty(n_frames * 8192 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    for f0 in range(0, n_frames, batch):
        res = det.detect(torch.from_numpy(frames[f0:f0 + batch]).cuda())
        trk.run(res.det, res.det_count, f0, rows, count)
        torch.cuda.synchronize()
        assert int(res.status.max().item()) == 0
    n_tracks, next_id, err = trk.info()
    assert err == 0
    got = rows_to_numpy(rows, int(count.item()))
    ref_rows, ref_trk = oracle.track_frames(frames, fps=fps, max_det=8192, shadows=2)
    ref = np.array([r[:7] for r in ref_rows])
    # the clip does what the test is for: tracks lost and carried, tracks dropped, ids issued after the first frame
    lost_rows = int((ref[:, 4] == 0).sum() - (ref[:, 4] == 0)[ref[:, 0] == 0].sum())
    first_ids = set(ref[ref[:, 0] == 0][:, 1].astype(int))
    last_ids = set(ref[ref[:, 0] == n_frames - 1][:, 1].astype(int))
    assert lost_rows > 1000 and len(first_ids - last_ids) > 50 and next_id > len(first_ids) + 100, (lost_rows, len(first_ids - last_ids), next_id)
    assert next_id == ref_trk.next_id and n_tracks == len(last_ids)
    from conftest import parity_report
    report = parity_report(got, ref_rows)
    _write_parity("4k_32_frames", report)
    compare_rows(got, ref_rows)
    assert report["worst_well_conditioned_relative"] <= 1e-9, report
    assert report["beyond_1e-5_with_shadows_within_1e-5"] == 0, report


def test_reader_protocol_reported_against_delivered_frames(tmp_path, oracle):
    """f1, the part of cv2.VideoCapture's contract that track_bacteria acts on (track_eval.py:73-93, 156-178, 402-404): the
    container REPORTS a frame count, cap.read() DELIVERS frames until it fails, and the two may differ.  AVI files whose
    stream header declares more / fewer frames than they store go through ``open_video`` + ``track_bacteria``; the outcome
    -- skipped, how many frames were tracked, error or not, None or a result, which fps -- must be what the oracle's
    restatement of the reference's control flow says for the same two numbers and settings."""
    from avi_tools import write_avi
    from ysmr_amd.frames import open_video
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    stored = 48
    clip = SyntheticVideo(96, 128, 12, seed=5).frames(stored)
    cases = [(stored, {}), (stored + 1, {}), (stored + 2, {}), (stored + 2, {"stop evaluation on error": False}),
             (stored - 1, {}), (stored - 6, {"stop evaluation on error": False}), (30, {"minimal frame count": 40}),
             (stored, {"force tracking.ini fps settings": True, "frames per second": 12.5}),
             (stored + 1, {"minimal frame count": stored + 1})]
    for k, (declared, extra) in enumerate(cases):
        path = tmp_path / f"r{k}.avi"
        write_avi(path, clip, 8, fps=(25, 1), declared=declared)
        video = open_video(str(path))
        assert (video.frame_count, video.frames_available, video.fps) == (declared, stored, 25.0)
        video.close()
        settings = _settings(**extra)
        want = oracle.reader_protocol(declared, stored, 25.0, settings)
        out = tmp_path / f"out{k}"
        out.mkdir()
        got = track_bacteria(str(path), settings=_settings(**extra), result_folder=str(out))
        assert (got is None) == want["returns_none"], (declared, extra, want)
        if want["skipped"]:
            assert not list(out.iterdir())                  # returned before anything was written (:73-77)
            continue
        csv = out / f"r{k}_list.csv"
        assert csv.exists()                                 # rows tracked before a read error stay on disk (:368-370)
        frames_in_csv = int(np.loadtxt(csv, delimiter=",", skiprows=1, usecols=1).max()) + 1
        assert frames_in_csv == want["frames_processed"] == stored
        if got is not None:
            df, fps, h, w, csv_path = got
            assert fps == want["fps"] and (h, w) == (96, 128) and int(df["POSITION_T"].max()) + 1 == stored


def test_resident_clip_needs_no_wait_behind_the_link():
    """TrackingPipeline.detect_async(frames_ready=...): a clip that is already in HBM does not wait for the link launches
    pending on the caller's stream (bench.py), an upload is waited for through its event alone; the rows are those of the
    blanket wait."""
    import torch
    from ysmr_amd.helper_file import default_settings
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import TrackingPipeline
    from ysmr_amd.tracker import rows_to_numpy
    n, b, h, w = 48, 8, 240, 320
    host = torch.from_numpy(SyntheticVideo(h, w, 30, seed=5).frames(n))
    resident = host.cuda()
    torch.cuda.synchronize()
    up = torch.cuda.Stream()

    def run(mode):
        pipe = TrackingPipeline(h, w, 30.0, default_settings(), batch=b, max_det=256, capacity=256, rows_per_flush=n * 256)
        pipe.reset()
        pending = None
        for f0 in list(range(0, n, b)) + [None]:
            nxt = None
            if f0 is not None:
                if mode == "event":
                    with torch.cuda.stream(up):
                        dev = host[f0:f0 + b].pin_memory().to("cuda", non_blocking=True)
                        ev = torch.cuda.Event(); ev.record(up)
                    dev.record_stream(pipe.side)
                    nxt = (pipe.detect_async(dev, frames_ready=ev), f0)
                else:
                    nxt = (pipe.detect_async(resident[f0:f0 + b], frames_ready=False if mode == "resident" else None), f0)
            if pending is not None:
                (slot, res, ready), p0 = pending
                pipe.link(slot, res, ready, p0)
            pending = nxt
        return pipe.take_rows()

    ref = run("blanket")
    assert len(ref) > 1000
    for mode in ("resident", "event"):
        got = run(mode)
        assert got.tobytes() == ref.tobytes(), mode


def test_rows_sort_on_device():
    """ysmr_rows_sort: (TRACK_ID, POSITION_T) order, the key of sort_list (helper_file.py:1538-1574)."""
    import torch
    from ysmr_amd import _lib
    from ysmr_amd.tracker import rows_to_numpy, sort_rows
    rng = np.random.default_rng(0)
    for n in (1, 7, 5000, 300001):
        rows = np.zeros(n, _lib.ROW_DTYPE)
        perm = rng.permutation(n)
        rows["track_id"], rows["frame"] = perm % 977, perm // 977          # unique pairs
        rows["x"] = rng.uniform(0, 1000, n)
        rows["w"] = rng.uniform(0, 10, n).astype(np.float32)
        dev = torch.from_numpy(rows.view(np.uint8)).cuda()
        got = rows_to_numpy(sort_rows(dev, n), n)
        ref = rows[np.lexsort((rows["frame"], rows["track_id"]))]
        assert got.tobytes() == ref.tobytes()
    # The tracker's own table needs no sort (row (id, f) belongs at offset[id] + f - first_frame[id]); tables
    # of any other shape are recognised on the device and go through the stable radix sort instead:
    # ids beyond the row count, tracks with gaps in their frames, and equal keys (input order kept)
    for n, shape in ((9000, "sparse ids"), (9000, "gaps"), (70001, "duplicates"), (4000, "a duplicate and a gap that cancel")):
        rows = np.zeros(n, _lib.ROW_DTYPE)
        perm = rng.permutation(n)
        if shape == "a duplicate and a gap that cancel":
            # every track's frames are gapless and unique except one: 0, 1, 1, 3 -- as many rows as last - first + 1
            rows["track_id"], rows["frame"] = perm % 40, perm // 40
            pick = np.nonzero(rows["track_id"] == 7)[0]
            pick = pick[np.argsort(rows["frame"][pick])]
            rows["frame"][pick[2]] = rows["frame"][pick[1]]
        elif shape == "sparse ids":
            rows["track_id"], rows["frame"] = (perm % 31) * 1_000_003 + 17, perm // 31
        elif shape == "gaps":
            rows["track_id"], rows["frame"] = perm % 31, (perm // 31) * 3 + (perm % 2)
        else:
            rows["track_id"], rows["frame"] = perm % 50, (perm // 50) % 40
        rows["x"] = np.arange(n)
        dev = torch.from_numpy(rows.view(np.uint8)).cuda()
        got = rows_to_numpy(sort_rows(dev, n), n)
        order = np.lexsort((np.arange(n), rows["frame"], rows["track_id"]))      # stable
        assert got.tobytes() == rows[order].tobytes(), shape


def test_track_bacteria_output_equals_the_reference_detour(tmp_path):
    """a19 + f2: the csv and the DataFrame of track_bacteria against what the reference's own
    sequence -- append Python-formatted rows, pandas.read_csv, sort_values, to_csv
    (track_eval.py:313-316, 340-346, 393; helper_file.py:1538-1574) -- makes of the same rows."""
    import pandas as pd
    import torch
    from ysmr_amd.helper_file import rows_to_csv_text, sort_list
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import TrackingPipeline, track_bacteria
    frames = SyntheticVideo(208, 272, 30, seed=23, dropout=0.05, speckle=0.05).frames(64)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16, max_det=256, capacity=256)
    assert res is not None
    df, _, _, _, csv_path = res
    # the same rows in emission order, then the reference's detour on the host
    pipe = TrackingPipeline(208, 272, 30.0, _settings(), batch=16, max_det=256, capacity=256, rows_per_flush=64 * 256)
    dev = torch.from_numpy(frames).cuda()
    for f0 in range(0, 64, 16):
        slot, r, ready = pipe.detect_async(dev[f0:f0 + 16])
        pipe.link(slot, r, ready, f0)
    rows = pipe.take_rows()
    ref_path = tmp_path / "ref_list.csv"
    with open(ref_path, "w", newline="") as fh:
        fh.write("TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE\n")
        fh.write(rows_to_csv_text(rows))
    df_ref = sort_list(file_path=str(ref_path), save_file=True)
    pd.testing.assert_frame_equal(df, df_ref, check_exact=True)
    assert open(csv_path, "rb").read() == ref_path.read_bytes()


def test_track_bacteria_bgr_file_and_ragged_last_batch(tmp_path, oracle):
    """Colour input (what cv2.VideoCapture delivers: a1 runs inside the strip kernel) with tinted
    blobs, a frame count that leaves a last batch of one frame, and a width that is a multiple of 4."""
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    gray = SyntheticVideo(160, 212, 20, seed=31, dropout=0.03, speckle=0.03).frames(49)   # 3 x 16 + 1
    rng = np.random.default_rng(2)
    tint = rng.uniform(0.6, 1.0, (1, 1, 1, 3))
    frames = np.clip(gray[..., None] * tint + rng.integers(0, 6, gray.shape + (3,)), 0, 255).astype(np.uint8)
    path = tmp_path / "colour.npy"
    np.save(path, frames)
    res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16, max_det=256, capacity=256)
    assert res is not None
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, shadows=2)
    assert len(ref_rows) == len(res[0])
    compare_rows(_rows_from_df(res[0]), ref_rows)


def test_track_bacteria_without_objects_returns_none(tmp_path, caplog):
    """track_eval.py:389-392: nothing tracked -> warning, None (and no exception)."""
    import logging
    from ysmr_amd.track_eval import track_bacteria
    frames = np.full((48, 64, 80), 40, np.uint8)
    path = tmp_path / "empty.npy"
    np.save(path, frames)
    with caplog.at_level(logging.WARNING, logger="ysmr"):
        assert track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16) is None
    assert any("Did not track any objects" in r.getMessage() for r in caplog.records)


def test_full_size_batch_properties():
    """BASELINE's metric configuration at full size (64 frames of 1228x922, ~500 blobs), checked through
    properties that need no oracle run: the label map against SciPy's 8-connected labelling of the
    mask (a label is its component's first pixel in raster order + 1), mask == (labels != 0),
    detections ordered by descending first pixel, a second batch through the SAME detector (sparse
    clear from the previous pixel list) giving exactly what a fresh detector gives, and rows that
    are complete and ordered per frame."""
    import torch
    from scipy import ndimage
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    from ysmr_amd import _lib
    B, H, W = 64, 922, 1228
    frames = SyntheticVideo(H, W, 500, seed=3).frames(2 * B)
    dev = torch.from_numpy(frames).cuda()
    p = threshold_params(True, 5, 2.0)
    det = Detector(B, H, W, max_det=2048, params=p)
    first = det.detect(dev[:B])
    labels_a = first.labels.cpu().numpy().copy()
    res = det.detect(dev[B:])                       # reuses the workspace: sparse clear of what batch 0 wrote
    labels, mask = res.labels.cpu().numpy(), res.mask.cpu().numpy()
    cnt, anchors, status = res.det_count.cpu().numpy(), res.anchors.cpu().numpy(), res.status.cpu().numpy()
    assert (status == 0).all() and (labels_a != 0).any()
    fresh = Detector(B, H, W, max_det=2048, params=p).detect(dev[B:])
    assert torch.equal(fresh.labels, res.labels) and torch.equal(fresh.mask, res.mask)
    assert torch.equal(fresh.det_count, res.det_count)
    valid = torch.arange(2048, device="cuda")[None, :] < res.det_count[:, None]          # rows past det_count are scratch
    assert torch.equal(fresh.det[valid], res.det[valid]) and torch.equal(fresh.anchors[valid], res.anchors[valid])
    np.testing.assert_array_equal(mask, (labels != 0).astype(np.uint8) * 255)
    flat_index = np.arange(H * W, dtype=np.int64).reshape(H, W)
    for f in (0, 31, 63):
        lab, n = ndimage.label(mask[f] != 0, structure=np.ones((3, 3)))
        first_px = ndimage.minimum(flat_index, lab, index=np.arange(1, n + 1)).astype(np.int64)
        want = np.zeros(n + 1, np.int64)
        want[1:] = first_px + 1
        np.testing.assert_array_equal(labels[f], want[lab])
        assert cnt[f] <= n and (np.diff(anchors[f][:cnt[f]]) < 0).all()       # reverse raster order, nested ones dropped
        assert set(anchors[f][:cnt[f]]) <= set(first_px)
    trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=2048, max_det=2048)
    rows = torch.empty(B * 2048 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    trk.run(res.det, res.det_count, 0, rows, count)
    torch.cuda.synchronize()
    n_tracks, next_id, err = trk.info()
    got = rows_to_numpy(rows, int(count.item()))
    assert err == 0 and len(got) > 0
    assert (np.diff(got["frame"]) >= 0).all()
    for f in range(B):
        ids = got["track_id"][got["frame"] == f]
        assert (np.diff(ids) > 0).all() and len(ids) >= cnt[f] - 0      # every detection of frame 0 became a track; ids ascend
        if f == 0:
            assert len(ids) == cnt[0]
    assert got["track_id"].max() == next_id - 1 and (got["frame"] == B - 1).sum() == n_tracks
    assert np.isfinite(got["x"]).all() and np.isfinite(got["y"]).all()


@pytest.mark.parametrize("white,offset", [(True, 5), (False, -10)])
def test_track_bacteria_mean_gray_branch(tmp_path, oracle, white, offset):
    """'adaptive double threshold' < 0: one level per frame from the 5 s moving average of
    mean +- stddev +- offset (track_eval.py:219-253); the list runs across the batches of the file.
    (Dark on bright: the reference negates the offset first, track_eval.py:132, so a POSITIVE setting
    moves the level towards the background; -10 keeps it 10 below mean - stddev.)"""
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(160, 208, 20, seed=9, dropout=0.03).frames(60)
    drift = (10 * np.sin(np.arange(60) / 6.0)).astype(np.int32)[:, None, None]
    frames = (frames.astype(np.int32) + drift).clip(0, 255).astype(np.uint8)
    if not white:
        frames = 255 - frames
    path = tmp_path / "level.npy"
    np.save(path, frames)
    with open(tmp_path / "level_meta.json", "w") as fh:
        fh.write('{"fps": 4.0}')
    s = _settings(**{"adaptive double threshold": -1.0, "white bacteria on dark background": white,
                     "threshold offset for detection": offset})
    res = track_bacteria(str(path), settings=s, result_folder=str(tmp_path), batch=16, max_det=512, capacity=512)
    assert res is not None and res[1] == 4.0
    ref_rows, _ = oracle.track_frames(frames, fps=4.0, white_on_dark=white, offset=offset, adt=-1.0, shadows=2)
    assert len(ref_rows) > 500
    compare_rows(_rows_from_df(res[0]), ref_rows)


def test_device_frame_feed_order_and_reuse(tmp_path):
    """DeviceFrameFeed: every batch arrives once, in order, with the file's bytes -- also when its two
    device buffers are reused many times and the consumer is slow to release them."""
    import time
    import torch
    from ysmr_amd.frames import DeviceFrameFeed, open_video
    rng = np.random.default_rng(8)
    for shape in [(75, 40, 52), (21, 16, 24, 3)]:
        clip = rng.integers(0, 256, shape, dtype=np.uint8)
        path = tmp_path / f"feed{len(shape)}.npy"
        np.save(path, clip)
        video = open_video(str(path))
        feed = DeviceFrameFeed(video, 8, "cuda:0", depth=2, readers=3)
        seen = 0
        for k, (dev, f0, n, slot) in enumerate(feed):
            assert f0 == seen and n == min(8, shape[0] - f0) and dev.shape[0] == n
            if k % 3 == 0:
                time.sleep(0.01)                       # let the producer run into the held slot
            host = dev.cpu().numpy()                   # (synchronises: the upload is complete)
            np.testing.assert_array_equal(host, clip[f0:f0 + n])
            done = torch.cuda.Event()
            done.record()
            feed.release(slot, done)
            seen += n
        assert seen == shape[0]
        feed.close()
        video.close()


def test_ysmr_worker_processes_started_from_a_gpu_free_parent(tmp_path):
    """The branch of ysmr(multiprocess=True) a multi-GPU node takes -- one spawned worker process per GPU, started by
    a parent that has not touched the GPU -- run for real on this one-GPU box: a fresh interpreter is told there are
    two devices, deals four videos to 'cuda:0' and 'cuda:1', and its two workers (which see one device and fold the
    ordinal) analyse two videos each, two streams at a time."""
    import subprocess
    import sys
    from ysmr_amd.synth import SyntheticVideo
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paths = []
    for i in range(4):
        p = tmp_path / f"w{i}.npy"
        np.save(p, SyntheticVideo(96, 128, 6, seed=20 + i).frames(44))
        paths.append(str(p))
    out = tmp_path / "res"
    script = f"""
import sys, json
sys.path.insert(0, {root!r})
import torch
torch.cuda.device_count = lambda: 2            # (device_count does not initialise the GPU; nothing else is touched)
from ysmr_amd import ysmr
from ysmr_amd.helper_file import default_settings
from ysmr_amd.main import _OFFLINE_KEYS
s = default_settings(**{{"user input": False, "select files": False, "display video analysis": False, "log to file": False,
                        "minimal frame count": 40, **{{k: False for k in _OFFLINE_KEYS}}}})
done = ysmr({paths!r}, settings=s, result_folder={str(out)!r}, multiprocess=True, streams_per_gpu=2)
assert not torch.cuda.is_initialized()          # the parent never touched the GPU
print("RESULT " + json.dumps([[p, r] for p, r in done]))
"""
    proc = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    import json
    done = json.loads(line[len("RESULT "):])
    assert [p for p, _ in done] == paths and all(r is True for _, r in done)
    assert "running the" not in proc.stderr            # not the in-process fallback: real worker processes
    for i in range(4):
        text = (out / f"w{i}_list.csv").read_text().splitlines()
        assert text[0] == "TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE" and len(text) > 100


def test_uncompressed_avi_is_unpacked_on_the_device(tmp_path):
    """f1: DeviceFrameFeed uploads the stored DIB frames of an uncompressed AVI as they are (bottom-up or top-down,
    rows padded to 4 bytes, 8-bit gray / palette indices / 24-bit BGR, dropped frames, an OpenDML continuation) and
    ysmr_unpack_dib_batch turns them into frames on the device: the same bytes as the host reader delivers."""
    import ctypes
    import torch
    from avi_tools import write_avi
    from ysmr_amd import _lib
    from ysmr_amd.frames import DeviceFrameFeed, open_video
    rng = np.random.default_rng(21)
    palette = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    cases = [dict(shape=(19, 33, 50), bits=8), dict(shape=(11, 20, 37), bits=8, top_down=True, dropped=(3, 4)),
             dict(shape=(13, 18, 21, 3), bits=24, split=6), dict(shape=(9, 25, 31), bits=8, palette=palette),
             dict(shape=(5, 1, 1), bits=8), dict(shape=(7, 9, 3, 3), bits=24, top_down=True)]
    for i, c in enumerate(cases):
        clip = rng.integers(0, 256, c.pop("shape"), dtype=np.uint8)
        path = tmp_path / f"u{i}.avi"
        # what cap.read() owes for this clip, built from the clip itself and not by the product's host reader: a dropped
        # (empty) chunk repeats the frame before it, palette indices become the palette's B, G, R, everything else is the
        # array that went into the file
        want = clip.copy()
        for d in c.get("dropped", ()):
            want[d] = want[d - 1]
        if c.get("palette") is not None:
            want = c["palette"][want]
        write_avi(path, clip, c.pop("bits"), **c)
        video = open_video(str(path))
        assert video.raw_layout is not None
        np.testing.assert_array_equal(video.read(0, video.frame_count), want)   # (the host reader owes the same)
        feed = DeviceFrameFeed(video, 4, "cuda:0", depth=2, readers=2)
        assert feed._raw is not None
        got = []
        for dev, f0, n, slot in feed:
            got.append(dev.cpu().numpy().copy())
            done = torch.cuda.Event(); done.record()
            feed.release(slot, done)
        feed.close(); video.close()
        np.testing.assert_array_equal(np.concatenate(got), want)
    # argument checks of the entry point
    L = _lib.lib()
    buf = torch.zeros(64, dtype=torch.uint8, device="cuda")
    for args in [(0, 16, 2, 2, 1, 4, 1), (1, 16, 2, 2, 2, 4, 1), (1, 16, 2, 5, 1, 4, 1), (1, 4, 2, 2, 1, 4, 1)]:
        n, raw_bytes, h, w, bpp, stride, up = args
        assert L.ysmr_unpack_dib_batch(None, buf.data_ptr(), n, raw_bytes, h, w, bpp, stride, up, None, buf.data_ptr()) == _lib.YSMR_ERR_ARG


def test_ysmr_multiprocess_two_streams_per_gpu(tmp_path):
    """ysmr(multiprocess=True): single-use worker processes, two at a time on the one GPU of the box."""
    from ysmr_amd import ysmr
    from ysmr_amd.main import _OFFLINE_KEYS
    from ysmr_amd.synth import SyntheticVideo
    paths = []
    for i in range(3):
        p = tmp_path / f"m{i}.npy"
        np.save(p, SyntheticVideo(96, 128, 6, seed=10 + i).frames(44))
        paths.append(str(p))
    out = tmp_path / "res"
    done = ysmr(paths, settings=_settings(**{k: False for k in _OFFLINE_KEYS}), result_folder=str(out),
                multiprocess=True, streams_per_gpu=2)
    assert [p for p, _ in done] == paths and all(r is True for _, r in done)
    for i in range(3):
        text = (out / f"m{i}_list.csv").read_text().splitlines()
        assert text[0] == "TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE" and len(text) > 100


def test_track_bacteria_reads_uncompressed_avi(tmp_path):
    """The same clip as .npy and as a raw 8-bit AVI (bottom-up, padded rows) gives the same table."""
    from avi_tools import write_avi
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(118, 150, 9, seed=6).frames(50)              # width 150: DIB rows padded to 152
    np.save(tmp_path / "a.npy", frames)
    write_avi(tmp_path / "b.avi", frames, 8, fps=(30, 1))
    os.makedirs(tmp_path / "ra"); os.makedirs(tmp_path / "rb")
    ra = track_bacteria(str(tmp_path / "a.npy"), settings=_settings(), result_folder=str(tmp_path / "ra"), batch=16)
    rb = track_bacteria(str(tmp_path / "b.avi"), settings=_settings(), result_folder=str(tmp_path / "rb"), batch=16)
    assert ra is not None and rb is not None and rb[1:4] == (30.0, 118, 150)
    assert ra[0].equals(rb[0]) and len(ra[0]) > 200
    assert open(ra[4], "rb").read() == open(rb[4], "rb").read()
    # 24-bit BGR frames (top-down here): unpacked on the device, a1 (BGR2GRAY) inside the threshold kernel
    rng = np.random.default_rng(4)
    tint = np.clip(frames[..., None] * rng.uniform(0.6, 1.0, (1, 1, 1, 3)), 0, 255).astype(np.uint8)
    np.save(tmp_path / "c.npy", tint)
    write_avi(tmp_path / "d.avi", tint, 24, fps=(30, 1), top_down=True)
    os.makedirs(tmp_path / "rc"); os.makedirs(tmp_path / "rd")
    rc = track_bacteria(str(tmp_path / "c.npy"), settings=_settings(), result_folder=str(tmp_path / "rc"), batch=16)
    rd = track_bacteria(str(tmp_path / "d.avi"), settings=_settings(), result_folder=str(tmp_path / "rd"), batch=16)
    assert rc is not None and rd is not None and rc[0].equals(rd[0]) and len(rc[0]) > 200
    assert open(rc[4], "rb").read() == open(rd[4], "rb").read()


def test_track_bacteria_row_buffer_smaller_than_the_video(tmp_path, monkeypatch):
    """When the rows of a video do not fit the device buffer they are moved to the host in between and
    ordered in one go at the end: same DataFrame, same csv as the all-on-device run."""
    from ysmr_amd import track_eval
    from ysmr_amd.synth import SyntheticVideo
    frames = SyntheticVideo(120, 160, 18, seed=12, dropout=0.02).frames(90)
    np.save(tmp_path / "v.npy", frames)
    os.makedirs(tmp_path / "whole"); os.makedirs(tmp_path / "pieces")
    kw = dict(batch=16, max_det=64, capacity=64)
    whole = track_eval.track_bacteria(str(tmp_path / "v.npy"), settings=_settings(), result_folder=str(tmp_path / "whole"), **kw)
    monkeypatch.setattr(track_eval, "ROW_BUDGET_MAX", 1)
    small = _settings(**{"list save length interval": 100})       # buffer = 2 batches x 64 rows per frame
    pieces = track_eval.track_bacteria(str(tmp_path / "v.npy"), settings=small, result_folder=str(tmp_path / "pieces"), **kw)
    assert whole is not None and pieces is not None and len(whole[0]) > 1200
    assert whole[0].equals(pieces[0])
    assert open(whole[4], "rb").read() == open(pieces[4], "rb").read()


def test_rows_can_be_persisted_while_the_video_runs(tmp_path, monkeypatch):
    """'hip persist rows': full row buffers are appended to <name>_list.csv during the run, as the reference does every
    'list save length interval' rows (an interrupted run leaves what was tracked so far); the final table and file
    are those of the default run."""
    from ysmr_amd import track_eval
    from ysmr_amd.synth import SyntheticVideo
    frames = SyntheticVideo(120, 160, 18, seed=12, dropout=0.02).frames(90)
    np.save(tmp_path / "v.npy", frames)
    os.makedirs(tmp_path / "whole"); os.makedirs(tmp_path / "kept")
    kw = dict(batch=16, max_det=64, capacity=64)
    whole = track_eval.track_bacteria(str(tmp_path / "v.npy"), settings=_settings(), result_folder=str(tmp_path / "whole"), **kw)
    seen = []
    original = track_eval._persist_chunk

    def spy(list_name, rows, first):
        original(list_name, rows, first)
        text = open(list_name).read().splitlines()
        seen.append((first, len(rows), len(text)))
        assert text[0] == "TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE"
        t = np.array([int(line.split(",")[1]) for line in text[1:]])
        assert (np.diff(t) >= 0).all()                       # in the order they were tracked: by frame

    monkeypatch.setattr(track_eval, "_persist_chunk", spy)
    s = _settings(**{"hip persist rows": True, "list save length interval": 100})     # buffer = 2 batches x 64 rows per frame
    kept = track_eval.track_bacteria(str(tmp_path / "v.npy"), settings=s, result_folder=str(tmp_path / "kept"), **kw)
    assert whole is not None and kept is not None
    assert len(seen) >= 2 and seen[0][0] and not seen[1][0]
    assert all(lines == 1 + sum(n for _, n, _ in seen[:i + 1]) for i, (_, _, lines) in enumerate(seen))
    assert whole[0].equals(kept[0])
    assert open(whole[4], "rb").read() == open(kept[4], "rb").read()


from conftest import ROUND  # noqa: E402


def _write_parity(name, report):
    """The round's parity record: under gpurun_out/ (what comes back from the GPU box) and, where the tree is
    writable, under profiles/ with the round in the name -- DESIGN.md section 2 quotes that file."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for folder in ("gpurun_out", "profiles"):
        try:
            os.makedirs(os.path.join(root, folder), exist_ok=True)
            with open(os.path.join(root, folder, f"{ROUND}_parity_{name}.json"), "w") as fh:
                json.dump(report, fh, indent=1)
        except OSError:
            pass


def test_bench_config_rows_hold_1e9_outside_the_oracles_ill_conditioned_set(tmp_path, oracle):
    """BASELINE configs[2] at full size (1228x922, ~500 blobs, 200 frames) through track_bacteria: every
    row the reference's arithmetic determines is within 1e-9 of the oracle; the rows it does not determine
    (oracle shadow filters, conftest.compare_rows) are a small, counted fraction, each within conftest.AMPLIFICATION
    x what one ulp does to the reference itself, and every row beyond north_star's 1e-5 relative is one whose own
    +-1-ulp shadows are more than 1e-5 apart (the claim "the reference does not determine this row to 1e-5", tested).
    The numbers go to profiles/<ROUND>_parity_bench_config.json; DESIGN.md section 2 quotes them, and tests/test_host.py::
    test_design_quotes_the_parity_record_it_names fails when the quote and the tracked file differ."""
    from conftest import parity_report
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(922, 1228, 500, seed=0, fps=30.0).frames(200)
    path = tmp_path / "bench.npy"
    np.save(path, frames)
    res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path))
    assert res is not None
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, shadows=2)
    got = _rows_from_df(res[0])
    report = parity_report(got, ref_rows)
    _write_parity("bench_config", report)
    ref = np.array(ref_rows)
    marked = ref[:, 7] > oracle.OracleTracker.ILL_CONDITIONED       # the marked rows themselves, for tests/tools/parity_rows.py
    np.savez(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", ROUND + "_parity_bench_rows.npz"),
             ref=ref[marked], x=got["x"][marked], y=got["y"][marked], gone=got["disappeared"][marked])
    n_loose, worst = compare_rows(got, ref_rows)
    assert report["rows_of_lost_tracks"] > 5000             # the clip does lose tracks (2 % dropout)
    assert n_loose == report["ill_conditioned_rows"] < 0.03 * len(got), report
    assert report["worst_well_conditioned_relative"] <= 1e-9, report
    assert report["beyond_1e-5_with_shadows_within_1e-5"] == 0, report


def test_config0_full_size_through_ysmr_with_the_default_settings(tmp_path, oracle):
    """BASELINE configs[0] as it stands: one 1228x922 video at 30 fps, ~50 bacteria, through ysmr() with the
    default tracking.ini (only the interactive switches off) -- so with the default 600-frame minimum, the default
    selection and statistics stages -- and the table compared with the oracle row for row."""
    import pandas as pd
    from ysmr_amd import ysmr
    from ysmr_amd.helper_file import default_settings, get_data
    from ysmr_amd.synth import SyntheticVideo
    frames = SyntheticVideo(922, 1228, 50, seed=4, fps=30.0).frames(630)
    path = tmp_path / "config0.npy"
    np.save(path, frames)
    s = default_settings(**{"user input": False, "select files": False, "display video analysis": False, "log to file": False})
    assert s["minimal frame count"] == 600
    out = tmp_path / "res"
    done = ysmr([str(path)], settings=s, result_folder=str(out))
    assert done is not None and done[0][0] == str(path) and done[0][1] is True
    table = get_data(str(out / "config0_list.csv"))
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, shadows=2)
    n_loose, _ = compare_rows(_rows_from_df(table), ref_rows)
    assert len(table) == len(ref_rows) > 25000 and n_loose < 0.01 * len(table)
    # the later stages ran on it: the selection's and the statistics' files are the oracle's
    selected, info = oracle.select_tracks_oracle(table, s, 30.0, 922, 1228)
    assert selected is not None and info["good_tracks"] >= 5
    # (the table was parsed back from its csv here, a second pass through pandas' float parser: same rows, values
    # equal to the last digit or two -- the byte-for-byte comparison of this file is test_gpu_select.py's)
    ours = pd.read_csv(out / "config0_selected_data.csv")
    assert list(ours.columns) == list(selected.columns) and len(ours) == len(selected)
    for c in ("index", "TRACK_ID", "POSITION_T"):
        np.testing.assert_array_equal(ours[c].to_numpy(), selected[c].to_numpy())
    for c in ("POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"):
        np.testing.assert_allclose(ours[c].to_numpy(), selected[c].to_numpy(), rtol=1e-12, atol=0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ref_stats = oracle.evaluate_tracks_oracle(selected, s, 30.0)
    stats = pd.read_csv(out / "config0_statistics.csv")
    assert len(stats) == len(ref_stats) == info["good_tracks"]


def test_video_denser_than_the_buffers_is_run_again_with_larger_ones(tmp_path, oracle, caplog):
    """The reference has no limit on objects per frame; a video that overflows max_det / capacity is found
    out after its first batch and run again with both doubled -- same table as with ample buffers."""
    import logging
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(200, 260, 40, seed=3).frames(48)
    path = tmp_path / "dense.npy"
    np.save(path, frames)
    with caplog.at_level(logging.WARNING, logger="ysmr"):
        res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16, max_det=16, capacity=16)
    assert res is not None
    assert sum("running it again" in r.getMessage() for r in caplog.records) == 2      # 16 -> 32 -> 64
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, shadows=2)
    compare_rows(_rows_from_df(res[0]), ref_rows)
    # the limits may also come from the settings dict; analyse() passes its own arguments through
    from ysmr_amd import analyse
    from ysmr_amd.main import _OFFLINE_KEYS
    s = _settings(**{k: False for k in _OFFLINE_KEYS}, **{"hip max detections per frame": 128, "hip max tracks": 128,
                                                          "hip frames per batch": 8})
    df = analyse(str(path), settings=s, result_folder=str(tmp_path / "b"), return_df=True)
    df2 = analyse(str(path), settings=_settings(**{k: False for k in _OFFLINE_KEYS}), result_folder=str(tmp_path / "c"),
                  return_df=True, batch=24, max_det=64, capacity=96)
    assert df is not None and df.equals(res[0]) and df2.equals(res[0])


def test_video_that_outgrows_the_batch_link_is_linked_per_frame_with_the_oracles_rows(tmp_path, oracle, caplog):
    """More objects than the 768 seats of the one-launch-per-batch link: ~820 blobs at 1228 x 922
    overflow capacity 768 in the first batch; track_bacteria runs the file again with capacity 1536 -- a handle the batch
    link does not serve, i.e. on k_frame, one launch per frame -- and the table must be the oracle's."""
    import logging
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import LAST_PIPELINE_FACTS, track_bacteria
    frames = SyntheticVideo(922, 1228, 820, seed=13).frames(40)
    path = tmp_path / "crowd.npy"
    np.save(path, frames)
    with caplog.at_level(logging.WARNING, logger="ysmr"):
        res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16, max_det=1024, capacity=768)
    assert res is not None
    assert sum("running it again" in r.getMessage() for r in caplog.records) == 1
    assert LAST_PIPELINE_FACTS == {"capacity": 1536, "max_det": 2048, "batched": False, "fused": True}
    ref_rows, tr = oracle.track_frames(frames, fps=30.0, shadows=2)
    assert max(np.bincount(np.array([r[0] for r in ref_rows], dtype=int))) > 768
    compare_rows(_rows_from_df(res[0]), ref_rows)


def test_rows_printed_while_the_video_runs_leave_the_same_file_and_table(tmp_path):
    """'hip stream rows' = True (``_RowDrain`` + ``ysmr_rows_stream_*``: every batch's rows leave the device behind its link
    launch and are printed while later batches run; their order is worked out at the end) against the default, serial
    tail: the same csv byte for byte, the same DataFrame -- also when the device row buffer is small enough to be started
    over in mid-video."""
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(300, 400, 60, seed=17).frames(96)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    for name in "abc":
        (tmp_path / name).mkdir()
    want = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path / "a"), batch=16)
    assert want is not None
    want_bytes = open(want[4], "rb").read()
    for name, extra in (("b", {}), ("c", {"list save length interval": 1})):      # (c: the smallest row buffer, 2 batches' worth)
        got = track_bacteria(str(path), settings=_settings(**{"hip stream rows": True}, **extra), result_folder=str(tmp_path / name),
                             batch=16, capacity=128)
        assert got is not None and got[0].equals(want[0]) and list(got[0].dtypes) == list(want[0].dtypes)
        assert open(got[4], "rb").read() == want_bytes


def test_rows_printed_on_the_device_leave_the_same_file_and_table(tmp_path):
    """Round 5: ``track_bacteria`` prints its ordered rows on the device (``ysmr_rows_format_device``: csv text and the
    DataFrame's columns, csrc/fmt.h) -- against the host path ('hip print rows on device' = False, ``ysmr_rows_write_csv_columns``):
    the same csv byte for byte, the same DataFrame bit for bit (helper_file.py:1403-1478, 860-905, 1366-1400)."""
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(300, 400, 60, seed=18).frames(96)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    for name in "ab":
        (tmp_path / name).mkdir()
    want = track_bacteria(str(path), settings=_settings(**{"hip print rows on device": False}), result_folder=str(tmp_path / "a"), batch=16)
    got = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path / "b"), batch=16)
    assert want is not None and got is not None
    assert got[0].equals(want[0]) and list(got[0].dtypes) == list(want[0].dtypes) and len(got[0]) > 3000
    for c in ("POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"):
        np.testing.assert_array_equal(got[0][c].to_numpy().view(np.uint64), want[0][c].to_numpy().view(np.uint64))
    assert open(got[4], "rb").read() == open(want[4], "rb").read()


def test_device_formatter_against_the_host_formatter_on_a_large_table(tmp_path):
    """``ysmr_rows_format_device`` on 400 000 rows of mixed values (uniform, float32-widened: ties, halves, negative, zeros, the
    neighbours of powers of two) against ``ysmr_rows_write_csv_columns``: file and columns identical; a table with a NaN is
    declined (``None``: the caller takes the host path), an empty table is its header."""
    import torch
    from ysmr_amd import _lib
    from ysmr_amd.helper_file import rows_device_to_csv_file_and_dataframe, rows_to_csv_file_and_dataframe
    rng = np.random.default_rng(33)
    n = 400000
    rows = np.zeros(n, _lib.ROW_DTYPE)
    rows["track_id"] = np.arange(n) // 200
    rows["frame"] = np.arange(n) % 200
    powers = np.ldexp(1.0, np.arange(-20, 24))
    edge = np.concatenate([powers, np.nextafter(powers, 0)[1:], np.nextafter(powers, np.inf), [0.0, -0.0, 0.1, 0.5, 1e-5, 123456.789]])
    rows["x"] = np.where(rng.random(n) < 0.5, rng.uniform(-5, 1300, n), rng.uniform(0, 4000, n).astype(np.float32).astype(np.float64))
    rows["x"][:len(edge)] = edge
    rows["y"] = rng.integers(0, 4000, n) + rng.choice([0, 0.5, 0.25, 0.125, 0.1, 0.3], n)
    rows["w"], rows["h"] = rng.uniform(0, 40, n).astype(np.float32), rng.uniform(0, 40, n).astype(np.float32)
    rows["angle"] = rng.uniform(-90, 90, n).astype(np.float32)
    dev = torch.from_numpy(rows.view(np.uint8).copy()).cuda()
    for via in (True, False):
        a, b = tmp_path / "host.csv", tmp_path / "device.csv"
        len_h, df_h = rows_to_csv_file_and_dataframe(rows, str(a), via_pandas=via)
        made = rows_device_to_csv_file_and_dataframe(dev, n, str(b), via_pandas=via)
        assert made is not None and made[0] == len_h
        assert a.read_bytes() == b.read_bytes()
        assert made[1].equals(df_h) and list(made[1].dtypes) == list(df_h.dtypes)
        for c in ("POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"):
            np.testing.assert_array_equal(made[1][c].to_numpy().view(np.uint64), df_h[c].to_numpy().view(np.uint64))
    rows["y"][12345] = np.nan
    assert rows_device_to_csv_file_and_dataframe(torch.from_numpy(rows.view(np.uint8).copy()).cuda(), n, None) is None
    empty = rows_device_to_csv_file_and_dataframe(dev, 0, str(tmp_path / "empty.csv"))
    assert empty is not None and len(empty[1]) == 0
    assert (tmp_path / "empty.csv").read_bytes() == b"TRACK_ID,POSITION_T,POSITION_X,POSITION_Y,WIDTH,HEIGHT,DEGREES_ANGLE\n"


def test_no_live_track_in_the_last_frame_means_nothing_tracked(tmp_path, caplog):
    """track_eval.py:387-392: the reference asks the LAST frame's tracker output for its last object id, so a
    video that ends on more than a second of empty frames 'did not track any objects' (the list is on disk)."""
    import logging
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(120, 160, 8, seed=4).frames(90)
    frames[45:] = 40                                   # nothing to see for 45 frames; tracks linger for 30
    path = tmp_path / "fade.npy"
    np.save(path, frames)
    with caplog.at_level(logging.WARNING, logger="ysmr"):
        res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16)
    assert res is None and any("Did not track any objects" in r.getMessage() for r in caplog.records)
    assert len((tmp_path / "fade_list.csv").read_text().splitlines()) > 100
    frames[80:] = SyntheticVideo(120, 160, 8, seed=4).frames(10)      # something alive at the end again
    np.save(path, frames)
    assert track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16) is not None


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs")
def test_track_bacteria_on_the_second_gpu_while_the_first_is_current(tmp_path, oracle):
    """Every launch goes to the stream of the device its buffers live on, whatever the caller's current
    device is (a fresh worker's is cuda:0)."""
    import torch
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.track_eval import track_bacteria
    frames = SyntheticVideo(160, 200, 20, seed=8).frames(48)
    path = tmp_path / "second.npy"
    np.save(path, frames)
    torch.cuda.set_device(0)
    res = track_bacteria(str(path), settings=_settings(), result_folder=str(tmp_path), batch=16, device="cuda:1")
    assert res is not None and torch.cuda.current_device() == 0
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, shadows=2)
    compare_rows(_rows_from_df(res[0]), ref_rows)

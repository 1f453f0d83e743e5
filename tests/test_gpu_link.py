"""GPU parity (run with -m gpu): HIP tracker/GSFF vs (a) fixtures captured from the reference's
own tracker.py/gsff.py and (b) the CPU oracle.

Integer quantities (ids, claims, counters, disappeared) are exact.  GSFF-smoothed positions:
north_star allows 1e-5 relative; the device differs from NumPy only through summation order of the
FIR dot products and exp(), so the tests hold it to 1e-9.
"""
import numpy as np
import pytest

from conftest import compare_rows, golden, rects_of, tracker_frames

pytestmark = pytest.mark.gpu
RTOL = 1e-9
ATOL = 1e-9   # pixels; positions that are exactly 0 in exact arithmetic come out as rounding noise


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


TRACKER_CASES = [("tracker_small_gsff.npz", None), ("tracker_small_nogsff.npz", None), ("tracker_mid_gsff.npz", None),
                 ("tracker_gap.npz", 5), ("tracker_2997.npz", 6)]


@pytest.mark.parametrize("name,max_gone", TRACKER_CASES)
def test_centroid_tracker_matches_reference_fixture(torch_cuda, name, max_gone):
    from ysmr_amd.tracker import CentroidTracker
    g = golden(name)
    fps = float(g["fps"])
    n_max = None if int(g["n_max"]) < 0 else int(g["n_max"])
    ct = CentroidTracker(max_disappeared=fps if max_gone is None else max_gone, fps=fps, n_min=int(g["n_min"]),
                         n_max=n_max, n_f=int(g["n_f"]), use_gsff=bool(g["use_gsff"]), capacity=1024, max_det=1024)
    off, coff = g["off"], g["claim_off"]
    for f, (det, info) in enumerate(tracker_frames(g)):
        objs, infos = ct.update(rects_of(det, info))
        sl = slice(off[f], off[f + 1])
        assert list(objs.keys()) == list(g["ids"][sl]), f"ids frame {f}"
        np.testing.assert_allclose(np.array(list(objs.values())).reshape(-1, 2), g["xy"][sl], rtol=RTOL, atol=ATOL,
                                   err_msg=f"xy frame {f}")
        got_info = np.array([list(infos[i]) for i in objs.keys()], dtype=float).reshape(-1, 3)
        np.testing.assert_array_equal(got_info, g["info"][sl], err_msg=f"info frame {f}")
        assert list(ct.disappeared.values()) == list(g["disappeared"][sl]), f"disappeared frame {f}"
        assert ct.nextObjectID == g["next_id"][f]
        assert sorted(ct.last_claims) == sorted(map(tuple, g["claims"][coff[f]:coff[f + 1]].tolist())), f"claims {f}"


GSFF_CASES = [("gsff_smooth_default.npz", 30.0, 0, 30, 3), ("gsff_jump_default.npz", 30.0, 0, 30, 3),
              ("gsff_turn_default.npz", 30.0, 0, 30, 3), ("gsff_lost_default.npz", 30.0, 0, 30, 3),
              ("gsff_smooth_2997.npz", 29.97, 0, 29.97, 3), ("gsff_jump_nf4.npz", 25.0, 4, 24, 4)]


@pytest.mark.parametrize("name,fps,n_min,n_max,n_f", GSFF_CASES)
def test_gsff_class_matches_reference_fixture(torch_cuda, name, fps, n_min, n_max, n_f):
    from ysmr_amd.gsff import GaussianSumFIR
    g = golden(name)
    f = GaussianSumFIR(delta_t=1 / fps, n_min=n_min, n_max=n_max, n_f=n_f)
    assert f.n_i == list(g["n_i"])
    state = {}
    for k, z in enumerate(g["fed"]):
        c, state = f.correct(measurement=np.array(z), **state)
        p, state = f.predict(**state)
        assert state["mode"] == g["mode"][k]
        np.testing.assert_allclose(c, g["correct"][k], rtol=RTOL, atol=ATOL, err_msg=f"correct step {k}")
        np.testing.assert_allclose(p, g["predict"][k], rtol=RTOL, atol=ATOL, err_msg=f"predict step {k}")


@pytest.mark.parametrize("cap,trials,n0_max,extra_max", [
    (1024, 25, 120, 200),      # k_frame: the model in its 16-bit LDS tables
    (4096, 10, 1500, 2500),    # k_link, tables in LDS: the model takes over the dead claim tables (up to 16384 slots)
    (8192, 6, 3000, 5100),     # ... up to 32768 slots (the 4K configuration's capacity)
    (16384, 3, 1500, 2500),    # k_link, tables in HBM: the one-thread model
])
def test_new_ids_follow_cpython_set_order(torch_cuda, oracle, cap, trials, n0_max, extra_max):
    """tracker.py:216 iterates a set: ids of tracks born in one frame follow CPython's hash-table
    order.  The device model must reproduce it for every mix of claimed/unclaimed columns."""
    from ysmr_amd.tracker import CentroidTracker
    rng = np.random.default_rng(11)
    for trial in range(trials):
        n0 = int(rng.integers(1, n0_max))
        extra = int(rng.integers(1, extra_max))
        if trial == 0:
            n0, extra = n0_max - 1, extra_max - 1          # the largest tables of the case
        base = rng.uniform(0, 5000, (n0, 2))
        ct = CentroidTracker(max_disappeared=30, fps=30.0, use_gsff=False, capacity=cap, max_det=cap)
        ot = oracle.OracleTracker(max_disappeared=30, fps=30.0, use_gsff=False)
        info = np.zeros((n0, 3))
        ct.update(rects_of(base, info)); ot.update(rects_of(base, info))
        pts = np.vstack([base + rng.normal(0, 0.01, base.shape), rng.uniform(6000, 9000, (extra, 2))])
        pts = pts[rng.permutation(len(pts))]
        info = np.zeros((len(pts), 3))
        objs, _ = ct.update(rects_of(pts, info))
        ids, xy, _, _ = ot.update(rects_of(pts, info))
        assert list(objs.keys()) == ids
        np.testing.assert_array_equal(np.array(list(objs.values())), xy, err_msg=f"trial {trial}")


@pytest.mark.parametrize("cap", [1024, 4096, 16384])
def test_set_order_when_the_unclaimed_columns_collide(torch_cuda, oracle, cap):
    """The set model inserts 64 keys at a time (set_insert_batch): the unclaimed columns are chosen so that their hash
    slots collide as badly as they can -- one residue class of the small tables, a dense run that overflows the linear
    probe window, both mixed -- and so that displacement chains cross the 64-key batches."""
    from ysmr_amd.tracker import CentroidTracker
    rng = np.random.default_rng(5)
    total = 900
    patterns = [np.arange(0, total, 128), np.arange(0, total, 32), np.arange(7, total, 8)[:90], np.arange(100, 190),
                np.r_[np.arange(0, total, 64), np.arange(300, 345), np.arange(513, 530)],
                np.r_[np.arange(3, total, 16), np.arange(640, 700)], np.arange(total - 70, total)]
    for cols in patterns:
        cols = np.unique(cols)
        n0 = total - len(cols)
        assert n0 >= total // 4          # (otherwise the reference iterates a plain copy: ascending)
        base = rng.uniform(0, 5000, (n0, 2))
        ct = CentroidTracker(max_disappeared=30, fps=30.0, use_gsff=False, capacity=cap, max_det=cap)
        ot = oracle.OracleTracker(max_disappeared=30, fps=30.0, use_gsff=False)
        info = np.zeros((n0, 3))
        ct.update(rects_of(base, info)); ot.update(rects_of(base, info))
        pts = np.empty((total, 2))
        old = np.setdiff1d(np.arange(total), cols)
        pts[old] = base + rng.normal(0, 0.01, base.shape)
        pts[cols] = rng.uniform(6000, 9000, (len(cols), 2))
        info = np.zeros((total, 3))
        objs, _ = ct.update(rects_of(pts, info))
        ids, xy, _, _ = ot.update(rects_of(pts, info))
        assert list(objs.keys()) == ids
        np.testing.assert_array_equal(np.array(list(objs.values())), xy)


def test_distance_ties_take_lowest_column_and_row(torch_cuda, oracle):
    from ysmr_amd.tracker import CentroidTracker
    ct = CentroidTracker(max_disappeared=30, fps=30.0, use_gsff=False, capacity=64, max_det=64)
    ot = oracle.OracleTracker(max_disappeared=30, fps=30.0, use_gsff=False)
    start = [((0.0, 0.0), (1, 1, 0)), ((10.0, 0.0), (1, 1, 0)), ((20.0, 5.0), (1, 1, 0))]
    # detections equidistant from track 0 / two tracks equidistant from one detection
    nxt = [((3.0, 4.0), (2, 2, 0)), ((-3.0, 4.0), (3, 3, 0)), ((5.0, 0.0), (4, 4, 0)), ((20.0, 5.0), (5, 5, 0))]
    for rects in (start, nxt, nxt):
        objs, infos = ct.update(rects)
        ids, xy, info, claims = ot.update(rects)
        assert list(objs.keys()) == ids
        np.testing.assert_array_equal(np.array(list(objs.values())), xy)
        assert sorted(ct.last_claims) == sorted(claims)
        assert [list(infos[i]) for i in ids] == [list(i) for i in info]


def test_pipeline_rows_match_oracle(torch_cuda, oracle):
    """frames -> HIP detect -> HIP link (device resident, batched) vs the CPU oracle end to end."""
    torch = torch_cuda
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    from ysmr_amd import _lib
    n_frames, h, w = 48, 240, 320
    frames = SyntheticVideo(h, w, 40, seed=7, dropout=0.05, speckle=0.05).frames(n_frames)
    ref_rows, _ = oracle.track_frames(frames, fps=30.0, shadows=2)
    det = Detector(16, h, w, max_det=256, params=threshold_params(True, 5, 2.0))
    trk = DeviceTracker(max_disappeared=30.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=256, max_det=256)
    rows = torch.empty(n_frames * 256 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    dev = torch.from_numpy(frames).cuda()
    for f0 in range(0, n_frames, 16):
        res = det.detect(dev[f0:f0 + 16])
        trk.run(res.det, res.det_count, f0, rows, count)
    torch.cuda.synchronize()
    assert trk.info()[2] == 0
    got = rows_to_numpy(rows, int(count.item()))
    assert (got["disappeared"] > 0).sum() > 100   # the clip does contain lost-track episodes
    n_loose, worst = compare_rows(got, ref_rows)
    assert n_loose < 0.05 * len(got)              # ... of which only a few are ill-conditioned in the reference itself


def test_table_grows_and_shrinks_abruptly(torch_cuda, oracle):
    """5 tracks, then 700 detections appear at once (registration of hundreds of tracks in one frame,
    CPython set order over a large table), most vanish (mass ageing and deregistration, compaction
    over several 256-row chunks), then 600 come back."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    rng = np.random.default_rng(12)
    cap = md = 1024
    def dets(n, jitter, base):
        xy = base[:n] + rng.normal(0, jitter, (n, 2))
        whd = np.column_stack([rng.uniform(1, 9, n), rng.uniform(1, 9, n), rng.uniform(-90, 0, n)])
        return np.column_stack([xy, whd]).astype(np.float32)
    base = rng.uniform(0, 4000, (700, 2))
    per_frame = [dets(5, 0.3, base) for _ in range(8)] + [dets(700, 0.3, base) for _ in range(8)] + \
                [dets(40, 0.3, base) for _ in range(8)] + [dets(600, 0.3, base) for _ in range(8)]
    ot = oracle.OracleTracker(max_disappeared=3.0, fps=30.0, n_min=0, n_max=30, n_f=3, use_gsff=True, shadows=2)
    ref_rows = []
    for f, d in enumerate(per_frame):
        ids, xy, info, _ = ot.update(oracle.det_to_rects(d))
        ref_rows += [(f, tid, float(xy[i][0]), float(xy[i][1]), *map(float, info[i]), float(ot.last_sens[i]))
                     for i, tid in enumerate(ids)]
    trk = DeviceTracker(max_disappeared=3.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=cap, max_det=md)
    rows = torch.empty(len(per_frame) * cap * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    for b0 in range(0, len(per_frame), 8):
        det = torch.zeros(8, md, 5, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
        for k, d in enumerate(per_frame[b0:b0 + 8]):
            det[k, :len(d)] = torch.from_numpy(d).cuda()
            cnt[k] = len(d)
        trk.run(det, cnt, b0, rows, count)
    assert trk.info()[2] == 0
    compare_rows(rows_to_numpy(rows, int(count.item())), ref_rows)


def test_centroid_tracker_public_attributes_and_payloads():
    """The reference's public attributes (objects, disappeared, nextObjectID, additional_info) and its
    tolerance for arbitrary additional_info payloads (tracker.py:73-82, 111-131)."""
    from ysmr_amd.tracker import CentroidTracker
    ct = CentroidTracker(max_disappeared=2, fps=30.0, use_gsff=False, capacity=16, max_det=4)
    objs, info = ct.update([((10.0, 10.0), (3.0, 1.0, -45.0)), ((50.0, 20.0), "tag")])
    assert list(objs) == [0, 1] and ct.nextObjectID == 2
    assert info[0] == (3.0, 1.0, -45.0) and info[1] == "tag"            # payloads come back as given
    assert list(ct.objects) == [0, 1] and np.allclose(ct.objects[1], [50.0, 20.0])
    assert dict(ct.disappeared) == {0: 0, 1: 0}
    objs, info = ct.update([((11.0, 10.0), (3.0, 1.0, -40.0))])
    assert dict(ct.disappeared) == {0: 0, 1: 1} and info[1] == [0, 0, 0] and ct.last_claims == [(0, 0)]
    with pytest.raises(ValueError):
        ct.update([((float(i), 0.0), (1, 1, 0)) for i in range(5)])       # more than max_det
    with pytest.raises(NotImplementedError):
        ct.update([((1.0, 2.0, 3.0), (1, 1, 0))])                         # luminosity dimension
    ct.update([]); ct.update([]); ct.update([])
    assert list(ct.objects) == [] and ct.nextObjectID == 2


def test_capacity_beyond_the_lds_tables(torch_cuda, oracle):
    """capacity / max_det = 16384: k_link's per-column and per-row tables no longer fit in LDS and live in HBM;
    the row minima of a batch come from the detection grid (ysmr_tracker_run) or from all pairs
    (ysmr_tracker_update).  Same rows as the oracle either way, including tracks far from every detection
    (the grid search gives up after four rings and falls back to all pairs)."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    rng = np.random.default_rng(3)
    cap = 16384
    base = rng.uniform(0, 3000, (900, 2))
    base[:5] += 40000.0                                   # a far-away cluster: most of the grid is empty
    frames = []
    for f in range(10):
        keep = rng.random(len(base)) > (0.15 if f % 3 else 0.5)      # heavy dropout every third frame: lost tracks
        xy = base[keep] + rng.normal(0, 0.4, (int(keep.sum()), 2))
        whd = np.column_stack([rng.uniform(1, 9, len(xy)), rng.uniform(1, 9, len(xy)), rng.uniform(0, 90, len(xy))])
        frames.append(np.column_stack([xy, whd]).astype(np.float32))
        base += rng.normal(0, 0.3, base.shape)
    ot = oracle.OracleTracker(max_disappeared=3.0, fps=30.0, n_min=0, n_max=30, n_f=3, shadows=2)
    ref_rows = []
    for f, d in enumerate(frames):
        ids, xy, info, _ = ot.update(oracle.det_to_rects(d))
        ref_rows += [(f, tid, float(xy[i][0]), float(xy[i][1]), *map(float, info[i]), float(ot.last_sens[i]))
                     for i, tid in enumerate(ids)]
    for mode in ("run", "update"):
        trk = DeviceTracker(max_disappeared=3.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=cap, max_det=cap)
        rows = torch.empty(len(frames) * 1024 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        count = torch.zeros(1, dtype=torch.int64, device="cuda")
        if mode == "run":
            for b0 in range(0, len(frames), 5):
                det = torch.zeros(5, cap, 5, dtype=torch.float32, device="cuda")
                cnt = torch.zeros(5, dtype=torch.int32, device="cuda")
                for k, d in enumerate(frames[b0:b0 + 5]):
                    det[k, :len(d)] = torch.from_numpy(d).cuda()
                    cnt[k] = len(d)
                trk.run(det, cnt, b0, rows, count)
            got = rows_to_numpy(rows, int(count.item()))
        else:
            parts = []
            one = torch.empty(cap * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
            n1 = torch.zeros(1, dtype=torch.int32, device="cuda")
            for f, d in enumerate(frames):
                trk.update(torch.from_numpy(d).cuda(), m=len(d), frame=f, rows=one, n_rows=n1)
                parts.append(rows_to_numpy(one, int(n1.item())).copy())
            got = np.concatenate(parts)
        assert trk.info()[2] == 0
        compare_rows(got, ref_rows)


def test_split_link_lane_per_track_over_ring_wraps_and_refreshes(torch_cuda, oracle):
    """Tables of more than 768 tracks link with two launches per frame, the second with a track per LANE (batch_link.h:
    k_track_lanes): filter state seat-major by slot, a 32-entry ring, window sums rebuilt every 64th frame.  140 frames of
    ~900 moving tracks with dropout (tracks lost, dropped and their slots taken by new ids) cross the ring's wrap four
    times and the refresh twice; batches of 16 and of 1 against the oracle (tracker.py:113-240)."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    rng = np.random.default_rng(11)
    n_frames, cap = 140, 2048
    base = rng.uniform(0, 3800, (900, 2))
    vel = rng.normal(0, 0.6, base.shape)
    frames = []
    for f in range(n_frames):
        keep = rng.random(len(base)) > 0.04
        if f % 37 == 36:
            keep[rng.integers(0, len(base), 60)] = False          # a burst of losses
        xy = base[keep] + rng.normal(0, 0.3, (int(keep.sum()), 2))
        whd = np.column_stack([rng.uniform(1, 9, len(xy)), rng.uniform(1, 9, len(xy)), rng.uniform(0, 90, len(xy))])
        frames.append(np.column_stack([xy, whd]).astype(np.float32))
        base += vel
        if f % 20 == 19:                                          # some leave for good, others appear
            gone = rng.integers(0, len(base), 25)
            base[gone] = rng.uniform(0, 3800, (25, 2))
    ot = oracle.OracleTracker(max_disappeared=4.0, fps=30.0, n_min=0, n_max=30, n_f=3, shadows=2)
    ref_rows = []
    for f, d in enumerate(frames):
        ids, xy, info, _ = ot.update(oracle.det_to_rects(d))
        ref_rows += [(f, tid, float(xy[i][0]), float(xy[i][1]), *map(float, info[i]), float(ot.last_sens[i]))
                     for i, tid in enumerate(ids)]
    for batch in (16, 1):
        trk = DeviceTracker(max_disappeared=4.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=cap, max_det=cap)
        assert not trk.batched
        rows = torch.empty(n_frames * 1100 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        count = torch.zeros(1, dtype=torch.int64, device="cuda")
        for b0 in range(0, n_frames, batch):
            part = frames[b0:b0 + batch]
            det = torch.zeros(len(part), cap, 5, dtype=torch.float32, device="cuda")
            cnt = torch.zeros(len(part), dtype=torch.int32, device="cuda")
            for k, d in enumerate(part):
                det[k, :len(d)] = torch.from_numpy(d).cuda()
                cnt[k] = len(d)
            trk.run(det, cnt, b0, rows, count)
        got = rows_to_numpy(rows, int(count.item()))
        assert trk.info()[2] == 0
        compare_rows(got, ref_rows)

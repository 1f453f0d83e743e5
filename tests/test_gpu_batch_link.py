"""GPU parity of the batch link (csrc/batch_link.h: k_bgrid + k_batch, one launch per batch of frames, a track per lane)
against (a) the fixtures captured from the reference's own tracker.py / gsff.py, (b) the CPU oracle and (c) the
per-frame kernels (k_frame), through ``ysmr_tracker_run``.

Integer quantities (ids, disappeared counters, row order) are exact; GSFF-smoothed positions are held to 1e-9 like
every other link test (north_star: 1e-5 relative).
"""
import numpy as np
import pytest

from conftest import compare_rows, golden, tracker_frames

pytestmark = pytest.mark.gpu
RTOL = ATOL = 1e-9


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def _run_frames(torch, trk, per_frame, batch, max_det, rows_cap, mode_switch=None):
    """Feed per_frame = [(det (m,2), info (m,3)), ...] through DeviceTracker.run in batches; returns the rows."""
    from ysmr_amd import _lib
    from ysmr_amd.tracker import rows_to_numpy
    rows = torch.empty(rows_cap * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    for k, b0 in enumerate(range(0, len(per_frame), batch)):
        chunk = per_frame[b0:b0 + batch]
        det = torch.zeros(len(chunk), max_det, 5, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(len(chunk), dtype=torch.int32, device="cuda")
        for i, (d, info) in enumerate(chunk):
            if len(d):
                det[i, :len(d)] = torch.from_numpy(np.column_stack([d, info]).astype(np.float32)).cuda()
            cnt[i] = len(d)
        if mode_switch is not None:
            mode_switch(k)
        trk.run(det, cnt, b0, rows, count)
    torch.cuda.synchronize()
    assert trk.info()[2] == 0
    return rows_to_numpy(rows, int(count.item()))


FIXTURES = [("tracker_small_gsff.npz", None), ("tracker_mid_gsff.npz", None), ("tracker_gap.npz", 5), ("tracker_2997.npz", 6)]


@pytest.mark.parametrize("batch", [256, 64, 7, 1])
@pytest.mark.parametrize("name,max_gone", FIXTURES)
def test_batch_link_matches_reference_fixture(torch_cuda, name, max_gone, batch):
    """The reference's own CentroidTracker outputs (tests/golden/gen_golden.py), frame by frame, from one launch per
    batch: ids in table order, filtered positions, boxes, disappeared counters, ids issued."""
    from ysmr_amd.tracker import DeviceTracker
    g = golden(name)
    fps = float(g["fps"])
    n_max = None if int(g["n_max"]) < 0 else int(g["n_max"])
    trk = DeviceTracker(max_disappeared=fps if max_gone is None else max_gone, fps=fps, n_min=int(g["n_min"]), n_max=n_max,
                        n_f=int(g["n_f"]), use_gsff=bool(g["use_gsff"]), capacity=512, max_det=512)
    assert trk.batched and not trk.fused
    per_frame = list(tracker_frames(g))
    got = _run_frames(torch_cuda, trk, per_frame, batch, 512, len(g["ids"]) + 8)
    off = g["off"]
    assert len(got) == len(g["ids"])
    frames = np.repeat(np.arange(len(off) - 1), np.diff(off))
    np.testing.assert_array_equal(got["frame"], frames)
    np.testing.assert_array_equal(got["track_id"], g["ids"])
    np.testing.assert_array_equal(got["disappeared"], g["disappeared"])
    np.testing.assert_allclose(got["x"], g["xy"][:, 0], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(got["y"], g["xy"][:, 1], rtol=RTOL, atol=ATOL)
    for k, key in enumerate(("w", "h", "angle")):
        np.testing.assert_array_equal(got[key], g["info"][:, k].astype(np.float32))
    assert trk.info()[1] == g["next_id"][-1]
    ids, xy, gone = trk.peek()
    sl = slice(off[-2], off[-1])
    np.testing.assert_array_equal(ids, g["ids"][sl])
    np.testing.assert_array_equal(gone, g["disappeared"][sl])


def test_batch_link_without_gsff(torch_cuda, oracle):
    """'disable gsff': rows carry the raw centroids, lost tracks stay where they were (tracker.py:228-230)."""
    from ysmr_amd.tracker import DeviceTracker
    rng = np.random.default_rng(3)
    base = rng.uniform(0, 900, (40, 2))
    per_frame = []
    for f in range(30):
        keep = rng.random(40) > 0.1
        xy = (base + rng.normal(0, 0.4, base.shape))[keep].astype(np.float32).astype(np.float64)
        per_frame.append((xy, np.column_stack([rng.uniform(1, 9, len(xy)), rng.uniform(1, 9, len(xy)), rng.uniform(0, 90, len(xy))]).astype(np.float32).astype(np.float64)))
    ot = oracle.OracleTracker(max_disappeared=4.0, fps=30.0, use_gsff=False)
    ref = []
    for f, (d, info) in enumerate(per_frame):
        ids, xy, inf, _ = ot.update(oracle.det_to_rects(np.column_stack([d, info]).astype(np.float32)))
        ref += [(f, tid, float(xy[i][0]), float(xy[i][1]), *map(float, inf[i])) for i, tid in enumerate(ids)]
    trk = DeviceTracker(max_disappeared=4.0, fps=30.0, use_gsff=False, capacity=128, max_det=64)
    assert trk.batched
    got = _run_frames(torch_cuda, trk, per_frame, 8, 64, len(ref) + 8)
    compare_rows(got, ref)


def test_batch_link_equals_per_frame_link_and_survives_switching(torch_cuda, oracle):
    """The same clip through (a) one launch per batch, (b) one launch per frame (``link_mode(1)``: k_frame), (c) both
    in alternation with single-frame ``update`` calls in between -- the track table changes its layout in HBM at every
    switch (k_to_std / k_to_batch) -- and the oracle.  Lost tracks, deregistrations, births throughout."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker
    rng = np.random.default_rng(8)
    n_blobs, n_frames = 300, 96
    pos = rng.uniform(20, 1200, (n_blobs, 2))
    vel = rng.normal(0, 1.5, (n_blobs, 2))
    per_frame = []
    for f in range(n_frames):
        pos = pos + vel + rng.normal(0, 0.2, pos.shape)
        keep = rng.random(n_blobs) > 0.05
        if 40 <= f < 52:
            keep[:60] = False                                   # a block of blobs vanishes for 12 frames: deregistrations
        xy = pos[keep]
        if f % 9 == 4:
            xy = np.vstack([xy, rng.uniform(1300, 1600, (5, 2))])  # spurious detections: births
        xy = xy.astype(np.float32).astype(np.float64)
        info = np.column_stack([rng.uniform(1, 9, len(xy)), rng.uniform(1, 9, len(xy)), rng.uniform(0, 90, len(xy))])
        per_frame.append((xy, info.astype(np.float32).astype(np.float64)))
    kw = dict(max_disappeared=8.0, fps=30.0, n_min=0, n_max=30, n_f=3)
    ot = oracle.OracleTracker(use_gsff=True, shadows=2, **kw)
    ref = []
    for f, (d, info) in enumerate(per_frame):
        ids, xy, inf, _ = ot.update(oracle.det_to_rects(np.column_stack([d, info]).astype(np.float32)))
        ref += [(f, tid, float(xy[i][0]), float(xy[i][1]), *map(float, inf[i]), float(ot.last_sens[i])) for i, tid in enumerate(ids)]
    assert ot.next_id > n_blobs + 30 and len(ot.tracks) < ot.next_id - 20      # births and deaths happened
    cap = 512
    a = DeviceTracker(capacity=cap, max_det=512, **kw)
    rows_a = _run_frames(torch, a, per_frame, 16, 512, len(ref) + 8)
    compare_rows(rows_a, ref)
    b = DeviceTracker(capacity=cap, max_det=512, **kw)
    b.link_mode(1)
    assert not b.batched and b.fused
    rows_b = _run_frames(torch, b, per_frame, 16, 512, len(ref) + 8)
    compare_rows(rows_b, ref)
    for key in ("frame", "track_id", "disappeared", "w", "h", "angle"):
        np.testing.assert_array_equal(rows_a[key], rows_b[key])
    np.testing.assert_allclose(rows_a["x"], rows_b["x"], rtol=1e-9, atol=1e-9)
    # (c) alternate the link modes batch by batch; the table is carried over
    c = DeviceTracker(capacity=cap, max_det=512, **kw)
    rows_c = _run_frames(torch, c, per_frame, 16, 512, len(ref) + 8, mode_switch=lambda k: c.link_mode(k & 1))
    compare_rows(rows_c, ref)
    # (d) single-frame updates (the per-frame kernels, rows and counters as device outputs) between batch launches
    d = DeviceTracker(capacity=cap, max_det=512, **kw)
    rows = torch.empty((len(ref) + 8) * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    one = torch.empty(cap * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    n_one = torch.zeros(1, dtype=torch.int32, device="cuda")
    from ysmr_amd.tracker import rows_to_numpy
    pieces = []
    f = 0
    while f < n_frames:
        nb = min(11, n_frames - f)
        chunk = per_frame[f:f + nb]
        det = torch.zeros(nb, 512, 5, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(nb, dtype=torch.int32, device="cuda")
        for i, (xy, info) in enumerate(chunk):
            det[i, :len(xy)] = torch.from_numpy(np.column_stack([xy, info]).astype(np.float32)).cuda()
            cnt[i] = len(xy)
        count.zero_()
        d.run(det, cnt, f, rows, count)
        pieces.append(rows_to_numpy(rows, int(count.item())).copy())
        f += nb
        if f < n_frames:
            xy, info = per_frame[f]
            dd = torch.from_numpy(np.column_stack([xy, info]).astype(np.float32)).cuda()
            d.update(dd, m=len(xy), frame=f, rows=one, n_rows=n_one)
            pieces.append(rows_to_numpy(one, int(n_one.item())).copy())
            f += 1
    assert d.info()[2] == 0
    compare_rows(np.concatenate(pieces), ref)


def test_batch_link_far_tracks_ties_and_empty_frames(torch_cuda, oracle):
    """The ring search's corner cases: tracks far from every detection (wider rings, then the scan of every detection),
    detections exactly equidistant from a track and tracks exactly equidistant from a detection (lowest column, lowest
    id), frames without detections (everything ages), a frame with one detection, a table emptied by ageing that fills
    again."""
    from ysmr_amd.tracker import DeviceTracker
    z3 = np.zeros((0, 3))

    def fr(points):
        p = np.array(points, dtype=np.float64).reshape(-1, 2)
        return p, np.tile([2.0, 3.0, 45.0], (len(p), 1))

    per_frame = [
        fr([(0, 0), (10, 0), (20, 5), (1000, 1000), (500, 3)]),
        fr([(3, 4), (-3, 4), (5, 0), (20, 5)]),                      # two detections equidistant from track 0; n >= m
        fr([(3, 4), (-3, 4), (5, 0), (20, 5)]),
        fr([]),
        fr([(2000, 2000)]),                                          # one detection, far from everyone: all propose it
        fr([(0, 0), (10, 0), (20, 5), (1000, 1000), (500, 3), (700, 700), (701, 700)]),
        fr([]), fr([]), fr([]), fr([]),                              # max_disappeared = 3: the table empties
        fr([(5, 5), (6, 6)]),
        fr([(5.5, 5.5)]),                                            # exactly between the two tracks
        fr([(5, 5), (6, 6), (5.5, 5.5)]),
    ]
    for use_gsff in (False, True):
        ot = oracle.OracleTracker(max_disappeared=3.0, fps=30.0, n_min=0, n_max=30, n_f=3, use_gsff=use_gsff)
        ref = []
        for f, (d, info) in enumerate(per_frame):
            ids, xy, inf, _ = ot.update(oracle.det_to_rects(np.column_stack([d, info]).astype(np.float32)) if len(d) else [])
            ref += [(f, tid, float(xy[i][0]), float(xy[i][1]), *map(float, inf[i])) for i, tid in enumerate(ids)]
        for batch in (13, 4, 1):
            trk = DeviceTracker(max_disappeared=3.0, fps=30.0, n_min=0, n_max=30, n_f=3, use_gsff=use_gsff, capacity=16, max_det=8)
            assert trk.batched
            got = _run_frames(torch_cuda, trk, per_frame, batch, 8, len(ref) + 4)
            compare_rows(got, ref)


def test_batch_link_capacity_overflow_is_reported(torch_cuda):
    """More tracks than the handle's capacity: the error bit, as with the per-frame kernels."""
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker
    trk = DeviceTracker(max_disappeared=3.0, fps=30.0, capacity=8, max_det=16)
    assert trk.batched
    det = torch_cuda.zeros(1, 16, 5, dtype=torch_cuda.float32, device="cuda")
    det[0, :12, 0] = torch_cuda.arange(12, dtype=torch_cuda.float32, device="cuda") * 10
    cnt = torch_cuda.tensor([12], dtype=torch_cuda.int32, device="cuda")
    rows = torch_cuda.empty(64 * _lib.ROW_DTYPE.itemsize, dtype=torch_cuda.uint8, device="cuda")
    count = torch_cuda.zeros(1, dtype=torch_cuda.int64, device="cuda")
    trk.run(det, cnt, 0, rows, count)
    n, next_id, err = trk.info()
    assert n == 8 and err & 1 and next_id == 12 and int(count.item()) == 8


def test_prepared_batches_give_the_same_rows(torch_cuda):
    """``ysmr_tracker_prepare`` on another stream (the detection stream of a pipeline) takes the binning launch off the
    link stream; the rows are those of an unprepared run, also when a prepared block is left unused and when the two
    blocks are used out of turn."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    rng = np.random.default_rng(2)
    base = rng.uniform(0, 1000, (200, 2))
    dets, cnts = [], []
    for b in range(5):
        det = torch.zeros(16, 256, 5, dtype=torch.float32, device="cuda")
        cnt = torch.zeros(16, dtype=torch.int32, device="cuda")
        for i in range(16):
            keep = rng.random(200) > 0.06
            xy = (base + rng.normal(0, 0.5, base.shape))[keep]
            det[i, :len(xy), :2] = torch.from_numpy(xy.astype(np.float32)).cuda()
            det[i, :len(xy), 2:] = 2.0
            cnt[i] = len(xy)
        dets.append(det); cnts.append(cnt)
    kw = dict(max_disappeared=4.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=256, max_det=256)
    out = []
    side = torch.cuda.Stream()
    for prepared in (False, True):
        trk = DeviceTracker(**kw)
        rows = torch.empty(5 * 16 * 256 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        count = torch.zeros(1, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for b in range(5):
            if prepared and b != 3:                      # (batch 3: prepared for nobody -- run bins it itself)
                side.wait_stream(torch.cuda.current_stream())    # (a block is free once the run that used it has executed)
                with torch.cuda.stream(side):
                    trk.prepare(dets[b], cnts[b], slot=b & 1)
                    if b == 1:
                        trk.prepare(dets[4], cnts[4], slot=0)   # a block prepared early, for a later batch; used in its turn
                torch.cuda.current_stream().wait_stream(side)
            trk.run(dets[b], cnts[b], 16 * b, rows, count)
        torch.cuda.synchronize()
        assert trk.info()[2] == 0
        out.append(rows_to_numpy(rows, int(count.item())).copy())
    assert len(out[0]) > 10000 and out[0].tobytes() == out[1].tobytes()


def test_batch_link_with_all_twelve_waves_seated(torch_cuda, oracle):
    """More than 704 live tracks for the whole clip (tests/link_clips.py: 720-754 of the 768 seats, deaths and births in
    that state): every wave of k_batch holds tracks, so there is no helper wave and the frame's chores -- the next frame's
    LDS-DMA, clearing its tables -- take their other branch (batch_link.h: helpers_from == BL_WAVES).  Frame 64, where
    the window sums are re-summed from the ring, lies inside the second launch (batches of 48) and inside the only one
    (one launch of 112 frames).  Against the oracle: ids, order, counters exact, positions to 1e-9."""
    from link_clips import crowded_clip, oracle_rows
    from ysmr_amd.tracker import DeviceTracker
    kw = dict(max_disappeared=5.0, fps=30.0, n_min=0, n_max=30, n_f=3)
    per_frame = crowded_clip()
    ref, live, ot = oracle_rows(oracle, per_frame, use_gsff=True, shadows=2, **kw)
    assert live.min() > 704 and live.max() <= 768
    for batch in (48, 112):
        trk = DeviceTracker(capacity=768, max_det=1024, **kw)
        assert trk.batched
        got = _run_frames(torch_cuda, trk, per_frame, batch, 1024, len(ref) + 8)
        compare_rows(got, ref)
        assert trk.info()[:2] == (int(live[-1]), ot.next_id)


@pytest.mark.parametrize("stationary", [False, True])
def test_batch_link_more_than_600_detections_per_frame(torch_cuda, oracle, stationary):
    """~760 tracks and 620-760 detections per frame: k_bgrid bins them into 48 x 48 cells (other LDS offsets than every
    other link test, whose frames hold at most 512), and bl_search_wave -- asked in every frame by the lost track far from
    all detections -- scans them exactly, without its float pre-pass (m > 512).  ``stationary`` (GSFF off, blobs that do not
    move): the planted EXACT ties -- two detections equidistant from a track (lowest column), two tracks equidistant from a
    detection (lowest id) -- in frames 5, 6, 20."""
    from link_clips import dense_detection_clip, oracle_rows
    from ysmr_amd.tracker import DeviceTracker
    kw = dict(max_disappeared=5.0, fps=30.0, n_min=0, n_max=30, n_f=3, use_gsff=not stationary)
    per_frame = dense_detection_clip(stationary=stationary)
    ref, live, ot = oracle_rows(oracle, per_frame, shadows=0 if stationary else 2, **kw)
    assert min(len(d) for d, _ in per_frame) > 600 and live.max() <= 768
    for batch in (40, 7):
        trk = DeviceTracker(capacity=768, max_det=1024, **kw)
        assert trk.batched
        got = _run_frames(torch_cuda, trk, per_frame, batch, 1024, len(ref) + 8)
        compare_rows(got, ref)
        assert trk.info()[:2] == (int(live[-1]), ot.next_id)


def test_batch_link_a_frame_of_1500_detections_overflows_cleanly(torch_cuda):
    """1500 detections in one frame (64 cells per side: the largest grid block) against 768 seats: the error bit, 768
    rows, nothing written out of bounds -- and after a reset the handle links again."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    rng = np.random.default_rng(4)
    trk = DeviceTracker(max_disappeared=3.0, fps=30.0, capacity=768, max_det=2048)
    assert trk.batched
    det = torch.zeros(2, 2048, 5, dtype=torch.float32, device="cuda")
    xy = np.column_stack([rng.uniform(0, 3000, 1500), rng.uniform(0, 2000, 1500)]).astype(np.float32)
    det[0, :1500, :2] = torch.from_numpy(xy).cuda()
    det[1, :1400, :2] = torch.from_numpy(xy[:1400] + 0.5).cuda()
    cnt = torch.tensor([1500, 1400], dtype=torch.int32, device="cuda")
    guard = 64
    rows = torch.full(((2 * 768 + guard) * _lib.ROW_DTYPE.itemsize,), 0xAB, dtype=torch.uint8, device="cuda")
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    trk.run(det, cnt, 0, rows[:2 * 768 * _lib.ROW_DTYPE.itemsize], count)
    torch.cuda.synchronize()
    n, next_id, err = trk.info()
    assert n == 768 and err & 1 and int(count.item()) <= 2 * 768
    assert bool((rows[2 * 768 * _lib.ROW_DTYPE.itemsize:] == 0xAB).all())
    trk.reset()
    count.zero_()
    det2 = torch.zeros(1, 2048, 5, dtype=torch.float32, device="cuda")
    det2[0, :700, :2] = torch.from_numpy(xy[:700]).cuda()
    trk.run(det2, torch.tensor([700], dtype=torch.int32, device="cuda"), 0, rows, count)
    torch.cuda.synchronize()
    got = rows_to_numpy(rows, int(count.item()))
    assert len(got) == 700 and trk.info()[0] == 700
    np.testing.assert_array_equal(got["track_id"], np.arange(700))
    np.testing.assert_allclose(got["x"], xy[:700, 0].astype(np.float64), rtol=1e-9, atol=1e-9)     # (the filter bank's output for a first measurement)


def test_a_prepared_block_does_not_outlive_a_switch_to_the_per_frame_link(torch_cuda):
    """ysmr_tracker_prepare remembers a binned block by the ADDRESSES of the detections it was made from.  A pipeline
    that prepares a batch, is then switched to the per-frame link for it (where prepare is a no-op), refills the same
    buffers without preparing them and switches back must not have its next batch linked with the old binning
    (ADVICE r04): rows equal those of a tracker that never prepared anything."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.tracker import DeviceTracker, rows_to_numpy
    rng = np.random.default_rng(12)
    base = rng.uniform(0, 1000, (150, 2))

    def batch_of(shift):
        det = torch.zeros(8, 256, 5, dtype=torch.float32)
        cnt = torch.zeros(8, dtype=torch.int32)
        for i in range(8):
            keep = rng.random(150) > 0.1
            xy = (base + shift + rng.normal(0, 0.5, base.shape))[keep]
            det[i, :len(xy), :2] = torch.from_numpy(xy.astype(np.float32))
            det[i, :len(xy), 2:] = 3.0
            cnt[i] = len(xy)
        return det, cnt

    batches = [batch_of(0.0), batch_of(2.0), batch_of(4.0)]
    kw = dict(max_disappeared=4.0, fps=30.0, n_min=0, n_max=30, n_f=3, capacity=256, max_det=256)
    out = []
    for prepared in (False, True):
        trk = DeviceTracker(**kw)
        det = torch.zeros(8, 256, 5, dtype=torch.float32, device="cuda")      # ONE pair of buffers, refilled per batch
        cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
        rows = torch.empty(3 * 8 * 256 * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        count = torch.zeros(1, dtype=torch.int64, device="cuda")
        for b, (d, c) in enumerate(batches):
            det.copy_(d); cnt.copy_(c)
            if prepared and b == 0:
                trk.prepare(det, cnt, slot=0)          # binned for batch 0 ...
            trk.link_mode(1 if b < 2 else 0)           # ... which the per-frame kernels link, as they do batch 1
            trk.run(det, cnt, 8 * b, rows, count)      # batch 2: a batch launch on the same addresses, other contents
        torch.cuda.synchronize()
        assert trk.info()[2] == 0
        out.append(rows_to_numpy(rows, int(count.item())).copy())
    assert len(out[0]) > 3000 and out[0].tobytes() == out[1].tobytes()

"""GPU parity (run with -m gpu on the MI355X box): HIP detection path vs the CPU oracle.

Bit-exact: class maps (thresholded masks), final mask, label map, detection count/order
(anchors), rectangle centre and size.  The rectangle angle goes through a float64 atan2 whose
device implementation is not correctly rounded; it may differ by one float32 ulp.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _detect_gpu(torch, frames, params=None, max_det=4096):
    from ysmr_amd.detect import Detector, threshold_params
    frames = np.ascontiguousarray(frames)
    b, h, w = frames.shape[:3]
    det = Detector(b, h, w, max_det=max_det, params=params or threshold_params(True, 5, 2.0))
    res = det.detect(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    return {k: getattr(res, k).cpu().numpy() for k in ("cls", "mask", "labels", "det_count", "det", "anchors", "status")}


def _assert_angle_close(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    ok = (a == b) | (np.abs(a - b) <= np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)))
    assert ok.all(), (a[~ok], b[~ok])


def _compare(oracle, frames, got, params, max_det=4096):
    for f in range(frames.shape[0]):
        ref = oracle.detect_frame(frames[f], params.inv, params.t_low, params.t_high, params.use_high, max_det)
        assert got["status"][f] == 0
        np.testing.assert_array_equal(got["cls"][f] & 3, ref.cls, err_msg=f"cls frame {f}")
        np.testing.assert_array_equal(got["mask"][f], ref.mask, err_msg=f"mask frame {f}")
        np.testing.assert_array_equal(got["labels"][f], ref.labels, err_msg=f"labels frame {f}")
        assert got["det_count"][f] == ref.count, f"count frame {f}"
        n = ref.count
        np.testing.assert_array_equal(got["anchors"][f][:n], ref.anchors)
        np.testing.assert_array_equal(got["det"][f][:n, :4], ref.det[:, :4], err_msg=f"rect frame {f}")
        _assert_angle_close(got["det"][f][:n, 4], ref.det[:, 4])


def test_threshold_only_matches_oracle(torch_cuda, oracle):
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    rng = np.random.default_rng(0)
    for (h, w) in [(1, 1), (5, 7), (13, 300), (64, 64), (97, 131), (200, 260)]:
        frames = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
        if h >= 64:
            frames[0] = SyntheticVideo(h, w, 12, seed=h).next_frame()
        for args in [(True, 5, 2.0), (False, 5, 2.0), (True, 3, 0.0), (True, 5, 2.5)]:
            p = threshold_params(*args)
            d = Detector(2, h, w, max_det=64, params=p)
            cls = d.threshold(torch.from_numpy(frames).cuda()).cpu().numpy()
            for f in range(2):
                blur = oracle.blur3(frames[f])
                ref = oracle.classify(blur, oracle.adaptive_mean(blur), p.inv, p.t_low, p.t_high, p.use_high)
                np.testing.assert_array_equal(cls[f], ref, err_msg=f"{h}x{w} {args} frame {f}")


def _threshold_reference(oracle, frames, p):
    out = []
    for f in range(frames.shape[0]):
        blur = oracle.blur3(frames[f])
        out.append(oracle.classify(blur, oracle.adaptive_mean(blur), p.inv, p.t_low, p.t_high, p.use_high))
    return np.stack(out)


def _mismatch_report(got, ref, limit=8):
    bad = np.argwhere(got != ref)
    return f"{len(bad)} bytes differ; first (frame, y, x): got/want " + ", ".join(
        f"{tuple(int(v) for v in b)}: {int(got[tuple(b)])}/{int(ref[tuple(b)])}" for b in bad[:limit])


@pytest.mark.parametrize("h,w", [(18, 64), (19, 68), (64, 64), (97, 132), (200, 260), (45, 1228), (40, 1232), (33, 1236),
                                 (35, 2472), (130, 1228)])
def test_threshold_matrix_pipe_kernel(torch_cuda, oracle, h, w):
    """The matrix-pipe threshold kernel (csrc/thr_mfma.hip) on the geometries that exercise its edges: one tile and
    many, widths that are / are not multiples of 16, one and several column panels (1236 splits into 2 x 624, 2472 into
    3), bands that end inside a 16-row step, both polarities, one and two levels.  Variant 0 (shipped), 3 (every pixel
    through its exact path) and 1 (the float32-chain kernels) must all give the oracle's bytes."""
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    rng = np.random.default_rng(h * 10007 + w)
    frames = rng.integers(0, 256, (3, h, w), dtype=np.uint8)
    frames[1] = (rng.normal(40, 2, (h, w))).round().clip(0, 255).astype(np.uint8)
    frames[1, ::9, ::11] = 200
    if h >= 64 and w >= 64:
        frames[2] = SyntheticVideo(h, w, max(4, h * w // 3000), seed=w).next_frame()
    dev = torch.from_numpy(frames).cuda()
    for args in [(True, 5, 2.0), (False, 5, 2.0), (True, 3, 0.0), (True, 5, 2.5)]:
        p = threshold_params(*args)
        ref = _threshold_reference(oracle, frames, p)
        d = Detector(3, h, w, max_det=64, params=p)
        for variant in (0, 3, 1):
            got = d.threshold(dev, variant=variant).cpu().numpy()
            assert np.array_equal(got, ref), f"{h}x{w} {args} variant {variant}: " + _mismatch_report(got, ref)


def test_threshold_matrix_pipe_distance(torch_cuda, oracle):
    """How far is the matrix pipe's mean from cv2's float32 chain?  The shipped kernel re-evaluates every pixel within
    EPS = 1/256 of a level; variant 2 only those within 1/512 (round 3 also ran 1/2048; since round 4 the scale 127.5 / EPS
    is an f16 operand, which ends at 1/512).  On uniform noise s - mean is spread over +-128, about 1/100 of the pixels per
    unit near the levels, so a distance d between the two means beyond 1/512 would show as ~ 4 (d - 1/512) / 100 wrong
    bytes per pixel: none are allowed (an earlier build that decided everything but exact ties differed in 3 of these
    3.9 M bytes, i.e. d ~ 2e-5)."""
    from ysmr_amd.detect import Detector, threshold_params
    torch = torch_cuda
    rng = np.random.default_rng(7)
    h, w = 400, 1228
    frames = rng.integers(0, 256, (8, h, w), dtype=np.uint8)
    dev = torch.from_numpy(frames).cuda()
    for args in [(True, 5, 2.0), (False, 5, 2.0), (True, 3, 0.0)]:
        p = threshold_params(*args)
        ref = _threshold_reference(oracle, frames, p)
        d = Detector(8, h, w, max_det=64, params=p)
        for variant in (2, 0):
            got = d.threshold(dev, variant=variant).cpu().numpy()
            assert np.array_equal(got, ref), f"{args} variant {variant}: " + _mismatch_report(got, ref)


def _detect_all(torch, det, dev):
    r = det.detect(dev)
    torch.cuda.synchronize()
    return [t.clone() for t in (r.cls, r.mask, r.labels, r.det_count, r.det, r.anchors, r.status)]


def test_beside_link_hint_changes_no_byte(torch_cuda):
    """The scheduling hints of ``cv_flavour`` only pick kernels and resident grids: YSMR_BESIDE_LINK (the float32-chain
    threshold kernel, smaller labelling grids: what TrackingPipeline passes next to the one-launch-per-frame link),
    YSMR_BESIDE_BATCH_LINK (the matrix-pipe kernel on 248 workgroups, its rows cut into ranges of equal cost) and
    YSMR_BESIDE_SPLIT_LINK (160 workgroups) leave class map, mask, label map, detections, anchors and counts the same
    bytes as a call without them."""
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    from ysmr_amd import _lib
    torch = torch_cuda
    h, w = 922, 1228
    dev = torch.from_numpy(SyntheticVideo(h, w, 500, seed=2).frames(8)).cuda()
    out = []
    for hint in (None, "link", "batch", "split"):
        d = Detector(8, h, w, max_det=2048, params=threshold_params(True, 5, 2.0), threshold_variant=1 if hint == "link" else 0,
                     beside_batch_link=hint == "batch", beside_split_link=hint == "split")
        assert bool(d.cv_flavour & _lib.BESIDE_LINK) == (hint == "link")
        assert bool(d.cv_flavour & _lib.BESIDE_BATCH_LINK) == (hint == "batch")
        assert bool(d.cv_flavour & _lib.BESIDE_SPLIT_LINK) == (hint == "split")
        out.append(_detect_all(torch, d, dev))
    assert int(out[0][3].min()) > 300 and int(out[0][6].max()) == 0
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("n", [256, 248])
def test_bench_launch_shape_256_frames_beside_the_batch_link(torch_cuda, oracle, n):
    """(n = 248, since round 5 what bench.py and track_bacteria launch: as many frames as workgroups, a whole frame each; n = 256,
    round 4's shape and any caller's who names that batch: ranges that end inside frames.)
    The launch shape bench.py and track_bacteria really use: detection of 256 frames of
    1228 x 922 in ONE call with YSMR_BESIDE_BATCH_LINK -- 248 workgroups of the matrix-pipe kernel, their rows cut into
    unequal ranges by TM_START_ROWS, items that start in the middle of a frame -- gives, byte for byte, the class map, mask,
    label map, detections and anchors of the default grid (256 workgroups, whole columns; its parity with the oracle on
    64-frame calls is test_detect_bench_batch_frame_by_frame); a sample of its frames -- the first, the last, and those in
    which the first workgroups' ranges end -- is also compared with the oracle directly."""
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    h, w = 922, 1228
    p = threshold_params(True, 5, 2.0)
    frames = SyntheticVideo(h, w, 500, seed=6).frames(n)
    dev = torch.from_numpy(frames).cuda()
    plain = _detect_all(torch, Detector(n, h, w, max_det=2048, params=p), dev)
    beside = _detect_all(torch, Detector(n, h, w, max_det=2048, params=p, beside_batch_link=True), dev)
    assert int(plain[3].min()) > 400 and int(plain[6].max()) == 0
    for a, b in zip(plain, beside):
        assert torch.equal(a, b)
    got = dict(zip(("cls", "mask", "labels", "det_count", "det", "anchors", "status"), (t.cpu().numpy() for t in beside)))
    for f in (0, 1, 8, 9, 16, 127, n - 1):     # (with 31 workgroups per XCD on 32 frames, a range ends inside frames 8 k + xcd)
        one = {k: v[f:f + 1] for k, v in got.items()}
        _compare(oracle, frames[f:f + 1], one, p, max_det=2048)


def test_4k_launch_shape_16_frames_beside_the_split_link(torch_cuda, oracle):
    """... and the shape of the 4K configuration: 16 frames of 3840 x 2160 with YSMR_BESIDE_SPLIT_LINK (the matrix-pipe
    kernel on 160 workgroups, four column panels per frame) against the default grid, byte for byte, and two of its frames
    against the oracle."""
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    h, w, n = 2160, 3840, 16
    p = threshold_params(True, 5, 2.0)
    frames = SyntheticVideo(h, w, 5000, seed=7).frames(n)
    dev = torch.from_numpy(frames).cuda()
    plain = _detect_all(torch, Detector(n, h, w, max_det=8192, params=p), dev)
    beside = _detect_all(torch, Detector(n, h, w, max_det=8192, params=p, beside_split_link=True), dev)
    assert int(plain[3].min()) > 4000 and int(plain[6].max()) == 0
    for a, b in zip(plain, beside):
        assert torch.equal(a, b)
    got = dict(zip(("cls", "mask", "labels", "det_count", "det", "anchors", "status"), (t.cpu().numpy() for t in beside)))
    for f in (0, 15):
        one = {k: v[f:f + 1] for k, v in got.items()}
        _compare(oracle, frames[f:f + 1], one, p, max_det=8192)


def test_threshold_dispatch_sets_the_callers_events(torch_cuda, oracle):
    """ysmr_threshold_timing: the kernel's own dispatch sets the two events (what bench.py's roofline divides by); the
    class map is the one an untimed call writes, for every kernel a geometry can take, and the hook is used up by one
    call."""
    from ysmr_amd.detect import Detector, threshold_params
    torch = torch_cuda
    rng = np.random.default_rng(3)
    for (h, w, ch), variants in (((200, 1228, 1), (0, 1)), ((64, 90, 3), (0,)), ((64, 92, 3), (0,))):
        shape = (4, h, w) if ch == 1 else (4, h, w, 3)
        dev = torch.from_numpy(rng.integers(0, 256, shape, dtype=np.uint8)).cuda()
        d = Detector(4, h, w, max_det=64, params=threshold_params(True, 5, 2.0))
        for variant in variants:
            plain = d.threshold(dev, variant=variant).clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            torch.cuda.synchronize()
            timed = d.threshold(dev, variant=variant, timing=(e0, e1)).clone()
            again = d.threshold(dev, variant=variant).clone()       # (no events pending any more)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            assert 0.001 < ms < 50.0, ms
            assert torch.equal(plain, timed) and torch.equal(plain, again)


@pytest.mark.parametrize("h,w", [(70, 90), (70, 92), (130, 1228), (61, 16)])
def test_threshold_bgr(torch_cuda, oracle, h, w):
    """a1: BGR input (what cv2.VideoCapture delivers).  W % 4 == 0 takes the strip kernel (several
    strips at 1228), other widths the tile kernel; colours are random so the fixed-point weights matter."""
    from ysmr_amd.detect import Detector
    torch = torch_cuda
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (2, h, w, 3), dtype=np.uint8)
    frames[1, :, :, :] = (frames[1, :, :, :1] // 3 + 30)       # low-contrast gray-ish frame: thresholds bite
    frames[1, ::7, ::5, 1] = 255
    d = Detector(2, h, w, max_det=64)
    cls = d.threshold(torch.from_numpy(frames).cuda()).cpu().numpy()
    for f in range(2):
        blur = oracle.blur3(oracle.bgr2gray(frames[f]))
        ref = oracle.classify(blur, oracle.adaptive_mean(blur), 0, 5, 7, 1)
        np.testing.assert_array_equal(cls[f], ref)


def test_detect_synthetic_video(torch_cuda, oracle):
    from ysmr_amd.detect import threshold_params
    from ysmr_amd.synth import SyntheticVideo
    p = threshold_params(True, 5, 2.0)
    frames = SyntheticVideo(300, 412, 80, seed=2).frames(5)      # W, H*W not multiples of 16
    _compare(oracle, frames, _detect_gpu(torch_cuda, frames, p), p)
    frames = SyntheticVideo(922, 1228, 500, seed=3).frames(2)     # the benchmark geometry
    _compare(oracle, frames, _detect_gpu(torch_cuda, frames, p), p)


def test_detect_bench_batch_frame_by_frame(torch_cuda, oracle):
    """The detection half of BASELINE configs[1] / configs[2] exactly as bench.py launches it -- 64 frames of
    1228x922 with ~500 blobs in ONE call (frames dealt to the XCDs by the threshold kernel and by k_windows, resident
    grids striding over their work, a second call clearing through the first call's component boxes) -- compared
    with the oracle on every frame: class map, mask, label map, anchors, rectangles."""
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    p = threshold_params(True, 5, 2.0)
    video = SyntheticVideo(922, 1228, 500, seed=0)
    det = Detector(64, 922, 1228, max_det=2048, params=p)
    for call in range(2):
        frames = video.frames(64)
        res = det.detect(torch.from_numpy(frames).cuda())
        torch.cuda.synchronize()
        got = {k: getattr(res, k).cpu().numpy() for k in ("cls", "mask", "labels", "det_count", "det", "anchors", "status")}
        assert got["det_count"].min() > 400
        _compare(oracle, frames, got, p, max_det=2048)


def test_stalled_barrier_is_reported_and_the_next_call_recovers(torch_cuda, oracle):
    """k_residue's software grid barrier gives up when a workgroup never arrives (a word in the test's own workspace
    header makes one stay away, include/ysmr_hip.h: YSMR_WS_FAULT_RESIDUE_STALL):
    every frame of that call reports YSMR_DET_STALLED, the call returns, and the NEXT call on the same detector -- the same
    label map, mask and workspace, which the stalled call left half written -- gives the oracle's result again."""
    from ysmr_amd import _lib
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    p = threshold_params(True, 5, 2.0)
    video = SyntheticVideo(300, 412, 60, seed=11)
    det = Detector(3, 300, 412, max_det=512, params=p)

    def run(frames):
        res = det.detect(torch.from_numpy(frames).cuda())
        torch.cuda.synchronize()
        return {k: getattr(res, k).cpu().numpy() for k in ("cls", "mask", "labels", "det_count", "det", "anchors", "status")}

    def big_blobs(frames):          # islands larger than a 16 x 16 window: residue for the barrier kernel
        frames = frames.copy()
        frames[:, 40:75, 50:95] = 200
        frames[:, 150:170, 200:260] = 180
        return frames

    first = big_blobs(video.frames(3))
    _compare(oracle, first, run(first), p, max_det=512)
    word = torch.tensor([_lib.WS_FAULT_RESIDUE_STALL], dtype=torch.int64).to(torch.int32).view(torch.uint8).cuda()
    det._ws[_lib.WS_FAULT_OFFSET:_lib.WS_FAULT_OFFSET + 4] = word
    stalled = run(big_blobs(video.frames(3)))
    assert (stalled["status"] & _lib.DET_STALLED).all(), stalled["status"]
    after = big_blobs(video.frames(3))
    _compare(oracle, after, run(after), p, max_det=512)
    assert int(det._ws[_lib.WS_FAULT_OFFSET:_lib.WS_FAULT_OFFSET + 4].view(torch.int32).item()) == 0     # (good for one call)


def test_detect_dense_noise_stresses_union_find(torch_cuda, oracle):
    """Uniform noise gives dense, convoluted masks: long union-find chains, holes, nesting."""
    from ysmr_amd.detect import threshold_params
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 256, (3, 150, 201), dtype=np.uint8)
    frames[1] = (rng.random((150, 201)) < 0.5) * np.uint8(200) + 20
    for args in [(True, 5, 2.0), (True, 1, 0.0), (False, 5, 2.0)]:
        p = threshold_params(*args)
        _compare(oracle, frames, _detect_gpu(torch_cuda, frames, p, max_det=16384), p, max_det=16384)


def test_residue_one_workgroup_per_frame(torch_cuda, oracle):
    """Batches of eight frames and more run the union-find passes with one workgroup per frame (k_residue_frames) instead of
    the grid-barrier kernel: the three ways such a workgroup finds its pixels -- its part of the list gathered in LDS (a few
    large islands per frame), the whole list re-read in every pass (one frame with more residue than the LDS list holds),
    every pixel of the frame (the list itself overflowed) -- against the oracle, and the same frames through the
    grid-barrier kernel (a batch of three)."""
    from ysmr_amd.detect import threshold_params
    from ysmr_amd.synth import SyntheticVideo
    p = threshold_params(True, 5, 2.0)
    rng = np.random.default_rng(14)
    h, w = 150, 201
    frames = SyntheticVideo(h, w, 14, seed=3).frames(32)
    for f in range(32):                       # islands beyond a 16 x 16 window, touching pairs, one with a hole
        y, x = 20 + (f * 3) % 60, 30 + (f * 7) % 100
        frames[f, y:y + 24, x:x + 31] = 200
        frames[f, y + 8:y + 12, x + 10:x + 16] = 40
        frames[f, 110:118, 20 + f:60 + f] = 180
    got = _detect_gpu(torch_cuda, frames, p, max_det=4096)
    _compare(oracle, frames, got, p, max_det=4096)
    few = _detect_gpu(torch_cuda, frames[:3], p, max_det=4096)
    for k in ("labels", "mask", "det", "det_count", "anchors"):
        np.testing.assert_array_equal(few[k], got[k][:3], err_msg=k)
    # one frame of coarse blocks: ~10 k residue pixels in that frame (beyond the 8192 of the LDS list), the list still
    # within its capacity of 1/8 of the batch's pixels
    coarse = frames.copy()
    blocks = (rng.random((h // 6 + 1, w // 6 + 1)) < 0.45)
    coarse[5] = np.kron(blocks, np.ones((6, 6), dtype=np.uint8))[:h, :w] * np.uint8(190) + 30
    _compare(oracle, coarse, _detect_gpu(torch_cuda, coarse, p, max_det=4096), p, max_det=4096)
    # noise everywhere: the list overflows, every workgroup walks its frame
    noise = rng.integers(0, 256, (32, 60, 81), dtype=np.uint8)
    _compare(oracle, noise, _detect_gpu(torch_cuda, noise, p, max_det=8192), p, max_det=8192)


def test_detect_nested_component_is_skipped(torch_cuda, oracle):
    from ysmr_amd.detect import threshold_params
    p = threshold_params(True, 5, 2.0)
    img = np.full((120, 160), 40, np.uint8)
    yy, xx = np.mgrid[0:120, 0:160]
    r = np.hypot(yy - 60, xx - 80)
    img[(r > 14) & (r < 18)] = 210          # ring
    img[(np.hypot(yy - 60, xx - 80) < 2.5)] = 210   # dot inside the ring's hole
    img[20:23, 20:28] = 200                 # ordinary blob
    img[(np.hypot(yy - 100, xx - 30) > 5) & (np.hypot(yy - 100, xx - 30) < 8)] = 220  # small ring, empty hole
    frames = np.stack([img, np.full_like(img, 40)])
    got = _detect_gpu(torch_cuda, frames, p)
    _compare(oracle, frames, got, p)
    ref = oracle.detect_frame(img)
    n_comp = len(np.unique(ref.labels)) - 1
    assert ref.count == n_comp - 1          # exactly the nested dot is dropped
    assert got["det_count"][1] == 0


def test_detect_empty_and_full(torch_cuda, oracle):
    from ysmr_amd.detect import threshold_params
    p = threshold_params(True, 5, 2.0)
    frames = np.stack([np.zeros((40, 50), np.uint8), np.full((40, 50), 255, np.uint8)])
    got = _detect_gpu(torch_cuda, frames, p)
    _compare(oracle, frames, got, p)
    assert list(got["det_count"]) == [0, 0]


def test_detect_overflow_is_flagged(torch_cuda):
    from ysmr_amd import _lib
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (1, 100, 100), dtype=np.uint8)
    got = _detect_gpu(torch_cuda, frames, max_det=8)
    assert got["status"][0] & _lib.DET_OVERFLOW
    assert got["det_count"][0] <= 8


def test_large_component_uses_arena(torch_cuda, oracle):
    """A blob wider than the LDS hull fast path (and one with a big hole) goes through the arena."""
    from ysmr_amd.detect import threshold_params
    p = threshold_params(True, 5, 2.0)
    img = np.full((200, 300), 30, np.uint8)
    yy, xx = np.mgrid[0:200, 0:300]
    img[np.abs((yy - 100) * 0.4 + (xx - 150) * 0.2) < 2.0] = 220       # long slanted bar
    r = np.hypot(yy - 100, xx - 60)
    img[(r > 35) & (r < 38)] = 220                                     # big ring (window > LDS? no, but wide hull)
    img[95:99, 55:62] = 220                                            # nested blob
    frames = img[None]
    _compare(oracle, frames, _detect_gpu(torch_cuda, frames, p), p)


def test_components_on_arbitrary_class_maps(torch_cuda, oracle):
    """ysmr_components_batch on caller-supplied class maps with ANY combination of the two bits
    (markers inside, outside and next to the mask): hysteresis = scipy's binary_propagation rule,
    then labels / nesting / rectangles as for thresholded frames."""
    import torch
    from scipy import ndimage
    from ysmr_amd.detect import Detector
    rng = np.random.default_rng(17)
    for (h, w, p_mask, p_mark) in [(61, 83, 0.45, 0.03), (40, 56, 0.6, 0.01), (97, 131, 0.3, 0.1), (33, 47, 0.15, 0.3)]:
        b = 3
        mask = rng.random((b, h, w)) < p_mask
        mark = rng.random((b, h, w)) < p_mark           # independent of the mask: all 4 byte values occur
        cls = (mask.astype(np.uint8) | (mark.astype(np.uint8) << 1))
        det = Detector(b, h, w, max_det=4096)
        res = det.components(cls=torch.from_numpy(cls).cuda())
        torch.cuda.synchronize()
        for f in range(b):
            ref_mask = ndimage.binary_propagation(mark[f], mask=mask[f])
            np.testing.assert_array_equal(res.mask[f].cpu().numpy() > 0, ref_mask, err_msg=f"{h}x{w} frame {f}")
            labels, rects, anchors, n = oracle.components(ref_mask.astype(np.uint8), max_det=4096)
            assert int(res.status[f].item()) == 0
            np.testing.assert_array_equal(res.labels[f].cpu().numpy(), labels)
            assert int(res.det_count[f].item()) == n
            np.testing.assert_array_equal(res.anchors[f, :n].cpu().numpy(), anchors)
            np.testing.assert_array_equal(res.det[f, :n, :4].cpu().numpy(), rects[:, :4])
            _assert_angle_close(res.det[f, :n, 4].cpu().numpy(), rects[:, 4])


def test_islands_at_the_limits_of_a_window(torch_cuda, oracle):
    """k_windows settles islands of up to 16 x 16 pixels in 64 x 64 bit windows around 32 x 32 cores and hands
    larger ones to the union-find passes: shapes whose extent is 14 .. 18 and 31 .. 33 pixels, at offsets that put
    them across core and window boundaries and against the frame's edges, with the marker patterns that matter to
    the hysteresis (marker at one end of a diagonal chain, marker-only pixels next to / diagonal to a thresh
    component, a ring with a dot in its hole, two components that touch only diagonally)."""
    import torch
    from scipy import ndimage
    from ysmr_amd.detect import Detector
    H, W = 330, 470
    shapes = []
    for L in (1, 2, 14, 15, 16, 17, 18, 31, 32, 33):
        bar = np.ones((1, L), np.uint8)
        shapes += [bar * 3, bar.T * 3, np.pad(bar, ((0, 0), (0, 0))) * 1 + np.eye(1, L, dtype=np.uint8) * 2]   # marker at one end
        diag = np.eye(L, dtype=np.uint8)
        shapes += [diag * 3, diag[::-1] * 3, diag + np.eye(L, dtype=np.uint8) * np.r_[2, np.zeros(L - 1, np.uint8)]]
    for L in (15, 16, 17):
        ell = np.zeros((L, L), np.uint8); ell[:, 0] = 3; ell[-1, :] = 3
        box = np.full((L, L), 1, np.uint8); box[L // 2, L // 2] = 3
        ring = np.zeros((L, L), np.uint8); ring[[0, -1], :] = 3; ring[:, [0, -1]] = 3; ring[L // 2, L // 2] = 3   # dot in the hole
        shapes += [ell, ell[::-1, ::-1].copy(), box, ring]
    blob = np.array([[0, 1, 1, 0], [1, 3, 3, 1], [0, 1, 1, 0]], np.uint8)
    shapes += [blob, np.array([[2, 0, 0], [1, 1, 0], [0, 0, 0]], np.uint8),       # marker-only pixel 4-next to thresh
               np.array([[2, 0, 0], [0, 1, 1], [0, 0, 0]], np.uint8),              # ... only diagonal to it: stays alone
               np.array([[3, 3, 0, 0], [0, 0, 1, 1]], np.uint8),                   # R and a marker-less part touching diagonally
               np.array([[3, 1, 0, 0], [0, 0, 3, 1]], np.uint8)]                   # two 4-components, one 8-component of R
    rng = np.random.default_rng(3)
    frames = []
    for shift in range(6):                       # the same catalogue at six alignments to the 32-pixel cores
        cls = np.zeros((H, W), np.uint8)
        y, x, row_h = 1 + 5 * shift, 0 if shift % 2 else 2 + 7 * shift, 0
        for sh in rng.permutation(len(shapes)):
            a = shapes[sh]
            if x + a.shape[1] > W:
                y, x, row_h = y + row_h + 3, (3 * shift) % 11, 0
            if y + a.shape[0] > H:
                break
            cls[y:y + a.shape[0], x:x + a.shape[1]] = a
            x += a.shape[1] + 3
            row_h = max(row_h, a.shape[0])
        frames.append(cls)
    edge = np.zeros((H, W), np.uint8)              # against the four edges and in the corners
    edge[0, :16] = 3; edge[:16, W - 1] = 3; edge[H - 1, W - 17:] = 3; edge[H - 16:, 0] = 3; edge[40:57, 0] = 3; edge[0, 100:117] = 3
    frames.append(edge)
    cls = np.stack(frames)
    b = cls.shape[0]
    det = Detector(b, H, W, max_det=1024)
    for rep in range(2):                           # (second call: cleared through the first call's boxes)
        res = det.components(cls=torch.from_numpy(cls).cuda())
        torch.cuda.synchronize()
        for f in range(b):
            ref_mask = ndimage.binary_propagation((cls[f] & 2) != 0, mask=(cls[f] & 1) != 0)
            np.testing.assert_array_equal(res.mask[f].cpu().numpy() > 0, ref_mask, err_msg=f"frame {f}")
            labels, rects, anchors, n = oracle.components(ref_mask.astype(np.uint8), max_det=1024)
            assert int(res.status[f].item()) == 0 and n > (3 if f == b - 1 else 40)
            np.testing.assert_array_equal(res.labels[f].cpu().numpy(), labels, err_msg=f"frame {f}")
            assert int(res.det_count[f].item()) == n
            np.testing.assert_array_equal(res.anchors[f, :n].cpu().numpy(), anchors)
            np.testing.assert_array_equal(res.det[f, :n, :4].cpu().numpy(), rects[:, :4])
            _assert_angle_close(res.det[f, :n, 4].cpu().numpy(), rects[:, 4])


def test_components_call_after_call_on_random_maps(torch_cuda, oracle):
    """Differential run of the component path: random class maps of mixed character -- sparse small islands (settled by
    k_windows), blobs of 10..40 pixels across (the residue passes), dense noise (the residue outgrows its list: the
    passes walk every pixel and redo the batch) -- in arbitrary order through ONE detector per geometry, so that
    every call also has to clear what the previous one left (component boxes, large boxes, overflowed tables)."""
    import torch
    from scipy import ndimage
    from ysmr_amd.detect import Detector
    rng = np.random.default_rng(99)

    def sparse(b, h, w):
        m = np.zeros((b, h, w), np.uint8)
        for f in range(b):
            for _ in range(max(1, h * w // 900)):
                y, x = rng.integers(0, h), rng.integers(0, w)
                hh, ww = rng.integers(1, 9), rng.integers(1, 9)
                m[f, y:y + hh, x:x + ww] |= rng.choice(np.array([1, 1, 3], np.uint8), size=m[f, y:y + hh, x:x + ww].shape)
        return m

    def blobs(b, h, w):
        m = np.zeros((b, h, w), np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        for f in range(b):
            for _ in range(max(1, h * w // 6000)):
                cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(5, 21)
                ring = rng.random() < 0.3
                d = np.hypot(yy - cy, xx - cx)
                m[f][(d < r) & (~ring | (d > r - 2.5))] |= 1
            m[f][(rng.random((h, w)) < 0.01) & (m[f] > 0)] |= 2
            m[f][rng.random((h, w)) < 0.0005] |= 2                      # markers outside the mask
        return m

    def dense(b, h, w):
        return rng.choice(np.array([0, 1, 2, 3], np.uint8), size=(b, h, w), p=[0.5, 0.3, 0.05, 0.15])

    kinds = [sparse, blobs, dense]
    for (b, h, w, max_det) in [(3, 97, 131, 512), (1, 64, 64, 64), (4, 150, 37, 256), (2, 33, 260, 24), (5, 70, 92, 1024)]:
        det = Detector(b, h, w, max_det=max_det)
        for call in range(6):
            cls = kinds[rng.integers(0, 3)](b, h, w)
            res = det.components(cls=torch.from_numpy(cls).cuda())
            torch.cuda.synchronize()
            for f in range(b):
                ref_mask = ndimage.binary_propagation((cls[f] & 2) != 0, mask=(cls[f] & 1) != 0)
                np.testing.assert_array_equal(res.mask[f].cpu().numpy() > 0, ref_mask, err_msg=f"{b}x{h}x{w} call {call} frame {f}")
                labels, rects, anchors, n = oracle.components(ref_mask.astype(np.uint8), max_det=max_det)
                np.testing.assert_array_equal(res.labels[f].cpu().numpy(), labels, err_msg=f"{b}x{h}x{w} call {call} frame {f}")
                n_all = len(np.unique(labels)) - 1
                assert (int(res.status[f].item()) & 1) == (1 if n_all > max_det else 0)
                if n_all <= max_det:
                    assert int(res.det_count[f].item()) == n
                    np.testing.assert_array_equal(res.anchors[f, :n].cpu().numpy(), anchors)
                    np.testing.assert_array_equal(res.det[f, :n, :4].cpu().numpy(), rects[:, :4])


def test_detector_reuse_clears_what_the_previous_call_wrote(torch_cuda, oracle):
    """The workspace clears labels/mask sparsely from the previous call's pixel list (ysmr_hip.h:
    ysmr_detect_workspace_init).  Reusing one Detector for different clips, a shorter batch, a dense
    batch (list overflow -> dense fallback) and again a sparse one must match the oracle each time,
    and so must a call after someone else has scribbled over the label map (after re-init)."""
    torch = torch_cuda
    from ysmr_amd import _lib
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    p = threshold_params(True, 5, 2.0)
    h, w = 200, 312
    det = Detector(4, h, w, max_det=8192, params=p)
    rng = np.random.default_rng(11)
    clips = [SyntheticVideo(h, w, 60, seed=5).frames(4), SyntheticVideo(h, w, 25, seed=6).frames(4),
             SyntheticVideo(h, w, 40, seed=7).frames(2),                          # shorter batch
             rng.integers(0, 256, (4, h, w), dtype=np.uint8),                     # dense: > 1/8 foreground
             SyntheticVideo(h, w, 30, seed=8).frames(4), SyntheticVideo(h, w, 30, seed=9).frames(4)]
    for k, frames in enumerate(clips):
        res = det.detect(torch.from_numpy(frames).cuda())
        torch.cuda.synchronize()
        got = {n: getattr(res, n).cpu().numpy() for n in ("cls", "mask", "labels", "det_count", "det", "anchors", "status")}
        _compare(oracle, frames, got, p, max_det=8192)
    # foreign writes into the outputs are only legal together with a re-init of the workspace
    det._labels.fill_(7)
    det._mask.fill_(9)
    L = _lib.lib()
    _lib.check(L.ysmr_detect_workspace_init(_lib.stream_ptr(), det._ws.data_ptr(), det._ws.numel()), "ysmr_detect_workspace_init")
    res = det.detect(torch.from_numpy(clips[0]).cuda())
    torch.cuda.synchronize()
    got = {n: getattr(res, n).cpu().numpy() for n in ("cls", "mask", "labels", "det_count", "det", "anchors", "status")}
    _compare(oracle, clips[0], got, p, max_det=8192)


def test_many_large_components_call_after_call(torch_cuda, oracle):
    """More large components (boxes of 64 x 64 pixels and more) per batch than the workspace header once had names for (16): the
    record k_windows leaves covers components of any size, so such batches are cleared like any other -- discs and rings of
    radius 34..60 at other places in every call, label map and mask against scipy / the oracle in full."""
    import torch
    from scipy import ndimage
    from ysmr_amd.detect import Detector
    rng = np.random.default_rng(5)
    b, h, w, max_det = 3, 300, 420, 256
    det = Detector(b, h, w, max_det=max_det)
    yy, xx = np.mgrid[0:h, 0:w]
    for call in range(4):
        cls = np.zeros((b, h, w), np.uint8)
        for f in range(b):
            for _ in range(9):
                cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(34, 61)
                d = np.hypot(yy - cy, xx - cx)
                cls[f][(d < r) & ((rng.random() < 0.5) | (d > r - 3))] |= 1
            cls[f][(rng.random((h, w)) < 0.002) & (cls[f] > 0)] |= 2
        res = det.components(cls=torch.from_numpy(cls).cuda())
        torch.cuda.synchronize()
        assert int(res.status.max().item()) == 0
        for f in range(b):
            ref_mask = ndimage.binary_propagation((cls[f] & 2) != 0, mask=(cls[f] & 1) != 0)
            np.testing.assert_array_equal(res.mask[f].cpu().numpy() > 0, ref_mask, err_msg=f"call {call} frame {f}")
            labels, rects, anchors, n = oracle.components(ref_mask.astype(np.uint8), max_det=max_det)
            np.testing.assert_array_equal(res.labels[f].cpu().numpy(), labels, err_msg=f"call {call} frame {f}")
            assert int(res.det_count[f].item()) == n


def test_detector_without_a_final_mask_call_after_call(torch_cuda):
    """mask_dev = NULL (Detector(want_mask=False)): the label map is cleared and written the same way call after call -- the
    record of where the previous call wrote lives in the workspace, not in the mask -- and everything but the mask equals a
    detector's that keeps one, on clips that differ from call to call (a shorter batch in between: that call clears in full)."""
    torch = torch_cuda
    from ysmr_amd.detect import Detector, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    p = threshold_params(True, 5, 2.0)
    h, w = 200, 312
    with_mask = Detector(4, h, w, max_det=4096, params=p)
    without = Detector(4, h, w, max_det=4096, params=p, want_mask=False)
    rng = np.random.default_rng(3)
    clips = [SyntheticVideo(h, w, 60, seed=15).frames(4), SyntheticVideo(h, w, 25, seed=16).frames(4),
             SyntheticVideo(h, w, 40, seed=17).frames(2), SyntheticVideo(h, w, 30, seed=18).frames(4),
             rng.integers(0, 256, (4, h, w), dtype=np.uint8), SyntheticVideo(h, w, 30, seed=19).frames(4)]
    for k, frames in enumerate(clips):
        dev = torch.from_numpy(frames).cuda()
        a, b = with_mask.detect(dev), without.detect(dev)
        torch.cuda.synchronize()
        assert b.mask is None and a.mask is not None
        for name in ("cls", "labels", "det_count", "anchors", "status"):
            assert torch.equal(getattr(a, name), getattr(b, name)), f"call {k}: {name}"
        n = a.det_count.cpu().numpy()
        for f in range(frames.shape[0]):
            assert torch.equal(a.det[f, :n[f]], b.det[f, :n[f]]), f"call {k} frame {f}: det"
        assert torch.equal((a.labels != 0), (a.mask != 0))


# ---- mean-gray branch ('adaptive double threshold' < 0; track_eval.py:219-253) -------------------
@pytest.mark.parametrize("channels", [1, 3])
@pytest.mark.parametrize("white", [True, False])
def test_mean_gray_branch_matches_oracle(torch_cuda, oracle, channels, white):
    """Levels, statistics, class map and detections of consecutive ragged batches of one video; fps 2
    makes the moving average 11 frames long, so the list is trimmed and carried across calls."""
    from ysmr_amd.detect import Detector, mean_gray_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    rng = np.random.default_rng(7 + channels)
    for (h, w) in [(1, 1), (3, 2), (5, 7), (40, 301), (64, 64), (97, 131), (130, 1228)]:
        n = 30
        if h >= 64:
            gray = SyntheticVideo(h, w, 14, seed=h + w).frames(n)
            gray = (gray.astype(np.int32) + (8 * np.sin(np.arange(n) / 3.0)).astype(np.int32)[:, None, None]).clip(0, 255).astype(np.uint8)
        else:
            gray = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
        if not white:
            gray = 255 - gray
        if channels == 3:
            frames = np.repeat(gray[..., None], 3, axis=3)
            frames[:, ::2, ::3, 1] = rng.integers(0, 256, frames[:, ::2, ::3, 1].shape, dtype=np.uint8)
        else:
            frames = gray
        p = mean_gray_params(white, 5, 2.0)
        assert (p.inv, p.offset, p.window) == (int(not white), 5.0 if white else -5.0, 11)
        det = Detector(16, h, w, max_det=8192, params=p)
        lv = oracle.MeanGrayLevels(2.0, white, 5)
        f0 = 0
        for b in (16, 1, 13):
            res = det.detect(torch.from_numpy(np.ascontiguousarray(frames[f0:f0 + b])).cuda())
            torch.cuda.synchronize()
            got = {k: getattr(res, k).cpu().numpy() for k in ("cls", "mask", "labels", "det_count", "det", "anchors", "status")}
            stats, levels = det.mean_stats[:b].cpu().numpy(), det.mean_levels[:b].cpu().numpy()
            for f in range(b):
                frame = frames[f0 + f]
                g = frame if channels == 1 else oracle.bgr2gray(frame)
                probe = oracle.MeanGrayLevels(2.0, white, 5)
                probe.levels = list(lv.levels)
                level, mean, sd, cur = probe.step(g)
                ref = oracle.detect_frame_mean_gray(frame, lv, 8192)
                assert (stats[f, 0], stats[f, 1], stats[f, 2]) == (mean, sd, cur), (h, w, f0 + f)
                assert levels[f] == min(max(level, -1), 256) and stats[f, 3] == levels[f]
                assert got["status"][f] == 0
                np.testing.assert_array_equal(got["cls"][f] & 3, ref.cls, err_msg=f"cls {h}x{w} frame {f0 + f}")
                np.testing.assert_array_equal(got["mask"][f], ref.mask)
                np.testing.assert_array_equal(got["labels"][f], ref.labels)
                assert got["det_count"][f] == ref.count
                k = ref.count
                np.testing.assert_array_equal(got["anchors"][f][:k], ref.anchors)
                np.testing.assert_array_equal(got["det"][f][:k, :4], ref.det[:, :4])
                _assert_angle_close(got["det"][f][:k, 4], ref.det[:, 4])
            f0 += b


def test_mean_gray_long_batch_and_reset(torch_cuda, oracle):
    """A batch longer than the window (the ring is overwritten within the call), then a state reset."""
    from ysmr_amd.detect import Detector, mean_gray_params
    torch = torch_cuda
    rng = np.random.default_rng(3)
    frames = np.clip(rng.normal(60, 10, (40, 24, 36)) + 25 * np.sin(np.arange(40) / 2.0)[:, None, None], 0, 255).astype(np.uint8)
    p = mean_gray_params(True, 5, 1.0)      # window 6
    det = Detector(40, 24, 36, max_det=512, params=p)
    for _ in range(2):
        lv = oracle.MeanGrayLevels(1.0, True, 5)
        det.threshold(torch.from_numpy(frames).cuda())
        det.threshold(torch.from_numpy(frames[:7]).cuda())
        torch.cuda.synchronize()
        ref = [lv.step(f)[0] for f in np.concatenate([frames, frames[:7]])]
        assert det.mean_levels[:7].cpu().tolist() == ref[40:]
        det.mean_state.reset()
    assert len(set(ref)) > 3


def test_mean_gray_bad_arguments(torch_cuda):
    from ysmr_amd import _lib
    L = _lib.lib()
    assert L.ysmr_mean_threshold_state_bytes(0) == 0 and L.ysmr_mean_threshold_state_bytes(151) == 151 * 8 + 8
    rc = L.ysmr_mean_threshold_batch(None, None, 1, 8, 8, 1, 0, 5.0, 0, None, None, None, None, 0)
    assert rc == 1 and b"window" in L.ysmr_last_error()
    rc = L.ysmr_mean_threshold_batch(None, None, 1, 8, 8, 1, 0, 5.0, 3, None, None, None, None, 0)
    assert rc == 1 and b"NULL" in L.ysmr_last_error()


def test_components_on_tiny_frames(torch_cuda, oracle):
    """Frames of fewer than 16 pixels: several frames share one 16-byte chunk of the class map."""
    from ysmr_amd.detect import Detector
    torch = torch_cuda
    rng = np.random.default_rng(4)
    for (h, w) in [(1, 1), (1, 2), (3, 2), (2, 5), (3, 5), (4, 4), (1, 17)]:
        b = 37
        cls = rng.choice(np.array([0, 1, 2, 3], np.uint8), size=(b, h, w), p=[0.4, 0.3, 0.1, 0.2])
        det = Detector(b, h, w, max_det=32)
        res = det.components(cls=torch.from_numpy(cls).cuda())
        torch.cuda.synchronize()
        for f in range(b):
            ref_mask = oracle.propagate(cls[f]).astype(bool)
            labels, rects, anchors, n = oracle.components(ref_mask.astype(np.uint8) * 255, 32)
            np.testing.assert_array_equal(res.mask[f].cpu().numpy() > 0, ref_mask, err_msg=f"{h}x{w} frame {f}")
            np.testing.assert_array_equal(res.labels[f].cpu().numpy(), labels, err_msg=f"{h}x{w} frame {f}")
            assert int(res.det_count[f]) == n and int(res.status[f]) == 0
            np.testing.assert_array_equal(res.anchors[f, :n].cpu().numpy(), anchors)
            np.testing.assert_array_equal(res.det[f, :n, :4].cpu().numpy(), rects[:, :4])


def test_mean_gray_benchmark_geometry(torch_cuda, oracle):
    """1228x922, ~500 blobs (several strips, 45-odd segments per strip): levels and detections of a few
    consecutive frames, gray and BGR."""
    from ysmr_amd.detect import Detector, mean_gray_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    gray = SyntheticVideo(922, 1228, 500, seed=4).frames(3)
    for frames in (gray, np.repeat(gray[..., None], 3, axis=3)):
        p = mean_gray_params(True, 5, 30.0)
        det = Detector(3, 922, 1228, max_det=2048, params=p)
        res = det.detect(torch.from_numpy(np.ascontiguousarray(frames)).cuda())
        torch.cuda.synchronize()
        lv = oracle.MeanGrayLevels(30.0, True, 5)
        for f in range(3):
            ref = oracle.detect_frame_mean_gray(frames[f], lv, 2048)
            assert int(res.status[f]) == 0 and int(res.det_count[f]) == ref.count > 300
            np.testing.assert_array_equal(res.cls[f].cpu().numpy() & 3, ref.cls)
            np.testing.assert_array_equal(res.labels[f].cpu().numpy(), ref.labels)
            np.testing.assert_array_equal(res.det[f, :ref.count, :4].cpu().numpy(), ref.det[:, :4])


def test_mean_gray_levels_outside_the_u8_range(torch_cuda, oracle):
    """Levels beyond [0, 255] (cv2.threshold then sets everything or nothing): saturated frames with a
    large offset, black frames with a negative one; both polarities."""
    from ysmr_amd.detect import Detector, mean_gray_params
    torch = torch_cuda
    h, w = 24, 40
    cases = [(np.full((4, h, w), 255, np.uint8), True, 5), (np.zeros((4, h, w), np.uint8), True, -5),
             (np.zeros((4, h, w), np.uint8), False, 5), (np.full((4, h, w), 255, np.uint8), False, -5)]
    for frames, white, offset in cases:
        det = Detector(4, h, w, max_det=64, params=mean_gray_params(white, offset, 30.0))
        res = det.detect(torch.from_numpy(frames).cuda())
        torch.cuda.synchronize()
        lv = oracle.MeanGrayLevels(30.0, white, offset)
        for f in range(4):
            ref = oracle.detect_frame_mean_gray(frames[f], lv, 64)
            np.testing.assert_array_equal(res.cls[f].cpu().numpy() & 3, ref.cls)
            assert int(res.det_count[f]) == ref.count and int(res.status[f]) == 0
            np.testing.assert_array_equal(res.det[f, :ref.count].cpu().numpy(), ref.det)
        levels = det.mean_levels[:4].cpu().tolist()
        assert all(v in (-1, 256) or 0 <= v <= 255 for v in levels)
    assert oracle.MeanGrayLevels(30.0, True, 5).step(np.full((h, w), 255, np.uint8))[0] == 260


def test_detect_extreme_aspect_ratios(torch_cuda, oracle):
    """The geometry limits of the C ABI (16384 x 16384): a frame as wide / as tall as allowed."""
    from ysmr_amd.detect import threshold_params
    rng = np.random.default_rng(9)
    p = threshold_params(True, 5, 2.0)
    for (h, w) in [(3, 16384), (16384, 4), (2, 16380)]:
        frames = np.full((2, h, w), 40, np.uint8) + rng.integers(0, 3, (2, h, w), dtype=np.uint8)
        ys, xs = rng.integers(0, h, 60), rng.integers(0, w, 60)
        frames[0, ys, xs] = 200
        frames[1, ys[:30], xs[:30]] = 230
        _compare(oracle, frames, _detect_gpu(torch_cuda, frames, p, max_det=512), p, max_det=512)
    from ysmr_amd import _lib
    L = _lib.lib()
    assert L.ysmr_detect_workspace_bytes(1, 16385, 8, 8) > 0          # (the size query does not validate ...)
    rc = L.ysmr_threshold_batch(None, None, 1, 16385, 8, 1, 0, 5, 7, 1, None, 0)
    assert rc == 1 and b"16384" in L.ysmr_last_error()                 # ... the calls do


def test_detection_writes_stay_inside_their_buffers(torch_cuda, oracle):
    """Every output and the workspace carved out of ONE allocation with guard bands between them: no kernel
    of the chain may write outside the buffer it was given.  The geometries are the ones in which an index
    can run past an end: more foreground than the residue list holds (count > cap: the passes walk every
    pixel and redo what k_windows settled), a batch whose size is neither a multiple of 16 nor of 4, frames
    smaller than one 64 x 64 window and rows that end inside a 16-byte chunk, repeated calls on the same buffers
    (clearing by the previous call's component boxes), and max_det smaller than the component count.
    (Round 1 recorded one unexplained GPU memory fault in an uncommitted intermediate state of these passes,
    gpurun_out/prof_v17.log; this is the test that guards the candidates: DESIGN.md section 9.)"""
    import ctypes
    torch = torch_cuda
    from ysmr_amd import _lib
    L = _lib.lib()
    GUARD, FILL = 4096, 0xA5
    rng = np.random.default_rng(11)

    def run(b, h, w, max_det, maps, from_frames):
        n = b * h * w
        ws_bytes = L.ysmr_detect_workspace_bytes(b, h, w, max_det)
        sizes = {"frames": n, "cls": n, "mask": n, "labels": 4 * n, "det_count": 4 * b, "det": 20 * b * max_det,
                 "anchors": 4 * b * max_det, "status": 4 * b, "ws": ws_bytes}
        off, pos = {}, GUARD
        for k, sz in sizes.items():
            off[k] = pos
            pos = (pos + sz + GUARD + 255) // 256 * 256
        arena = torch.full((pos,), FILL, dtype=torch.uint8, device="cuda")
        base = arena.data_ptr()
        assert base % 256 == 0
        ptr = {k: base + o for k, o in off.items()}
        for k in ("labels", "mask", "det_count", "det", "anchors", "status"):   # (what a caller's torch.zeros would hold)
            arena[off[k]:off[k] + sizes[k]] = 0
        st = _lib.stream_ptr()
        _lib.check(L.ysmr_detect_workspace_init(st, ptr["ws"], ws_bytes), "ws init")
        for m in maps:
            key = "frames" if from_frames else "cls"
            arena[off[key]:off[key] + n] = torch.from_numpy(np.ascontiguousarray(m).reshape(-1)).cuda()
            if from_frames:
                rc = L.ysmr_detect_batch(st, ptr["frames"], b, h, w, 1, 0, 5, 7, 1, ptr["ws"], ws_bytes, ptr["cls"], ptr["mask"],
                                         ptr["labels"], ptr["det_count"], ptr["det"], ptr["anchors"], max_det, ptr["status"], 0)
            else:
                rc = L.ysmr_components_batch(st, b, h, w, ptr["ws"], ws_bytes, ptr["cls"], ptr["mask"], ptr["labels"],
                                             ptr["det_count"], ptr["det"], ptr["anchors"], max_det, ptr["status"], 0)
            _lib.check(rc, "detect")
            torch.cuda.synchronize()
            host = arena.cpu().numpy()
            inside = np.zeros(len(host), bool)
            for k, sz in sizes.items():
                inside[off[k]:off[k] + sz] = True
            bad = np.flatnonzero(~inside & (host != FILL))
            owner = lambda i: max((k for k in off if off[k] <= i), key=lambda k: off[k], default="(front guard)")  # noqa: E731
            assert len(bad) == 0, f"{len(bad)} guard bytes overwritten, first at +{bad[0] - off.get(owner(bad[0]), 0)} after '{owner(bad[0])}' ({b}x{h}x{w})"
            labels = host[off["labels"]:off["labels"] + 4 * n].view(np.int32).reshape(b, h, w)
            mask = host[off["mask"]:off["mask"] + n].reshape(b, h, w)
            cls = host[off["cls"]:off["cls"] + n].reshape(b, h, w)
            for f in range(b):                         # and the result is still the oracle's
                want = oracle.propagate(cls[f] & 3)
                np.testing.assert_array_equal(mask[f] != 0, want != 0)
                np.testing.assert_array_equal(labels[f] != 0, want != 0)

    dense = lambda b, h, w: rng.choice(np.array([0, 1, 3], np.uint8), size=(b, h, w), p=[0.4, 0.3, 0.3])   # noqa: E731
    sparse = lambda b, h, w: rng.choice(np.array([0, 1, 3], np.uint8), size=(b, h, w), p=[0.97, 0.02, 0.01])  # noqa: E731
    run(3, 7, 13, 64, [dense(3, 7, 13), sparse(3, 7, 13), dense(3, 7, 13)], False)        # total 273: % 16 = 1, % 4 = 1
    run(5, 3, 5, 8, [dense(5, 3, 5), dense(5, 3, 5)], False)                               # frames of 15 px, max_det overflow
    run(2, 33, 36, 16, [sparse(2, 33, 36), sparse(2, 33, 36), dense(2, 33, 36), sparse(2, 33, 36)], False)
    from ysmr_amd.synth import SyntheticVideo
    v = SyntheticVideo(50, 68, 10, seed=2)
    run(3, 50, 68, 32, [v.frames(3), v.frames(3), rng.integers(0, 256, (3, 50, 68), dtype=np.uint8)], True)
    run(2, 9, 23, 32, [rng.integers(0, 256, (2, 9, 23), dtype=np.uint8)] * 2, True)       # tile kernel (W % 4 != 0)


@pytest.mark.parametrize("version,flags", [("4.10.0", 0), ("4.5.0", 1), ("3.4.18", 3)])
def test_opencv_flavours(torch_cuda, oracle, version, flags):
    """cv_flavour: OpenCV 3.x BGR2GRAY coefficients and the minAreaRect convention of OpenCV < 4.5.1, selected
    by the optional settings key 'opencv version' -- HIP vs the oracle's statement of the same conventions
    (adaptive and mean-gray branch, strip and tile kernels)."""
    from ysmr_amd import _lib
    from ysmr_amd.detect import Detector, mean_gray_params, threshold_params
    from ysmr_amd.synth import SyntheticVideo
    torch = torch_cuda
    assert _lib.cv_flavour_of(version) == flags and _lib.cv_flavour_of(None) == 0 and _lib.cv_flavour_of("4.5.1") == 0
    rng = np.random.default_rng(5)
    for (h, w) in ((96, 132), (50, 71)):
        gray = SyntheticVideo(h, w, 14, seed=w).frames(2)
        tint = rng.uniform(0.5, 1.0, (1, 1, 1, 3))
        frames = np.clip(gray[..., None] * tint + rng.integers(0, 9, gray.shape + (3,)), 0, 255).astype(np.uint8)
        p = threshold_params(True, 5, 2.0)
        d = Detector(2, h, w, max_det=256, params=p, cv_flavour=version)
        res = d.detect(torch.from_numpy(frames).cuda())
        torch.cuda.synchronize()
        for f in range(2):
            ref = oracle.detect_frame(frames[f], p.inv, p.t_low, p.t_high, p.use_high, 256, cv_flavour=flags)
            np.testing.assert_array_equal(res.mask[f].cpu().numpy(), ref.mask)
            n = ref.count
            assert int(res.det_count[f]) == n and n > 5
            got = res.det[f, :n].cpu().numpy()
            np.testing.assert_array_equal(got[:, :4], ref.det[:, :4])
            _assert_angle_close(got[:, 4], ref.det[:, 4])
            boxes = got[:, 3] > 0
            assert boxes.any()
            if flags & 1:
                assert np.all((got[boxes, 4] >= -90) & (got[boxes, 4] < 0))
            else:
                assert np.all((got[boxes, 4] >= 0) & (got[boxes, 4] <= 90))
        # mean-gray branch: the statistics are taken over the converted frame
        mp = mean_gray_params(True, 5, 30.0)
        d = Detector(2, h, w, max_det=256, params=mp, cv_flavour=version)
        d.threshold(torch.from_numpy(frames).cuda())
        lv = oracle.MeanGrayLevels(30.0, True, 5)
        for f in range(2):
            _, mean, sd, _ = lv.step(oracle.bgr2gray(frames[f], flags))
            assert d.mean_stats[f, 0].item() == mean and d.mean_stats[f, 1].item() == sd
    with pytest.raises(_lib.YsmrLibraryError):
        Detector(1, 8, 8, cv_flavour=32).threshold(torch.zeros(1, 8, 8, dtype=torch.uint8, device="cuda"))

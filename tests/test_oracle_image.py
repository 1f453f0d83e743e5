"""Image-half oracle (oracle/ysmr_oracle.c).

a4 is pinned against scipy.ndimage.binary_propagation (the function the reference calls).
a1/a2/a3/a5/a6 are "parity unpinned" (cv2 absent, reference has no vectors): here the restatement
is cross-checked against independent formulations (SciPy float64 filters, scipy.ndimage.label,
brute-force minimum-area search) and, opportunistically, against cv2 when it can be imported.
"""
import numpy as np
import pytest
from scipy import ndimage

from conftest import golden


def test_threshold_params(oracle):
    # defaults: white on dark, offset 5, adt 2.0 -> thresh = s-m > 5, markers = s-m > 7
    assert oracle.threshold_params(True, 5, 2.0) == (0, 5, 7, 1)
    assert oracle.threshold_params(True, 5, 2.5) == (0, 5, 7, 1)      # ceil(-7.5) = -7
    assert oracle.threshold_params(True, 5, 0.0) == (0, 5, 5, 0)
    # dark on bright: thresh = s-m <= -5, markers = s-m <= -3 (the looser set)
    assert oracle.threshold_params(False, 5, 2.0) == (1, -5, -3, 1)
    with pytest.raises(ValueError):
        oracle.threshold_params(True, 5, -1.0)


def test_propagation_golden(oracle):
    g = golden("propagation.npz")
    names = sorted({k.rsplit("_", 1)[0] for k in g.files})
    assert len(names) >= 7
    for n in names:
        cls = (g[f"{n}_mask"] > 0).astype(np.uint8) | ((g[f"{n}_markers"] > 0).astype(np.uint8) << 1)
        out = oracle.propagate(cls)
        np.testing.assert_array_equal(out * 255, g[f"{n}_out"], err_msg=n)


def test_propagation_live_scipy(oracle):
    rng = np.random.default_rng(3)
    for _ in range(5):
        mask = rng.random((61, 83)) < 0.5
        mark = rng.random((61, 83)) < 0.02
        cls = mask.astype(np.uint8) | (mark.astype(np.uint8) << 1)
        ref = ndimage.binary_propagation(mark, mask=mask)
        np.testing.assert_array_equal(oracle.propagate(cls).astype(bool), ref)


def test_blur3_exact(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    k = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]])
    ref = (ndimage.correlate(img.astype(np.int64), k, mode="mirror") + 8) >> 4  # mirror == REFLECT_101
    np.testing.assert_array_equal(oracle.blur3(img), ref.astype(np.uint8))
    one = np.full((1, 5), 7, np.uint8)
    np.testing.assert_array_equal(oracle.blur3(one), one)


def test_gauss11_kernel(oracle):
    k = oracle.gauss11()
    x = np.arange(11) - 5.0
    t = np.exp(-x * x / 8.0)
    np.testing.assert_array_equal(k, (t / t.sum()).astype(np.float32))
    assert abs(float(k.astype(np.float64).sum()) - 1) < 1e-6
    np.testing.assert_allclose(k[:6], [0.008812, 0.027144, 0.065114, 0.121649, 0.176998, 0.200565], atol=1e-6)


def test_adaptive_mean_vs_float64(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (64, 80), dtype=np.uint8)
    k = oracle.gauss11().astype(np.float64)
    f = ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="nearest")
    f = ndimage.correlate1d(f, k, axis=0, mode="nearest")
    got = oracle.adaptive_mean(img).astype(np.int64)
    ref = np.rint(f).astype(np.int64)
    bad = got != ref
    # f32 rounding may only matter where the exact value sits on a .5 rounding boundary
    assert np.all(np.abs(np.abs(f[bad] - np.floor(f[bad])) - 0.5) < 1e-3)
    assert bad.mean() < 1e-3
    flat = np.full((30, 30), 200, np.uint8)
    np.testing.assert_array_equal(oracle.adaptive_mean(flat), flat)


def _partition_equal(a, b):
    pairs = np.unique(np.stack([a.ravel(), b.ravel()], 1), axis=0)
    return len(np.unique(pairs[:, 0])) == len(pairs) == len(np.unique(pairs[:, 1]))


def test_label8_canonical(oracle):
    rng = np.random.default_rng(2)
    fg = (rng.random((48, 64)) < 0.35).astype(np.uint8)
    lab = oracle.label8(fg)
    ref, n = ndimage.label(fg, structure=np.ones((3, 3)))
    assert _partition_equal(lab, ref)
    # canonical: label - 1 == raster index of the first pixel of the component
    for v in np.unique(lab[lab > 0]):
        assert np.flatnonzero(lab.ravel() == v)[0] == v - 1


def test_components_order_and_nesting(oracle):
    fg = np.zeros((20, 24), np.uint8)
    fg[2:9, 2:9] = 1; fg[3:8, 3:8] = 0      # ring
    fg[5, 5] = 1                            # nested in the ring's hole -> skipped (RETR_EXTERNAL)
    fg[12:14, 15:20] = 1                    # plain blob
    fg[1, 20] = 1                           # single pixel, top right
    fg[18, 0] = 1                           # touches the left border
    labels, det, anchors, n = oracle.components(fg)
    assert n == 4
    # reverse raster order of first pixels
    assert list(anchors) == sorted(anchors, reverse=True)
    assert list(anchors) == [18 * 24 + 0, 12 * 24 + 15, 2 * 24 + 2, 1 * 24 + 20]
    assert labels[5, 5] == 5 * 24 + 5 + 1   # still labelled, just not a detection
    np.testing.assert_array_equal(det[3], [20, 1, 0, 0, 0])          # 1 point: size 0, angle 0
    np.testing.assert_allclose(det[1][:2], [17.0, 12.5])
    assert sorted(det[1][2:4]) == [1.0, 4.0]
    # a U-shaped component does not enclose: blob in its concavity is external
    fg = np.zeros((12, 12), np.uint8)
    fg[2:9, 2] = 1; fg[2:9, 8] = 1; fg[8, 2:9] = 1; fg[4, 5] = 1
    assert oracle.components(fg)[3] == 2


def _brute_min_area(points):
    """Minimum-area enclosing rectangle by trying every hull edge direction (float64)."""
    from scipy.spatial import ConvexHull
    pts = np.asarray(points, float)
    hull = pts[ConvexHull(pts).vertices]
    best = None
    for i in range(len(hull)):
        e = hull[(i + 1) % len(hull)] - hull[i]
        e = e / np.linalg.norm(e)
        nrm = np.array([-e[1], e[0]])
        u, v = hull @ e, hull @ nrm
        area = (u.max() - u.min()) * (v.max() - v.min())
        if best is None or area < best[0] - 1e-9:
            c = e * (u.max() + u.min()) / 2 + nrm * (v.max() + v.min()) / 2
            best = (area, c, sorted([u.max() - u.min(), v.max() - v.min()]))
    return best


def test_min_area_rect_generic(oracle):
    rng = np.random.default_rng(5)
    for trial in range(200):
        n = rng.integers(4, 40)
        # an oriented integer rod
        ang = rng.uniform(0, np.pi)
        t = rng.uniform(-4, 4, n); s = rng.uniform(-1.2, 1.2, n)
        pts = np.unique(np.rint(np.stack([30 + t * np.cos(ang) - s * np.sin(ang),
                                          30 + t * np.sin(ang) + s * np.cos(ang)], 1)).astype(np.int32), axis=0)
        if len(pts) < 3 or np.linalg.matrix_rank(pts - pts[0]) < 2:
            continue
        r = oracle.min_area_rect(pts)
        area, centre, size = _brute_min_area(pts)
        assert abs(r[2] * r[3] - area) < 1e-3 * max(area, 1)
        # every point lies inside the returned rectangle
        a = np.deg2rad(r[4]); e = np.array([np.cos(a), np.sin(a)]); nrm = np.array([-e[1], e[0]])
        d = pts - r[:2]
        assert np.all(np.abs(d @ e) <= r[2] / 2 + 1e-3) and np.all(np.abs(d @ nrm) <= r[3] / 2 + 1e-3)
        assert -90.0 <= r[4] <= 0.0 or r[4] == 0.0 or -90.0001 < r[4] < 90.0001


def test_min_area_rect_degenerate(oracle):
    np.testing.assert_array_equal(oracle.min_area_rect([[3, 4]]), [3, 4, 0, 0, 0])
    np.testing.assert_array_equal(oracle.min_area_rect([[3, 4], [6, 4], [4, 4], [5, 4]]), [4.5, 4, 3, 0, 180])
    np.testing.assert_array_equal(oracle.min_area_rect([[3, 4], [3, 6], [3, 5]]), [3, 5, 2, 0, -90])
    r = oracle.min_area_rect([[0, 0], [1, 1], [2, 2]])
    np.testing.assert_allclose(r[:4], [1, 1, 2 * np.sqrt(2), 0], rtol=1e-6)
    # 2x2 square
    r = oracle.min_area_rect([[0, 0], [1, 0], [0, 1], [1, 1]])
    np.testing.assert_allclose(r[:2], [0.5, 0.5]); assert r[2] * r[3] == 1.0


def test_detect_frame_on_synthetic(oracle):
    from ysmr_amd.synth import SyntheticVideo
    v = SyntheticVideo(200, 260, 30, seed=4)
    fr = v.next_frame()
    fd = oracle.detect_frame(fr)
    assert 20 <= fd.count <= 40
    assert set(np.unique(fd.cls)) <= {0, 1, 3}
    np.testing.assert_array_equal(fd.mask > 0, fd.labels > 0)
    # final mask == scipy's propagation of the class map's two bits
    ref = ndimage.binary_propagation((fd.cls & 2) > 0, mask=(fd.cls & 1) > 0)
    np.testing.assert_array_equal(fd.mask > 0, ref)
    # 3-channel gray is identical to single channel (BGR2GRAY is the identity when B=G=R)
    fd3 = oracle.detect_frame(np.repeat(fr[:, :, None], 3, axis=2))
    np.testing.assert_array_equal(fd3.det, fd.det)
    # dark-on-bright quirk: result is the (looser) marker set itself
    inv = oracle.detect_frame(255 - fr, *oracle.threshold_params(False, 5, 2.0))
    np.testing.assert_array_equal(inv.mask > 0, (inv.cls & 2) > 0)


def test_cv2_cross_check_if_available(oracle):
    """Opportunistic: when an OpenCV happens to be importable, the restated a1-a3 / a5 / a6 are compared with it and the
    release and the mismatch counts are printed (-s shows them); which conventions the installed release follows is
    taken from its version (ysmr_amd._lib.cv_flavour_of), so both angle conventions and both sets of gray coefficients
    are checked against the release that defines them."""
    cv2 = pytest.importorskip("cv2")
    from ysmr_amd import _lib
    from ysmr_amd.synth import SyntheticVideo
    flavour = _lib.cv_flavour_of(cv2.__version__)
    fr = SyntheticVideo(300, 400, 60, seed=9).next_frame()
    blurred = cv2.GaussianBlur(fr, (3, 3), 0)
    thresh = cv2.adaptiveThreshold(blurred, 255, cv2.ADAPTIVE_THRESH_GAUSSIAN_C, cv2.THRESH_BINARY, 11, -5)
    mark = cv2.adaptiveThreshold(blurred, 255, cv2.ADAPTIVE_THRESH_GAUSSIAN_C, cv2.THRESH_BINARY, 11, -7.0)
    fd = oracle.detect_frame(fr, cv_flavour=flavour)
    np.testing.assert_array_equal(oracle.blur3(fr), blurred)
    mism_t = int(((fd.cls & 1) > 0).__xor__(thresh > 0).sum())
    mism_m = int(((fd.cls & 2) > 0).__xor__(mark > 0).sum())
    bgr = np.random.default_rng(1).integers(0, 256, (64, 80, 3), dtype=np.uint8)
    mism_g = int((oracle.bgr2gray(bgr, flavour) != cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)).sum())
    contours = cv2.findContours(fd.mask.copy(), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_NONE)[-2]
    rects = np.array([[*r[0], *r[1], r[2]] for r in map(cv2.minAreaRect, contours)], np.float32).reshape(-1, 5)
    n = min(len(rects), len(fd.det))
    mism_r = int((np.abs(rects[:n] - fd.det[:n]) > 1e-3).any(axis=1).sum()) + abs(len(rects) - len(fd.det))
    print(f"cv2 {cv2.__version__} (cv_flavour {flavour}): mismatching pixels thresh {mism_t}, markers {mism_m}, "
          f"gray {mism_g}; detections {len(rects)} vs {len(fd.det)}, rectangles differing by more than 1e-3: {mism_r}")
    assert mism_t + mism_m <= 4 and mism_g == 0 and mism_r <= max(1, n // 50)


# ---- mean-gray branch (track_eval.py:219-253; 'adaptive double threshold' < 0) -----------------
def _literal_levels(grays, fps, white, offset_setting):
    """The reference's statements with NumPy (1,1) arrays standing in for cv2.meanStdDev's outputs."""
    offset = offset_setting if white else offset_setting * -1
    threshold_list, out = [], []
    for gray in grays:
        g = gray.astype(np.float64)
        mean, stddev = np.array([[g.mean()]]), np.array([[g.std()]])
        if white:
            cur = mean + stddev + offset
        else:
            cur = mean - stddev - offset
        threshold_list.append(cur)
        out.append((int((sum(threshold_list) / len(threshold_list)).item()), len(threshold_list)))
        if len(threshold_list) > fps * 5:
            del threshold_list[0]
    return out


@pytest.mark.parametrize("fps,white", [(2.0, True), (2.5, False), (0.1, True), (30.0, True)])
def test_mean_gray_levels(oracle, fps, white):
    rng = np.random.default_rng(int(fps * 10))
    grays = [np.clip(rng.normal(40 + 30 * np.sin(k / 5.0), 6 + k % 3, (24, 31)), 0, 255).astype(np.uint8)
             for k in range(40)]
    lv = oracle.MeanGrayLevels(fps, white, 5)
    got = [lv.step(g) for g in grays]
    ref = _literal_levels(grays, fps, white, 5)
    window = int(np.floor(fps * 5)) + 1
    assert max(n for _, n in ref) == min(window, len(grays))       # the window ysmr_mean_threshold_batch is given
    # np.mean/np.std differ from the integer-sum form in the last bits only: levels may differ where the
    # average sits within 1e-9 of an integer, nowhere else
    for (level, mean, sd, cur), (ref_level, _), g in zip(got, ref, grays):
        assert abs(mean - g.astype(np.float64).mean()) < 1e-12 and abs(sd - g.astype(np.float64).std()) < 1e-10
        assert level == ref_level
    flat = np.full((8, 8), 17, np.uint8)
    assert oracle.MeanGrayLevels.mean_stddev(flat) == (17.0, 0.0)


def test_level_classify(oracle):
    b = np.arange(256, dtype=np.uint8).reshape(16, 16)
    for level in (-1, 0, 100, 254, 255, 256):
        np.testing.assert_array_equal(oracle.level_classify(b, level, 0) != 0, b.astype(int) > level)
        np.testing.assert_array_equal(oracle.level_classify(b, level, 1) != 0, b.astype(int) <= level)
    assert set(np.unique(oracle.level_classify(b, 100, 0))) == {0, 3}


def test_mean_gray_against_cv2_when_available(oracle):
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(5)
    gray = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    mean, sd = cv2.meanStdDev(gray)
    assert oracle.MeanGrayLevels.mean_stddev(gray) == (mean.item(), sd.item())
    blurred = cv2.GaussianBlur(gray, (3, 3), 0)
    for level in (-3, 0, 90, 255, 300):
        for inv, ty in ((0, cv2.THRESH_BINARY), (1, cv2.THRESH_BINARY_INV)):
            ref = cv2.threshold(blurred, level, 255, ty)[1]
            np.testing.assert_array_equal((oracle.level_classify(blurred, level, inv) != 0) * 255, ref)


def test_detect_frame_mean_gray(oracle):
    from ysmr_amd.synth import SyntheticVideo
    vid = SyntheticVideo(120, 160, 10, seed=2)
    lv = oracle.MeanGrayLevels(30.0, True, 5)
    fd = oracle.detect_frame_mean_gray(vid.next_frame(), lv)
    assert fd.count >= 5 and fd.det.shape == (fd.count, 5)
    assert set(np.unique(fd.cls)) <= {0, 3} and np.array_equal(fd.mask != 0, fd.cls != 0)
    assert np.array_equal(fd.labels != 0, fd.mask != 0)

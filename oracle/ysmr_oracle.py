"""CPU oracle for the YSMR detect-and-link hot path -- TEST INFRASTRUCTURE, never product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  ``ysmr_amd`` (the product) never does; it fails loudly without its HIP library.

Two halves:

* image half (a1-a6): ctypes wrapper around ``oracle/ysmr_oracle.c`` (see its header for the
  restated OpenCV algorithms and the "parity unpinned" note for a1/a2/a3/a5/a6; a4 is pinned
  against ``scipy.ndimage.binary_propagation`` in tests/test_oracle_image.py);
* link half (a7-a19): NumPy/SciPy restatement of ``ysmr/tracker.py:27-230`` and
  ``ysmr/gsff.py:28-347``, pinned bit-for-bit against the imported reference by
  ``tests/golden/gen_golden.py`` (fixtures committed under tests/golden/).

The link half deliberately keeps the reference's per-track Python loop structure (it is also the
CPU baseline timed by bench.py), but holds state in flat lists/arrays instead of OrderedDicts.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np
from scipy.spatial.distance import cdist

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libysmr_oracle.so")


def build(force: bool = False) -> str:
    """Compile oracle/ysmr_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "ysmr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def _c():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f32p = ctypes.POINTER(ctypes.c_float)
        ci = ctypes.c_int
        lib.yo_bgr2gray.argtypes = [u8p, ci, ci, u8p]
        lib.yo_blur3.argtypes = [u8p, ci, ci, u8p]
        lib.yo_gauss11.argtypes = [f32p]
        lib.yo_adaptive_mean.argtypes = [u8p, ci, ci, u8p]
        lib.yo_classify.argtypes = [u8p, u8p, ctypes.c_long, ci, ci, ci, ci, u8p]
        lib.yo_propagate.argtypes = [u8p, ci, ci, u8p]
        lib.yo_label8.argtypes = [u8p, ci, ci, i32p]
        lib.yo_min_area_rect_xy.argtypes = [i32p, ci, f32p]
        lib.yo_components.argtypes = [u8p, ci, ci, i32p, f32p, i32p, ci]
        lib.yo_components.restype = ci
        lib.yo_detect_frame.argtypes = [u8p, ci, ci, ci, ci, ci, ci, ci, u8p, u8p, i32p, f32p,
                                        i32p, ci]
        lib.yo_detect_frame.restype = ci
        for f in (lib.yo_bgr2gray, lib.yo_blur3, lib.yo_gauss11, lib.yo_adaptive_mean,
                  lib.yo_classify, lib.yo_propagate, lib.yo_label8, lib.yo_min_area_rect_xy):
            f.restype = None
        _lib = lib
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


# ------------------------------------------------------------------------------------------------
# image half
# ------------------------------------------------------------------------------------------------
def threshold_params(white_on_dark: bool, offset, adt):
    """Integer form of the two cv2.adaptiveThreshold calls (ysmr/track_eval.py:127-132, 185-208).

    cv2 computes ``idelta = ceil(C)`` (THRESH_BINARY) or ``floor(C)`` (THRESH_BINARY_INV) and sets a
    pixel iff ``s - m > -idelta`` resp. ``s - m <= -idelta``.  The reference negates
    the offset for dark-on-bright videos before use (track_eval.py:132).
    Returns (inv, t_low, t_high, use_high) with use_high False when adt == 0;
    adt < 0 selects the mean-gray branch (MeanGrayLevels below) and raises here.
    """
    if adt < 0:
        raise ValueError("adaptive double threshold < 0 selects the mean-gray branch: use MeanGrayLevels")
    inv = not white_on_dark
    off = -offset if inv else offset
    c1 = off * -1
    c2 = (off + adt) * -1
    if inv:
        t_low, t_high = -math.floor(c1), -math.floor(c2)
    else:
        t_low, t_high = -math.ceil(c1), -math.ceil(c2)
    return int(inv), int(t_low), int(t_high), int(adt > 0)


def gauss11():
    k = np.zeros(11, np.float32)
    _c().yo_gauss11(_p(k, ctypes.c_float))
    return k


CV_ANGLE_PRE451, CV_GRAY_3X = 1, 2     # cv_flavour bits (include/ysmr_hip.h)


def bgr2gray(bgr, cv_flavour=0):
    """cv2.cvtColor(BGR2GRAY) on u8: OpenCV 4.x 15-bit fixed point (the C restatement) or, with CV_GRAY_3X,
    OpenCV 3.x's 14-bit coefficients (upstream-recollection)."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    if cv_flavour & CV_GRAY_3X:
        v = bgr.astype(np.int64)
        return ((v[..., 0] * 1868 + v[..., 1] * 9617 + v[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)
    h, w = bgr.shape[:2]
    out = np.empty((h, w), np.uint8)
    _c().yo_bgr2gray(_p(bgr, ctypes.c_uint8), h, w, _p(out, ctypes.c_uint8))
    return out


def blur3(gray):
    gray = np.ascontiguousarray(gray, np.uint8)
    out = np.empty_like(gray)
    _c().yo_blur3(_p(gray, ctypes.c_uint8), gray.shape[0], gray.shape[1], _p(out, ctypes.c_uint8))
    return out


def adaptive_mean(blurred):
    blurred = np.ascontiguousarray(blurred, np.uint8)
    out = np.empty_like(blurred)
    _c().yo_adaptive_mean(_p(blurred, ctypes.c_uint8), blurred.shape[0], blurred.shape[1],
                          _p(out, ctypes.c_uint8))
    return out


def classify(blurred, mean, inv, t_low, t_high, use_high):
    out = np.empty_like(blurred)
    _c().yo_classify(_p(blurred, ctypes.c_uint8), _p(mean, ctypes.c_uint8), blurred.size, inv,
                     t_low, t_high, use_high, _p(out, ctypes.c_uint8))
    return out


def propagate(cls):
    cls = np.ascontiguousarray(cls, np.uint8)
    out = np.empty_like(cls)
    _c().yo_propagate(_p(cls, ctypes.c_uint8), cls.shape[0], cls.shape[1], _p(out, ctypes.c_uint8))
    return out


def label8(fg):
    fg = np.ascontiguousarray(fg, np.uint8)
    out = np.empty(fg.shape, np.int32)
    _c().yo_label8(_p(fg, ctypes.c_uint8), fg.shape[0], fg.shape[1], _p(out, ctypes.c_int32))
    return out


def min_area_rect(points_xy):
    """points_xy: (n, 2) integer pixel coordinates -> (cx, cy, w, h, angle) float32."""
    xy = np.ascontiguousarray(points_xy, np.int32).reshape(-1, 2)
    rect = np.zeros(5, np.float32)
    _c().yo_min_area_rect_xy(_p(xy, ctypes.c_int32), xy.shape[0], _p(rect, ctypes.c_float))
    return rect


def components(fg, max_det=None):
    fg = np.ascontiguousarray(fg, np.uint8)
    h, w = fg.shape
    if max_det is None:
        max_det = h * w // 2 + 1
    labels = np.empty((h, w), np.int32)
    det = np.zeros((max_det, 5), np.float32)
    anchors = np.zeros(max_det, np.int32)
    n = _c().yo_components(_p(fg, ctypes.c_uint8), h, w, _p(labels, ctypes.c_int32),
                           _p(det, ctypes.c_float), _p(anchors, ctypes.c_int32), max_det)
    m = min(n, max_det)
    return labels, det[:m].copy(), anchors[:m].copy(), n


@dataclass
class FrameDetections:
    cls: np.ndarray
    mask: np.ndarray
    labels: np.ndarray
    det: np.ndarray       # (M, 5) float32: cx, cy, w, h, angle
    anchors: np.ndarray   # (M,) int32
    count: int


def rect_convention(det, cv_flavour=0):
    """cv2.minAreaRect's result as OpenCV < 4.5.1 reports it (CV_ANGLE_PRE451).  The C restatement's hull order
    and rotating calipers give angles in [0, 90] -- the range OpenCV documents from 4.5.1 on; earlier releases
    report the same rectangle with its angle in [-90, 0) and width / height named the other way round, 90
    becoming -90 with the sides as they are (upstream-recollection; rectangles of 1 or 2 hull
    points, h == 0, are left as they are)."""
    det = np.array(det, np.float32, copy=True)
    if cv_flavour & CV_ANGLE_PRE451 and len(det):
        box = det[:, 3] > 0
        axis = box & (det[:, 4] == np.float32(90))
        turn = box & ~axis
        det[axis, 4] = -90
        det[turn, 2], det[turn, 3] = det[turn, 3].copy(), det[turn, 2].copy()
        det[turn, 4] = det[turn, 4] - np.float32(90)
    return det


def detect_frame(frame, inv=0, t_low=5, t_high=7, use_high=1, max_det=None, cv_flavour=0) -> FrameDetections:
    """The image half of one loop iteration (track_eval.py:180-303) on an (H,W) or (H,W,3) frame."""
    frame = np.ascontiguousarray(frame, np.uint8)
    if cv_flavour:
        gray = frame if frame.ndim == 2 else bgr2gray(frame, cv_flavour)
        fd = detect_frame(gray, inv, t_low, t_high, use_high, max_det)
        fd.det = rect_convention(fd.det, cv_flavour)
        return fd
    h, w = frame.shape[:2]
    ch = 1 if frame.ndim == 2 else frame.shape[2]
    if max_det is None:
        max_det = 65536
    cls = np.empty((h, w), np.uint8)
    mask = np.empty((h, w), np.uint8)
    labels = np.empty((h, w), np.int32)
    det = np.zeros((max_det, 5), np.float32)
    anchors = np.zeros(max_det, np.int32)
    n = _c().yo_detect_frame(_p(frame, ctypes.c_uint8), h, w, ch, inv, t_low, t_high, use_high,
                             _p(cls, ctypes.c_uint8), _p(mask, ctypes.c_uint8),
                             _p(labels, ctypes.c_int32), _p(det, ctypes.c_float),
                             _p(anchors, ctypes.c_int32), max_det)
    m = min(n, max_det)
    return FrameDetections(cls, mask, labels, det[:m].copy(), anchors[:m].copy(), n)


class MeanGrayLevels:
    """The mean-gray branch's threshold level (ysmr/track_eval.py:219-242, taken when
    'adaptive double threshold' < 0).  Per frame: cv2.meanStdDev(gray) -> mean +- stddev +- offset
    appended to ``threshold_list``; the level is ``int(sum(list) / len(list))``; the list is
    trimmed AFTER that, when it is longer than 5 * fps.  cv2.meanStdDev on u8 [upstream-
    recollection]: exact integer sums, then ``mean = s * (1/N)``,
    ``stddev = sqrt(max(sq * (1/N) - mean * mean, 0))`` in double."""

    def __init__(self, fps, white_on_dark=True, offset=5):
        self.fps = fps
        self.white = bool(white_on_dark)
        self.offset = offset if self.white else offset * -1   # track_eval.py:132
        self.levels = []

    @staticmethod
    def mean_stddev(gray):
        g = np.asarray(gray, np.uint8).astype(np.int64)
        s, sq = int(g.sum()), int((g * g).sum())
        scale = 1.0 / g.size
        mean = s * scale
        return mean, math.sqrt(max(sq * scale - mean * mean, 0.0))

    def step(self, gray):
        """-> (integer level, mean, stddev, this frame's level)"""
        mean, sd = self.mean_stddev(gray)
        cur = (mean + sd + self.offset) if self.white else (mean - sd - self.offset)
        self.levels.append(cur)
        acc = 0
        for v in self.levels:          # Python's sum(): left to right, starting from int 0
            acc = acc + v
        level = int(acc / len(self.levels))
        if len(self.levels) > self.fps * 5:
            del self.levels[0]
        return level, mean, sd, cur


def level_classify(blurred, level, inv):
    """cv2.threshold(blurred, level, 255, THRESH_BINARY / THRESH_BINARY_INV) (track_eval.py:248-253)
    as a class map: 3 (thresh and marker bit) where set.  u8 source: dst = src > level (BINARY)."""
    fg = np.asarray(blurred, np.uint8).astype(np.int32) > int(level)
    if inv:
        fg = ~fg
    return np.where(fg, 3, 0).astype(np.uint8)


def detect_frame_mean_gray(frame, levels: MeanGrayLevels, max_det=None) -> FrameDetections:
    """One loop iteration of the mean-gray branch (track_eval.py:180-182, 219-253, 273-303)."""
    frame = np.ascontiguousarray(frame, np.uint8)
    gray = frame if frame.ndim == 2 else bgr2gray(frame)
    level, _, _, _ = levels.step(gray)
    cls = level_classify(blur3(gray), level, not levels.white)
    mask = np.where(cls != 0, 255, 0).astype(np.uint8)
    labels, det, anchors, n = components(mask, max_det if max_det is not None else 65536)
    return FrameDetections(cls, mask, labels, det, anchors, n)


def det_to_rects(det):
    """(M,5) float32 -> the reference's rects list [((x, y), (w, h, deg)), ...]
    (reshape_result, ysmr/helper_file.py:1336-1347); values become Python floats like cv2's."""
    return [((float(d[0]), float(d[1])), (float(d[2]), float(d[3]), float(d[4]))) for d in det]


# ------------------------------------------------------------------------------------------------
# link half: GaussianSumFIR (ysmr/gsff.py) and CentroidTracker (ysmr/tracker.py)
# ------------------------------------------------------------------------------------------------
def horizon_sizes(n_min, n_max, n_f):
    """gsff.py:87-109: n_i = int(n_min + p*i), p = (n_max - n_min)/n_f, i = 1..n_f."""
    step = (n_max - n_min) / n_f
    return [int(n_min + step * i) for i in range(1, n_f + 1)]


def lsf_gain(size, dt, a=None, c=None):
    """gsff.py:111-153: gain = (L^T L)^-1 L^T with L = H_bar A^-N, H_bar = [C; CA; ...; CA^(N-1)]."""
    if a is None:
        a = np.array([[1, 0, dt, 0], [0, 1, 0, dt], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    if c is None:
        c = np.array([[1, 0, 0, 0], [0, 1, 0, 0]])
    rows = c
    power = a
    for _ in range(size - 1):
        rows = np.concatenate((rows, np.dot(c, power)), axis=0)
        power = np.dot(power, a)
    big_l = np.dot(rows, np.linalg.matrix_power(np.linalg.inv(a), size))
    return np.dot(np.linalg.inv(np.dot(big_l.T, big_l)), big_l.T)


@dataclass
class GsffState:
    """Per-track filter memory (the kwargs dict the reference threads through correct/predict)."""
    mode: int = 0
    history: list | None = None          # list of measurement vectors, newest last
    weights: np.ndarray | None = None
    likelihoods: list | None = None
    x_hat: np.ndarray | None = None      # (dim, mode)


class OracleGSFF:
    def __init__(self, delta_t, n_min=0, n_max=30, n_f=3, a=None, c=None,
                 likelihood_minimum=10 ** -20, inv_cov=None, x_hat_array_length=2, perturb=0):
        # perturb = +1 / -1: a SHADOW filter whose FIR outputs and likelihoods are moved by one ulp,
        # in opposite directions for even and odd filters (see OracleTracker, ``shadows``); 0 = the
        # reference's arithmetic, pinned by tests/golden/
        self.perturb = perturb
        self.lik_min = likelihood_minimum
        self.dim = x_hat_array_length
        self.n_f = n_f
        self.n_i = horizon_sizes(n_min, n_max, n_f)
        self.gains = [lsf_gain(n, delta_t, a, c) for n in self.n_i]
        self.inv_cov = np.linalg.inv(np.eye(2)) if inv_cov is None else inv_cov

    # gsff.py:156-177
    def _nudge(self, v, idx):
        up = (idx % 2 == 0) == (self.perturb > 0)
        return np.nextafter(v, np.inf if up else -np.inf)

    def _fir(self, idx, history):
        n = self.n_i[idx]
        flat = [v for m in history[-n:] for v in m]
        out = np.dot(self.gains[idx], flat)
        return self._nudge(out, idx) if self.perturb else out

    # gsff.py:179-202 (the FloatingPointError handlers are dead code under NumPy defaults)
    def _likelihood(self, z, y_hat, idx=0):
        d = z - y_hat
        lik = np.exp(-0.5 * np.dot(d.T, np.dot(self.inv_cov, d)))
        if self.perturb:
            lik = self._nudge(lik, idx + 1)
        return self.lik_min if lik < self.lik_min else lik

    def _refresh(self, st: GsffState):
        for i in range(st.mode):
            st.x_hat[:, i] = self._fir(i, st.history)[: self.dim]

    # gsff.py:204-249
    def predict(self, st: GsffState):
        if st.history is None:
            return None
        self._refresh(st)
        return np.sum(st.x_hat * st.weights, axis=1)

    # gsff.py:251-347
    def correct(self, z, st: GsffState):
        if st.history is None:
            st.history = [z] * self.n_i[0]
        grew = False
        if st.mode < self.n_f:
            while len(st.history) >= self.n_i[st.mode]:
                st.mode += 1
                grew = True
                if st.mode >= self.n_f:
                    break
        if grew:
            st.likelihoods = [self.lik_min] * st.mode
            st.x_hat = np.zeros((self.dim, st.mode))
            st.weights = 1 / st.mode * np.ones(st.mode)
            self._refresh(st)
        for i in range(st.mode):
            st.likelihoods[i] = self._likelihood(z, st.x_hat[:, i][:2], i)
        st.history.append(z)
        keep = self.n_i[-1] + 1
        if len(st.history) > keep:
            st.history = st.history[-keep:]
        total = sum(st.likelihoods * st.weights)
        for i in range(st.weights.shape[0]):
            st.weights[i] = st.likelihoods[i] * st.weights[i] / total
        return np.sum(st.x_hat * st.weights, axis=1)


@dataclass
class _Track:
    tid: int
    pos: np.ndarray
    info: object
    gone: int = 0
    gs: GsffState = field(default_factory=GsffState)
    shadow: list = field(default_factory=list)   # [[GsffState, pos], ...] one per shadow filter
    tied: bool = False    # its assignment once hung on a distance tie (see OracleTracker, ``shadows``)


class OracleTracker:
    """Restatement of CentroidTracker (ysmr/tracker.py:27-230).

    ``update(rects)`` returns ``(ids, xy, info, claims)``: ids in dict-iteration order of the
    reference (= ascending id), xy the filtered (or raw) positions (n,2) float64, info the list
    of (w,h,deg) tuples or [0,0,0] lists, claims the list of (row, col) pairs accepted this frame.
    Tie order of equal row minima: stable (min, row) -- the reference's default argsort is
    unspecified there.  New-track id order follows CPython set iteration
    (see update()); product code reproduces it with an explicit model of CPython's set table.

    ``shadows`` (0 or 2): conditioning probe for the tests.  Beside every track's filter run two
    shadow filters that see the same detections and the same claims, but whose FIR outputs and
    likelihoods are moved by ONE ULP (in opposite directions for even and odd filters, and oppositely
    in the two shadows), and which, while the track is lost, are fed their OWN prediction just as the
    real filter is fed its own (tracker.py:219-225).  ``last_sens[i]`` = largest relative deviation
    |shadow - real| / max(1, |real|) of track i's output this frame.  A row whose value moves by more
    than ILL_CONDITIONED under a one-ulp change of the reference's own intermediate results is not
    determined by the reference's arithmetic to better than that (any other BLAS, libm or summation
    order moves it as much); every other row is.  The real filter's arithmetic is untouched.

    The probe also watches the ASSIGNMENT: a track whose claim hangs on a distance tie -- two tracks equally
    far (to TIE relative) from the detection both want, e.g. two blobs that merge into one component exactly
    between them, or two detections equally far from one track -- is assigned by the last bits of the
    predicted positions, i.e. by how LAPACK rounded the reference's gain matrices (their rows sum to 1 + 1e-15,
    which moves x = 2996 to 2996.0000000000036).  Such tracks are marked ``tied`` for the rest of their
    life and report ``last_sens = inf``: which of the two got the detection is not a property of the algorithm.
    """

    ILL_CONDITIONED = 1e-12
    TIE = 1e-9

    def __init__(self, max_disappeared=50, fps=30, n_min=0, n_max=None, n_f=3, use_gsff=True, shadows=0):
        self.max_gone = max_disappeared
        self.use_gsff = use_gsff
        self.next_id = 0
        self.tracks: list[_Track] = []
        self.shadow_gsff = []
        self.last_sens = np.zeros(0)
        if use_gsff:
            if n_max is None:
                n_max = fps
            kw = dict(delta_t=1 / fps, n_min=n_min, n_max=n_max, n_f=n_f, likelihood_minimum=10 ** -20,
                      inv_cov=np.linalg.inv(np.eye(2)), x_hat_array_length=2)
            self.gsff = OracleGSFF(**kw)
            if shadows:
                self.shadow_gsff = [OracleGSFF(perturb=+1, **kw), OracleGSFF(perturb=-1, **kw)]

    def _mark_ties(self, d):
        """Tracks whose claim this frame depends on a distance tie (conditioning probe only)."""
        n, m = d.shape
        dmin, arg = d.min(axis=1), d.argmin(axis=1)
        tol = self.TIE * np.maximum(1.0, dmin)
        if m > 1:    # two detections equally near one track
            second = np.partition(d, 1, axis=1)[:, 1]
            for r in np.flatnonzero(second - dmin <= tol):
                self.tracks[r].tied = True
        by_col = np.argsort(arg, kind="stable")          # tracks proposing the same detection
        cols = arg[by_col]
        start = 0
        for end in list(np.flatnonzero(cols[1:] != cols[:-1]) + 1) + [n]:
            if end - start > 1:
                rows = by_col[start:end]
                best = dmin[rows].min()
                close = rows[dmin[rows] - best <= self.TIE * max(1.0, best)]
                if len(close) > 1:
                    for r in close:
                        self.tracks[r].tied = True
            start = end

    def _register(self, pos, info):
        self.tracks.append(_Track(self.next_id, pos, info, shadow=[[GsffState(), pos] for _ in self.shadow_gsff]))
        self.next_id += 1

    def _age(self, tr: _Track):
        """tracker.py:100-107 / 204-211; returns True when the track must be dropped."""
        tr.gone += 1
        tr.info = [0] * len(tr.info)
        return tr.gone > self.max_gone

    def update(self, rects):
        claims = []
        if len(rects) == 0:
            self.tracks = [t for t in self.tracks if not self._age(t)]
        else:
            pts = np.zeros((len(rects), len(rects[0][0])), dtype="float")
            infos = []
            for i, (xy, info) in enumerate(rects):
                pts[i] = xy
                infos.append(info)
            if not self.tracks:
                for i in range(len(pts)):
                    self._register(pts[i], infos[i])
            else:
                cur = np.array([t.pos for t in self.tracks])
                d = cdist(cur, pts)
                order = np.argsort(d.min(axis=1), kind="stable")
                nearest = d.argmin(axis=1)[order]
                if self.shadow_gsff:
                    self._mark_ties(d)
                rows_used, cols_used = set(), set()
                for r, c in zip(order, nearest):
                    if r in rows_used or c in cols_used:
                        continue
                    tr = self.tracks[r]
                    tr.pos = pts[c]
                    for sh in tr.shadow:
                        sh[1] = pts[c]
                    tr.info = infos[c]
                    tr.gone = 0
                    rows_used.add(r)
                    cols_used.add(c)
                    claims.append((int(r), int(c)))
                n_tr, n_det = d.shape
                if n_tr >= n_det:
                    dead = set()
                    for r in range(n_tr):
                        if r not in rows_used and self._age(self.tracks[r]):
                            dead.add(r)
                    if dead:
                        self.tracks = [t for i, t in enumerate(self.tracks) if i not in dead]
                else:
                    # tracker.py:193,216 iterates a Python *set*: new ids are handed out in
                    # CPython's hash-table order of set(range(M)).difference(used), which is NOT
                    # ascending in general (e.g. {5, 40} iterates 40, 5).  Restated literally.
                    for c in set(range(0, n_det)).difference(cols_used):
                        self._register(pts[c], infos[c])
        ids = [t.tid for t in self.tracks]
        info = [t.info for t in self.tracks]
        if self.use_gsff:
            out = np.zeros((len(self.tracks), 2))
            sens = np.zeros(len(self.tracks))
            for i, t in enumerate(self.tracks):
                out[i] = self.gsff.correct(t.pos, t.gs)
                t.pos = self.gsff.predict(t.gs)
                for g, sh in zip(self.shadow_gsff, t.shadow):
                    o = g.correct(sh[1], sh[0])
                    sh[1] = g.predict(sh[0])
                    sens[i] = max(sens[i], float(np.max(np.abs(o - out[i]) / np.maximum(1.0, np.abs(out[i])))))
                if t.tied:
                    sens[i] = np.inf
            self.last_sens = sens
        else:
            out = np.array([t.pos for t in self.tracks], dtype=float).reshape(-1, 2)
        return ids, out, info, claims


# ------------------------------------------------------------------------------------------------
# whole path: frames -> rows, mirroring the body of track_bacteria (track_eval.py:156-366)
# ------------------------------------------------------------------------------------------------
def track_frames(frames, fps=30.0, white_on_dark=True, offset=5, adt=2.0, use_gsff=True,
                 n_min=0, n_max=30, n_f=3, max_det=65536, tracker=None, frame0=0, shadows=0):
    """Run detect+link over an iterable of frames.  Returns (rows, tracker) with rows a list of
    (frame, id, x, y, w, h, deg) -- one per live track per frame (track_eval.py:313-316).
    ``shadows=2``: every row gets an eighth entry, OracleTracker's conditioning probe ``last_sens``."""
    mean_gray = MeanGrayLevels(fps, white_on_dark, offset) if adt < 0 else None
    if mean_gray is None:
        inv, t_low, t_high, use_high = threshold_params(white_on_dark, offset, adt)
    if tracker is None:
        tracker = OracleTracker(max_disappeared=fps, fps=fps, n_min=n_min, n_max=n_max, n_f=n_f,
                                use_gsff=use_gsff, shadows=shadows)
    rows = []
    for k, frame in enumerate(frames):
        if mean_gray is None:
            fd = detect_frame(frame, inv, t_low, t_high, use_high, max_det)
        else:
            fd = detect_frame_mean_gray(frame, mean_gray, max_det)
        ids, xy, info, _ = tracker.update(det_to_rects(fd.det))
        for i, tid in enumerate(ids):
            w, h, deg = info[i]
            row = (frame0 + k, tid, float(xy[i][0]), float(xy[i][1]), float(w), float(h), float(deg))
            rows.append(row + (float(tracker.last_sens[i]),) if tracker.shadow_gsff else row)
    return rows, tracker


# ------------------------------------------------------------------------------------------------
# the frame source's protocol (f1): what track_bacteria does with the frame count a container REPORTS and the
# frames cap.read() actually DELIVERS (ysmr/track_eval.py:73-93, 156-178, 368-405)
# ------------------------------------------------------------------------------------------------
def reader_protocol(reported_frames, delivered_frames, fps_of_container, settings):
    """Restates the control flow around cv2.VideoCapture in track_bacteria for a container that reports
    ``reported_frames`` (CAP_PROP_FRAME_COUNT) and whose cap.read() succeeds ``delivered_frames`` times.
    Returns a dict: skipped (returned None before reading), frames_processed, read_error (the critical log of :176),
    returns_none (the function's result is None), fps (what the tracker is built with, None when skipped)."""
    out = {"skipped": False, "frames_processed": 0, "read_error": False, "returns_none": False, "fps": None}
    if int(reported_frames) < settings["minimal frame count"]:                        # :73-77
        out.update(skipped=True, returns_none=True)
        return out
    if not settings["force tracking.ini fps settings"]:                               # :78-93
        fps = fps_of_container
    else:
        fps = settings["frames per second"]
    out["fps"] = fps
    curr, error_during_read = 0, False
    while True:                                                                       # :156
        ret = curr < delivered_frames                                                 # cap.read() :159
        if not ret and (reported_frames == curr + 1 or reported_frames == curr) and \
                reported_frames >= settings["minimal frame count"]:                   # :170-174
            break
        elif not ret:                                                                 # :175-178
            out["read_error"] = True
            error_during_read = settings["stop evaluation on error"]
            break
        curr += 1                                                                     # :348
    out["frames_processed"] = curr
    if curr == 0:
        out["returns_none"] = True                                                    # no objects: :388-392
    if error_during_read:                                                             # :402-404
        out["returns_none"] = True
    return out


# ------------------------------------------------------------------------------------------------
# selection of good tracks: select_tracks / find_good_tracks (ysmr/track_eval.py:408-843)
# ------------------------------------------------------------------------------------------------
def select_tracks_oracle(df, settings, fps, frame_height, frame_width):
    """CPU restatement of the reference's track selection on a (TRACK_ID, POSITION_T)-ordered
    DataFrame.  Uses the same pandas / numpy primitives as the reference (groupby median, Series
    quantile, Series mean, diff) so that their arithmetic -- median_linear, numpy's linear
    percentile, pairwise summation -- is the reference's by construction; the control flow is written
    out anew: clean-up (track_eval.py:623-684), bounds (:693-733), per-track splitting (:408-533),
    choice of the longest fragment and the length limit (:745-796), result table (:822-843).

    Returns (selected DataFrame or None, info dict).  info['status']: 0 ok, 1 too short, 2 too short
    after the clean-up, 3 no acceptable track."""
    info = {"status": 0, "kick_reasons": [0] * 9}
    min_len = int(round(fps, 0) * settings["minimal length in seconds"])
    limit = int(round(fps, 0) * settings["limit track length to x seconds"])
    t = df.copy()
    info["rows_before"] = len(t)
    if len(t) < min_len or len(t) == 0:
        info["status"] = 1
        return None, info
    info["tracks_before"] = int(t["TRACK_ID"].nunique())

    # ---- clean-up: every rejected measurement gets a NaN area, rows with NaN area are dropped
    by_track = t.groupby("TRACK_ID")
    area = t["WIDTH"] * t["HEIGHT"]
    typical = area.groupby(t["TRACK_ID"]).transform("median")
    ok = (typical >= settings["extreme area outliers lower end in px*px"]) & \
         (typical <= settings["extreme area outliers upper end in px*px"])
    if settings["exclude measurement when above x times average area"]:
        ok &= area <= typical * settings["exclude measurement when above x times average area"]
    ok &= area != 0
    span = (by_track["POSITION_T"].transform("last") - by_track["POSITION_T"].transform("first") + 1).astype(np.uint16)
    ok &= span >= min_len
    t["area"] = area
    t = t[ok].reset_index(drop=True)
    info["rows_after"] = len(t)
    if len(t) < min_len or len(t) == 0:
        info["status"] = 2
        return None, info
    ids = t["TRACK_ID"].to_numpy()
    first_rows = np.flatnonzero(np.r_[True, ids[1:] != ids[:-1]])
    last_rows = np.r_[first_rows[1:] - 1, len(t) - 1]
    info["tracks_after"] = len(first_rows)

    # ---- bounds
    t["ratio_wh"] = np.where(t["HEIGHT"] <= t["WIDTH"], t["HEIGHT"] / t["WIDTH"], t["WIDTH"] / t["HEIGHT"])
    q = settings["percent quantiles excluded area"]
    if q > 0:
        lower, upper = t["area"].quantile(q=[q, 1 - q])
    else:
        lower, upper = -1, np.inf
    info["area_lo"], info["area_hi"] = float(lower), float(upper)
    flagged = np.zeros(len(t), np.int8)
    info.update(q1_dist=0.0, q3_dist=0.0, dist_fence=0.0, dist_outliers=0, outliers_used=0)
    if settings["try to omit motility outliers"]:
        step = np.sqrt(np.square(t["POSITION_X"].diff()) + np.square(t["POSITION_Y"].diff())) / t["POSITION_T"].diff()
        step[first_rows] = 0
        q1, q3 = step.quantile(q=[0.25, 0.75])
        fence = (q3 - q1) * 3 + q3
        flagged = np.where(step > fence, 1, 0).astype(np.int8)
        share = flagged.sum() / len(t)
        info.update(q1_dist=float(q1), q3_dist=float(q3), dist_fence=float(fence), dist_outliers=int(flagged.sum()),
                    outliers_used=1)
        if share > settings["stop excluding motility outliers if total count above percent"]:
            flagged = np.zeros(len(t), np.int8)
            info["outliers_used"] = 0
    t["distance"] = flagged

    # ---- per track: split at holes / outliers until a fragment passes every test
    times = t["POSITION_T"]
    edge = settings["percent of screen edges to exclude"]
    short = 3 if min_len < 3 else min_len

    def examine(lo, hi, depth):
        """-> (passing fragments in table order, lowest rejection stage reached)"""
        stage, found, parts = 8, [], []
        if hi - lo + 1 >= min_len:
            stage = 7
            piece = t.iloc[lo:hi + 1]
            steps = piece["POSITION_T"].diff()
            if steps.max() <= settings["maximal consecutive holes"]:
                stage = 6
                if piece["distance"].sum() == 0:
                    stage = 5
                    duration = piece["POSITION_T"].iloc[-1] - piece["POSITION_T"].iloc[0] + 1
                    if duration / len(piece) < settings["maximal empty frames in %"]:
                        stage = 4
                        if lower <= piece["area"].mean() <= upper:
                            stage = 3
                            if settings["average width/height ratio min."] < piece["ratio_wh"].mean() \
                                    < settings["average width/height ratio max."]:
                                stage = 2
                                if edge * frame_height < piece["POSITION_Y"].mean() < (1 - edge) * frame_height and \
                                        edge * frame_width < piece["POSITION_X"].mean() < (1 - edge) * frame_width:
                                    stage = 1
                                    outside = (piece["POSITION_X"].min() < 0 or piece["POSITION_X"].max() > frame_width or
                                               piece["POSITION_Y"].min() < 0 or piece["POSITION_Y"].max() > frame_height)
                                    if edge == 0 or not outside:
                                        stage = 0
                                        found.append((lo, hi))
                else:
                    at = int(piece["distance"].idxmax())          # first flagged row; it is left out
                    parts = [(lo, at - 1), (at + 1, hi)]
            elif len(piece) >= 2:
                at = int(steps.idxmax())                          # first row after the largest hole
                parts = [(lo, at - 1), (at, hi)]
        if parts and depth < settings["maximal recursion depth"]:
            for a, b in parts:
                if b - a + 1 < short:
                    continue
                sub_found, sub_stage = examine(a, b, depth + 1)
                found.extend(sub_found)
                stage = min(stage, sub_stage)
        return found, stage

    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 4000))
    keep = np.zeros(len(t), bool)
    n_good = 0
    for lo, hi in zip(first_rows, last_rows):
        found, stage = examine(int(lo), int(hi), 0)
        info["kick_reasons"][stage] += 1
        if not found:
            continue
        a, b = max(found, key=lambda r: (r[1] - r[0], -r[0]))       # the longest, the earliest among equals
        if limit:
            until = limit + int(times.iloc[a]) - 1
            window = times.iloc[a:b + 1]
            hit = window[window == until] if settings["limit track length exactly"] else window[window <= until]
            if len(hit) == 0:
                continue
            b = int(hit.index[-1])
        keep[a:b + 1] = True
        n_good += 1
    info["good_tracks"] = n_good
    info["rows_selected"] = int(keep.sum())
    if not keep.any():
        info["status"] = 3
        return None, info
    cols = ["TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE"]
    out = t.loc[keep, cols].reset_index()      # keeps the cleaned table's index as column 'index'
    return out, info


# ------------------------------------------------------------------------------------------------
# per-track statistics: evaluate_tracks (ysmr/track_eval.py:846-1318), the numerical part
# ------------------------------------------------------------------------------------------------
EVAL_STATS_COLUMNS = ["Turn Points (TP/s)", "Distance (µm)", "Speed (µm/s)", "Time (s)", "Displacement (µm)",
                      "Perc. Motile", "Arc-Chord Ratio", "Bacteria Length", "Displacement divided by length",
                      "Motility Phenotype", "TRACK_ID", "Median Speed"]
EVAL_ROW_COLUMNS = ["TRACK_ID", "POSITION_T", "POSITION_X", "POSITION_Y", "WIDTH", "HEIGHT", "DEGREES_ANGLE", "angle_diff",
                    "moving", "turn_points", "tp_of_tracks", "travelled_dist", "motility_phenotype"]


def evaluate_tracks_oracle(df, settings, fps):
    """CPU restatement of the arithmetic of ``evaluate_tracks`` on a table of selected tracks (ordered by
    TRACK_ID, POSITION_T, default RangeIndex): the per-row columns of ``*_analysed.csv`` and the per-track
    table of ``*_statistics.csv``.  Parity unpinned, like select_tracks_oracle: track_eval.py imports cv2 at
    module level, so the reference's own function cannot be run here and it ships no vectors.  The same pandas /
    NumPy / SciPy primitives are used where their arithmetic matters (groupby sum / mean = Kahan summation,
    float16 ``bac_length`` whose group mean pandas returns as float32, ``scipy.signal.medfilt``,
    ``scipy.signal.argrelextrema``, ``scipy.spatial.distance.pdist``), the control flow is written out anew.
    Plots, logging and file output are not part of it.  Returns (rows DataFrame, statistics DataFrame)."""
    import pandas as pd
    from scipy.signal import argrelextrema, medfilt
    from scipy.spatial.distance import pdist
    t = df.reset_index(drop=True).copy()
    n = len(t)
    px = settings["pixel per micrometre"]
    tid = t["TRACK_ID"].to_numpy()
    first_row = np.r_[True, tid[1:] != tid[:-1]]                       # different_tracks(): starts of tracks
    starts = np.flatnonzero(first_row)
    by_track = t.groupby("TRACK_ID")
    # ---- steps between consecutive rows of a track (track_eval.py:900-905)
    xd, yd, td = t["POSITION_X"].diff(), t["POSITION_Y"].diff(), t["POSITION_T"].diff()
    xd[first_row], yd[first_row], td[first_row] = 0, 0, 1
    t_norm = (t["POSITION_T"] - by_track["POSITION_T"].transform("first")).astype(np.int32)
    t["WIDTH"] = t["WIDTH"] / px
    t["HEIGHT"] = t["HEIGHT"] / px
    t["bac_length"] = np.where(t["WIDTH"] >= t["HEIGHT"], t["WIDTH"], t["HEIGHT"]).astype(np.float16)
    t["travelled_dist"] = np.sqrt(np.square(xd) + np.square(yd)) / px
    # ---- moving: speed above 1e-3, median filtered over 3 rows and over about a second (:927-939)
    moving = np.where(t["travelled_dist"] / td > 10 ** -3, 1, 0).astype(np.int8)
    second = int(round(fps, 0))
    for k in (3, second + 1 if second % 2 == 0 else second):
        moving = pd.Series(moving).groupby(tid).transform(medfilt, kernel_size=k).to_numpy()
    t["moving"] = moving
    # ---- change of heading between rows, heading taken over `compare angle between n frames` rows (:941-961)
    lag = settings["compare angle between n frames"]
    heading = np.degrees(np.arctan2(by_track["POSITION_X"].diff(lag), by_track["POSITION_Y"].diff(lag)))
    turn = abs(pd.Series(heading).groupby(tid).diff().fillna(0))
    t["angle_diff"] = np.where(360 - turn <= turn, 360 - turn, turn).astype(np.int32)
    candidates = np.where((t["angle_diff"] > settings["minimal angle in degrees for turning point"]) & (t["moving"] == 1),
                          t["angle_diff"], 0).astype(np.int32)
    # ---- turning points: the candidates that are the largest within 10 rows either side (:968-975;
    # argrelextrema_groupby's shift loop never runs: range(-1, -5) is empty)
    tp = np.zeros(n, np.int8)
    for a, b in zip(starts, np.r_[starts[1:], n]):
        seg = candidates[a:b]
        peaks = argrelextrema(seg, np.greater_equal, order=10)[0]
        tp[a:b][peaks] = np.where(seg[peaks] != 0, 1, 0)
    tp[first_row] = 1
    # ---- stretches between turning points, numbered through the whole table (:976-989).  A stretch starts where
    # the column turns from 0 to 1 (a track start right behind a turning point does not start a new one) and the
    # table's last row keeps number 0: it is the loop's stop index.
    run_start = tp.astype(bool) & np.r_[True, tp[:-1] == 0]
    number = np.cumsum(run_start) - 1
    number[-1] = 0
    t["tp_of_tracks"] = np.where(t["moving"] == 0, np.nan, number.astype(np.uint64))
    tp_dist = t.groupby("tp_of_tracks")["travelled_dist"].transform("sum")
    # ---- displacement within about ten seconds, longest stretch, both in body lengths (:990-1006)
    t["x_norm"] = (t["POSITION_X"] - by_track["POSITION_X"].transform("first")) / px
    t["y_norm"] = (t["POSITION_Y"] - by_track["POSITION_Y"].transform("first")) / px
    by_track = t.groupby("TRACK_ID")
    body = by_track["bac_length"].transform("mean")
    spans = [10] + [v / 2 for v in (settings["minimal length in seconds"], settings["limit track length to x seconds"])
                    if 0 < v / 2 < 10]
    lag_s = int(round(fps * min(spans), 0))
    reach = np.sqrt(np.square(by_track["x_norm"].diff(lag_s)) + np.square(by_track["y_norm"].diff(lag_s)))
    reach = reach.groupby(tid).transform("max") / body
    stretch = tp_dist.groupby(tid).transform("max") / body
    phenotype = np.where((reach > 1.5) & (stretch > 5), 2, np.where((reach > 1.5) & (stretch <= 5), 1, 0)).astype(np.int8)
    t["motility_phenotype"] = phenotype
    # ---- per track (:1029-1090)
    widest = by_track.apply(lambda g: pdist(np.column_stack([g["x_norm"], g["y_norm"]])).max(), include_groups=False)
    frames_last = pd.Series(t_norm.to_numpy()).groupby(tid).agg("last")
    per_second = t.groupby(["TRACK_ID", t.index // fps])["travelled_dist"].sum().groupby(level=0).median()
    moving_rows = by_track["moving"].agg("sum")
    percent_motile = moving_rows / (frames_last + 1) * 100
    seconds = (frames_last + 1) / fps
    path = by_track["travelled_dist"].agg("sum")
    chord = np.sqrt(np.square(by_track["x_norm"].agg("last")) + np.square(by_track["y_norm"].agg("last")))
    speed = np.where(moving_rows != 0, path / seconds, 0)
    arc_chord = np.where(path != 0, chord / path, 0)
    t["turn_points"] = np.where(phenotype != 0, tp, 0).astype(np.int8)
    t.loc[first_row, "turn_points"] = 1
    turns = (t.groupby("TRACK_ID")["turn_points"].agg("sum") - 1) * fps
    turns = np.where(moving_rows != 0, turns / moving_rows, 0)
    length = by_track["bac_length"].agg("mean")
    in_lengths = np.where(length != 0, widest / length, 0)
    idx = frames_last.index
    stats = pd.concat([pd.Series(turns, index=idx), path, pd.Series(speed, index=idx), seconds, widest, percent_motile,
                       pd.Series(arc_chord, index=idx), pd.Series(length), pd.Series(in_lengths, index=idx),
                       by_track["motility_phenotype"].agg("last"), by_track["TRACK_ID"].agg("last"),
                       pd.Series(per_second, index=idx)], keys=EVAL_STATS_COLUMNS, axis=1)
    return t.loc[:, EVAL_ROW_COLUMNS], stats

"""Device-side detection (a1-a6): thin host wrapper over ``ysmr_detect_batch``.

PyTorch is used only as the owner of device buffers and streams; all arithmetic happens in the
HIP kernels of ``csrc/detect.hip``.  Replaces, for a batch of frames at once, the per-frame
sequence cvtColor -> GaussianBlur -> 2 x adaptiveThreshold -> binary_propagation -> findContours
-> minAreaRect of the reference (ysmr/track_eval.py:180-303).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import _lib

__all__ = ["ThresholdParams", "threshold_params", "MeanGrayParams", "mean_gray_params", "MeanGrayState",
           "Detector", "DetectResult"]


@dataclass(frozen=True)
class ThresholdParams:
    inv: int
    t_low: int
    t_high: int
    use_high: int


def threshold_params(white_on_dark: bool, offset, adt) -> ThresholdParams:
    """Integer thresholds equivalent to the reference's two ``cv2.adaptiveThreshold`` calls.

    ysmr/track_eval.py:127-132 picks THRESH_BINARY / THRESH_BINARY_INV (negating the offset for
    dark-on-bright) and :189-208 passes ``C = -offset`` and ``C = -(offset + adt)``.  OpenCV turns
    C into ``idelta = ceil(C)`` (BINARY) or ``floor(C)`` (INV) and tests ``src - mean > -idelta``
    resp. ``<= -idelta``.  ``adt == 0`` means a single threshold; ``adt < 0`` selects the
    reference's mean-gray branch (``mean_gray_params``).
    """
    if adt < 0:
        raise ValueError("'adaptive double threshold' < 0 selects the mean-gray branch: use mean_gray_params()")
    inv = not white_on_dark
    off = offset * -1 if inv else offset
    c_low, c_high = off * -1, (off + adt) * -1
    if inv:
        return ThresholdParams(1, -math.floor(c_low), -math.floor(c_high), int(adt > 0))
    return ThresholdParams(0, -math.ceil(c_low), -math.ceil(c_high), int(adt > 0))


@dataclass(frozen=True)
class MeanGrayParams:
    """Arguments of ``ysmr_mean_threshold_batch`` (the branch at ysmr/track_eval.py:219-253)."""
    inv: int
    offset: float
    window: int


def mean_gray_params(white_on_dark: bool, offset, fps) -> MeanGrayParams:
    """``offset`` is the tracking.ini value; the reference negates it for dark-on-bright videos
    (track_eval.py:132) before it enters ``mean - stddev - offset`` (:226).  The moving average runs
    over at most floor(5 * fps) + 1 frames: the list is trimmed after the average, when it is longer
    than ``fps_of_file * 5`` (:239-242)."""
    if not fps > 0:
        raise ValueError("fps must be positive")
    inv = not white_on_dark
    return MeanGrayParams(int(inv), float(offset * -1 if inv else offset), int(math.floor(fps * 5)) + 1)


class MeanGrayState:
    """The reference's ``threshold_list`` between batches (device buffer; all zero = empty list).
    One per video; shared by the detectors that take turns on that video's batches."""

    def __init__(self, window, device="cuda:0"):
        n = _lib.lib().ysmr_mean_threshold_state_bytes(int(window))
        if n == 0:
            raise ValueError("window must be >= 1")
        self.window = int(window)
        self.buf = torch.zeros(n // 8, dtype=torch.float64, device=device)

    def reset(self):
        self.buf.zero_()


@dataclass
class DetectResult:
    """Device tensors of one ``Detector.detect`` call (views into the detector's buffers)."""
    cls: torch.Tensor        # u8  [B,H,W]  bit0 thresh, bit1 markers (bit2 internal)
    mask: torch.Tensor       # u8  [B,H,W]  final mask {0,255}
    labels: torch.Tensor     # i32 [B,H,W]  0 / 1 + raster index of the component's first pixel
    det_count: torch.Tensor  # i32 [B]
    det: torch.Tensor        # f32 [B,max_det,5]  cx, cy, w, h, angle
    anchors: torch.Tensor    # i32 [B,max_det]
    status: torch.Tensor     # i32 [B]  _lib.DET_* bits


def _on_own_device(method):
    """Run a Detector method with the detector's GPU as the current device."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with _lib.on(self.device):
            return method(self, *args, **kwargs)
    return wrapped


def _padded(n_bytes, device):
    return torch.empty((n_bytes + 15) // 16 * 16, dtype=torch.uint8, device=device)


class Detector:
    """Owns the output/scratch buffers for a fixed (batch, H, W, max_det) geometry."""

    def __init__(self, batch, height, width, max_det=2048, params: ThresholdParams | MeanGrayParams | None = None,
                 device="cuda:0", want_mask=True, mean_state: MeanGrayState | None = None, cv_flavour=0,
                 threshold_variant=0, beside_batch_link=False, beside_split_link=False):
        self.B, self.H, self.W, self.max_det = int(batch), int(height), int(width), int(max_det)
        #: which threshold kernel ``detect`` / ``threshold`` take (``ysmr_threshold_batch_variant``): 0 = the library's choice
        #: (the matrix-pipe kernel, which fills whole compute units); 1 = the float32-chain kernels, whose resident grid
        #: leaves registers and LDS on every compute unit for a link kernel running beside it (``TrackingPipeline``)
        self.threshold_variant = int(threshold_variant)
        # which OpenCV release a1 / a6 follow (_lib.CV_*), plus the hint that a link runs beside: the labelling / geometry
        # kernels then keep to the resident grids that leave its workgroups their LDS
        self.cv_flavour = _lib.cv_flavour_of(cv_flavour) | (_lib.BESIDE_LINK if self.threshold_variant == 1 else 0) | \
            (_lib.BESIDE_BATCH_LINK if beside_batch_link else 0) | (_lib.BESIDE_SPLIT_LINK if beside_split_link else 0)
        self.params = params or threshold_params(True, 5, 2.0)
        self.device = torch.device(device)
        L = _lib.lib()
        self.mean_state = None
        if isinstance(self.params, MeanGrayParams):
            self.mean_state = mean_state or MeanGrayState(self.params.window, self.device)
            if self.mean_state.window != self.params.window:
                raise ValueError("mean_state was made for another window")
            self.mean_stats = torch.zeros(self.B, 4, dtype=torch.float64, device=self.device)
            self.mean_levels = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        n = self.B * self.H * self.W
        ws = L.ysmr_detect_workspace_bytes(self.B, self.H, self.W, self.max_det)
        if ws == 0:
            raise ValueError("invalid detector geometry")
        self._ws = torch.empty(ws, dtype=torch.uint8, device=self.device)
        # the workspace remembers what the previous call wrote into labels/mask: start it blank
        with _lib.on(self.device):
            _lib.check(L.ysmr_detect_workspace_init(_lib.stream_ptr(self.device), self._ws.data_ptr(), ws),
                       "ysmr_detect_workspace_init")
        self._cls = _padded(n, self.device)
        self._mask = _padded(n, self.device) if want_mask else None
        self._labels = torch.empty((n + 3) // 4 * 4, dtype=torch.int32, device=self.device)
        self.det_count = torch.zeros(self.B, dtype=torch.int32, device=self.device)
        self.det = torch.zeros(self.B, self.max_det, 5, dtype=torch.float32, device=self.device)
        self.anchors = torch.zeros(self.B, self.max_det, dtype=torch.int32, device=self.device)
        self.status = torch.zeros(self.B, dtype=torch.int32, device=self.device)

    def _view(self, buf, batch):
        n = batch * self.H * self.W
        return buf[:n].view(batch, self.H, self.W)

    def _check_frames(self, frames):
        if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous():
            raise ValueError("frames must be a contiguous uint8 device tensor")
        if frames.dim() not in (3, 4) or frames.shape[1] != self.H or frames.shape[2] != self.W:
            raise ValueError(f"frames must be [b,{self.H},{self.W}] or [b,{self.H},{self.W},3]")
        b = frames.shape[0]
        if not 1 <= b <= self.B:
            raise ValueError(f"batch {b} exceeds detector batch {self.B}")
        ch = 1 if frames.dim() == 3 else frames.shape[3]
        return b, ch

    @_on_own_device
    def threshold(self, frames: torch.Tensor, variant: int | None = None, timing=None) -> torch.Tensor:
        """a1-a3 only: class map u8 [b,H,W] (bit0 thresh, bit1 markers).  In the mean-gray branch the
        call also advances the moving-average state by these frames; per-frame mean, stddev, level and
        averaged level are left in ``mean_stats[:b]``, the integer levels in ``mean_levels[:b]``.
        ``variant``: which kernel (``ysmr_threshold_batch_variant`` in include/ysmr_hip.h; 0 = the shipped choice).
        ``timing``: a pair of ``torch.cuda.Event(enable_timing=True)`` that have been recorded before (so that they
        exist); the kernel's dispatch sets them to its own start and end (``ysmr_threshold_timing``)."""
        b, ch = self._check_frames(frames)
        p = self.params
        variant = self.threshold_variant if variant is None else variant
        if timing is not None and self.mean_state is None:
            _lib.check(_lib.lib().ysmr_threshold_timing(timing[0].cuda_event, timing[1].cuda_event), "ysmr_threshold_timing")
        if self.mean_state is not None:
            rc = _lib.lib().ysmr_mean_threshold_batch(
                _lib.stream_ptr(self.device), frames.data_ptr(), b, self.H, self.W, ch, p.inv, p.offset, p.window,
                self.mean_state.buf.data_ptr(), self.mean_stats.data_ptr(), self.mean_levels.data_ptr(),
                self._cls.data_ptr(), self.cv_flavour)
            _lib.check(rc, "ysmr_mean_threshold_batch")
            return self._view(self._cls, b)
        if variant:
            rc = _lib.lib().ysmr_threshold_batch_variant(_lib.stream_ptr(self.device), frames.data_ptr(), b, self.H, self.W,
                                                         ch, p.inv, p.t_low, p.t_high, p.use_high, self._cls.data_ptr(),
                                                         self.cv_flavour, int(variant))
            _lib.check(rc, "ysmr_threshold_batch_variant")
            return self._view(self._cls, b)
        rc = _lib.lib().ysmr_threshold_batch(_lib.stream_ptr(self.device), frames.data_ptr(), b, self.H, self.W, ch,
                                             p.inv, p.t_low, p.t_high, p.use_high, self._cls.data_ptr(), self.cv_flavour)
        _lib.check(rc, "ysmr_threshold_batch")
        return self._view(self._cls, b)

    @_on_own_device
    def components(self, batch=None, cls: torch.Tensor | None = None) -> DetectResult:
        """a4-a6 on the class map left by ``threshold`` (or on a caller-supplied u8 [b,H,W] map)."""
        if cls is not None:
            batch = cls.shape[0]
            self._view(self._cls, batch).copy_(cls)
        b = self.B if batch is None else int(batch)
        rc = _lib.lib().ysmr_components_batch(
            _lib.stream_ptr(self.device), b, self.H, self.W, self._ws.data_ptr(), self._ws.numel(), self._cls.data_ptr(),
            self._mask.data_ptr() if self._mask is not None else None, self._labels.data_ptr(),
            self.det_count.data_ptr(), self.det.data_ptr(), self.anchors.data_ptr(), self.max_det,
            self.status.data_ptr(), self.cv_flavour)
        _lib.check(rc, "ysmr_components_batch")
        return self._result(b)

    def _result(self, b):
        return DetectResult(self._view(self._cls, b),
                            self._view(self._mask, b) if self._mask is not None else None,
                            self._view(self._labels, b), self.det_count[:b], self.det[:b], self.anchors[:b],
                            self.status[:b])

    @_on_own_device
    def detect(self, frames: torch.Tensor) -> DetectResult:
        """a1-a6 for a batch of frames resident in HBM.  Asynchronous on the current stream."""
        b, ch = self._check_frames(frames)
        p = self.params
        if self.mean_state is not None or self.threshold_variant:
            self.threshold(frames)
            return self.components(b)
        rc = _lib.lib().ysmr_detect_batch(
            _lib.stream_ptr(self.device), frames.data_ptr(), b, self.H, self.W, ch, p.inv, p.t_low, p.t_high, p.use_high,
            self._ws.data_ptr(), self._ws.numel(), self._cls.data_ptr(),
            self._mask.data_ptr() if self._mask is not None else None, self._labels.data_ptr(),
            self.det_count.data_ptr(), self.det.data_ptr(), self.anchors.data_ptr(), self.max_det,
            self.status.data_ptr(), self.cv_flavour)
        _lib.check(rc, "ysmr_detect_batch")
        return self._result(b)

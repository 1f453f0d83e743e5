"""Nearest-neighbour tracker on the device.

``DeviceTracker``  -- handle-level wrapper over ``ysmr_tracker_*`` (state lives in HBM across
                      frames; used by the batched frame loop in track_eval.py).
``CentroidTracker`` -- the reference's class (ysmr/tracker.py:27-230) with the same constructor,
                      ``update(rects)`` contract and public attributes.  The linking decision
                      and the GSFF run in the HIP kernels; the host only mirrors the dictionaries
                      (so that ``additional_info`` may carry arbitrary Python objects, as upstream).
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

__all__ = ["sort_rows", "DeviceTracker", "CentroidTracker"]


def _on_own_device(method):
    """Run a DeviceTracker method with the tracker's GPU as the current device."""
    import functools

    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with _lib.on(self.device):
            return method(self, *args, **kwargs)
    return wrapped


class DeviceTracker:
    """Owns one ``ysmr_tracker`` handle (one video stream)."""

    def __init__(self, max_disappeared=50, fps=30, n_min=0, n_max=None, n_f=3, use_gsff=True,
                 capacity=1024, max_det=2048, device="cuda:0", gains=None):
        self.device = torch.device(device)
        self.capacity, self.max_det = int(capacity), int(max_det)
        self.use_gsff = bool(use_gsff)
        self._handle = ctypes.c_void_p()
        # gains: None = the library's closed form of the least-squares gain (gsff.py:111-153); an array laid
        # out as ysmr_gsff_gains() writes it overrides it (experiments with other gains)
        g = None if gains is None else np.ascontiguousarray(gains, np.float64).ravel()
        with torch.cuda.device(self.device):
            rc = _lib.lib().ysmr_tracker_create(
                float(max_disappeared), float(fps), int(n_min), float(-1 if n_max is None else n_max), int(n_f),
                int(self.use_gsff), self.capacity, self.max_det, None if g is None else g.ctypes.data,
                ctypes.byref(self._handle))
        _lib.check(rc, "ysmr_tracker_create")

    def close(self):
        if self._handle:
            _lib.lib().ysmr_tracker_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @_on_own_device
    def reset(self):
        _lib.check(_lib.lib().ysmr_tracker_reset(self._handle, _lib.stream_ptr(self.device)), "ysmr_tracker_reset")

    @_on_own_device
    def update(self, det, m=None, m_dev=None, frame=0, rows=None, n_rows=None, claim=None, n_before=None,
               new_cols=None, n_new=None):
        """One frame.  det: device tensor [m,5] float32 or float64."""
        f64 = int(det.dtype == torch.float64)
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        rc = _lib.lib().ysmr_tracker_update(self._handle, _lib.stream_ptr(self.device), det.data_ptr(), f64,
                                            -1 if m is None else int(m), ptr(m_dev), int(frame), ptr(rows),
                                            ptr(n_rows), ptr(claim), ptr(n_before), ptr(new_cols), ptr(n_new))
        _lib.check(rc, "ysmr_tracker_update")

    @_on_own_device
    def run(self, det, det_count, first_frame, rows, row_count):
        """Frames [first_frame, first_frame + B): det f32 [B,max_det,5], det_count i32 [B] on device;
        rows: uint8 buffer viewed as ysmr_row[]; row_count: int64 device scalar (advanced)."""
        b = det_count.numel()
        if det.shape[1] != self.max_det:
            raise ValueError("det must be [B, max_det, 5] with the tracker's max_det")
        rc = _lib.lib().ysmr_tracker_run(self._handle, _lib.stream_ptr(self.device), det.data_ptr(),
                                         det_count.data_ptr(), b, int(first_frame), rows.data_ptr(),
                                         rows.numel() // _lib.ROW_DTYPE.itemsize, row_count.data_ptr())
        _lib.check(rc, "ysmr_tracker_run")

    @_on_own_device
    def prepare(self, det, det_count, slot):
        """Bin a batch's detections for the ``run`` that follows, on the CURRENT stream (``ysmr_tracker_prepare``: takes
        that launch off the link stream's chain when called on the detection stream).  No-op unless ``batched``."""
        rc = _lib.lib().ysmr_tracker_prepare(self._handle, _lib.stream_ptr(self.device), det.data_ptr(), det_count.data_ptr(),
                                             det_count.numel(), int(slot))
        _lib.check(rc, "ysmr_tracker_prepare")

    @property
    def fused(self):
        """True when the handle links with one launch per frame (``k_frame``)."""
        return bool(_lib.lib().ysmr_tracker_fused(self._handle))

    @property
    def batched(self):
        """True when ``run`` links a whole batch with one launch (``k_batch``: a track per lane of one workgroup)."""
        return bool(_lib.lib().ysmr_tracker_batched(self._handle))

    def link_mode(self, mode):
        """0: the library's choice; 1: one launch per frame even where a batch launch would serve (measurement, tests).
        The track table is carried over."""
        _lib.check(_lib.lib().ysmr_tracker_link_mode(self._handle, int(mode)), "ysmr_tracker_link_mode")

    @_on_own_device
    def info(self):
        """(live tracks, next id, sticky error bits); synchronises."""
        a, b, c = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        rc = _lib.lib().ysmr_tracker_info(self._handle, _lib.stream_ptr(self.device), ctypes.byref(a), ctypes.byref(b),
                                          ctypes.byref(c))
        _lib.check(rc, "ysmr_tracker_info")
        return a.value, b.value, c.value

    @_on_own_device
    def peek(self):
        """Current (ids, positions (n,2) float64, disappeared) in id order; synchronises."""
        ids = torch.empty(self.capacity, dtype=torch.int32, device=self.device)
        xy = torch.empty(self.capacity, 2, dtype=torch.float64, device=self.device)
        gone = torch.empty(self.capacity, dtype=torch.int32, device=self.device)
        n = torch.zeros(1, dtype=torch.int32, device=self.device)
        rc = _lib.lib().ysmr_tracker_peek(self._handle, _lib.stream_ptr(self.device), ids.data_ptr(), xy.data_ptr(),
                                          gone.data_ptr(), n.data_ptr())
        _lib.check(rc, "ysmr_tracker_peek")
        k = int(n.item())
        return ids[:k].cpu().numpy(), xy[:k].cpu().numpy(), gone[:k].cpu().numpy()


class _SingleFilter:
    """One GaussianSumFIR state on the device (a one-track tracker fed one detection per step)."""

    def __init__(self, gsff):
        gains = np.concatenate([g.ravel() for g in gsff.gains])
        fps = 1.0 / gsff._delta_t
        self._trk = DeviceTracker(max_disappeared=30000.0, fps=fps, n_min=gsff._n_min, n_max=gsff._n_max,
                                  n_f=gsff.n_f, use_gsff=True, capacity=1, max_det=1, device=gsff._device,
                                  gains=gains)
        d = self._trk.device
        self._rows = torch.empty(_lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device=d)
        self._n = torch.zeros(1, dtype=torch.int32, device=d)
        self._pred = None
        self.mode = 0
        self._steps = 0
        self._n_i = list(gsff.n_i)

    def correct(self, z):
        det = torch.tensor([[z[0], z[1], 0.0, 0.0, 0.0]], dtype=torch.float64, device=self._trk.device)
        self._trk.update(det, m=1, rows=self._rows, n_rows=self._n)
        row = rows_to_numpy(self._rows, 1)[0]
        self._pred = self._trk.peek()[1][0]
        # mode as in gsff.py:283-289: history holds n_i[0] copies plus one entry per earlier step
        length = self._n_i[0] + self._steps
        self.mode = sum(1 for n in self._n_i if length >= n)
        self._steps += 1
        return np.array([row["x"], row["y"]])

    def predict(self):
        return None if self._pred is None else self._pred.copy()


def rows_to_numpy(rows_u8: torch.Tensor, count: int) -> np.ndarray:
    """Download `count` rows from a device uint8 buffer laid out as ysmr_row[]."""
    size = _lib.ROW_DTYPE.itemsize
    if count * size < (1 << 20):
        return rows_u8[: count * size].cpu().numpy().view(_lib.ROW_DTYPE)
    # a table: through pinned memory (one DMA at the link's rate; a pageable destination is staged piece by piece --
    # 10-19 ms for the 39 MB of a 1920-frame video against 1-2)
    host = torch.empty(count * size, dtype=torch.uint8, pin_memory=True)
    host.copy_(rows_u8[: count * size])
    return host.numpy().view(_lib.ROW_DTYPE)


def sort_rows(rows_u8: torch.Tensor, count: int) -> torch.Tensor:
    """Order `count` device rows by (TRACK_ID, POSITION_T) on the device -- what ``sort_list`` does to
    the csv after tracking (helper_file.py:1538-1574).  Returns a new device uint8 buffer."""
    size = _lib.ROW_DTYPE.itemsize
    out = torch.empty(max(count, 1) * size, dtype=torch.uint8, device=rows_u8.device)
    if count == 0:
        return out[:0]
    L = _lib.lib()
    ws_bytes = L.ysmr_rows_sort_workspace_bytes(count)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=rows_u8.device)
    with torch.cuda.device(rows_u8.device):
        _lib.check(L.ysmr_rows_sort(_lib.stream_ptr(rows_u8.device), rows_u8.data_ptr(), count, ws.data_ptr(), ws_bytes, out.data_ptr()),
                   "ysmr_rows_sort")
    return out[: count * size]


class CentroidTracker:
    """Drop-in for ``ysmr.tracker.CentroidTracker`` (tracker.py:27-230).

    ``update(rects)`` takes ``[((x, y), additional_info), ...]`` and returns
    ``(OrderedDict id -> np.array([x, y]), OrderedDict id -> additional_info)``: the filtered
    centroids when the GSFF is on (tracker.py:219-227), the raw ones otherwise (:228-230).
    Only 2-D centroids are supported (the luminosity dimension is off by default upstream and
    out of scope here).
    """

    def __init__(self, max_disappeared=50, fps=30, n_min=0, n_max=None, n_f=3, use_gsff=True,
                 capacity=4096, max_det=4096, device="cuda:0"):
        self.maxDisappeared = max_disappeared
        self.use_gsff = use_gsff
        self.additional_info = OrderedDict()
        self._ids = []
        self._dev = DeviceTracker(max_disappeared, fps, n_min, n_max, n_f, use_gsff, capacity, max_det, device)
        d = self._dev.device
        self._rows = torch.empty(capacity * _lib.ROW_DTYPE.itemsize, dtype=torch.uint8, device=d)
        self._claim = torch.empty(capacity, dtype=torch.int32, device=d)
        self._new = torch.empty(max_det, dtype=torch.int32, device=d)
        self._scal = torch.zeros(4, dtype=torch.int32, device=d)  # n_rows, n_before, n_new
        self._last_rows = np.zeros(0, _lib.ROW_DTYPE)

    # -- read-only mirrors of the reference's public attributes --------------------------------
    @property
    def nextObjectID(self):
        return self._dev.info()[1]

    @property
    def objects(self):
        ids, xy, _ = self._dev.peek()
        return OrderedDict((int(i), xy[k].copy()) for k, i in enumerate(ids))

    @property
    def disappeared(self):
        ids, _, gone = self._dev.peek()
        return OrderedDict((int(i), int(g)) for i, g in zip(ids, gone))

    def update(self, rects):
        m = len(rects)
        if m:
            if len(rects[0][0]) != 2:
                raise NotImplementedError("only (x, y) centroids are supported (luminosity tracking is out of scope)")
            if m > self._dev.max_det:
                raise ValueError(f"{m} detections exceed max_det={self._dev.max_det}")
            host = np.zeros((m, 5), np.float64)
            for i, (xy, info) in enumerate(rects):
                host[i, 0], host[i, 1] = xy
                try:
                    host[i, 2:5] = info
                except (TypeError, ValueError):
                    pass  # non-numeric payload: kept on the host only
            det = torch.from_numpy(host).to(self._dev.device)
        else:
            det = torch.zeros((1, 5), dtype=torch.float64, device=self._dev.device)
        s = self._scal
        self._dev.update(det, m=m, rows=self._rows, n_rows=s[0:1], claim=self._claim, n_before=s[1:2],
                         new_cols=self._new, n_new=s[2:3])
        n_rows, n_before, n_new = (int(v) for v in s[:3].cpu().numpy())
        rows = rows_to_numpy(self._rows, n_rows)
        claim = self._claim[:n_before].cpu().numpy()
        new_cols = self._new[:n_new].cpu().numpy()
        _, _, err = self._dev.info()
        if err:
            raise _lib.YsmrLibraryError(f"tracker capacity exceeded (error bits {err})")

        # host mirror of additional_info (tracker.py:183, 205, 217)
        aged = m == 0 or (n_before > 0 and n_before >= m)
        for r, tid in enumerate(self._ids):
            c = int(claim[r]) if r < len(claim) else -1
            if c >= 0:
                self.additional_info[tid] = rects[c][1]
            elif aged:
                self.additional_info[tid] = [0] * len(self.additional_info[tid])
        alive = [int(t) for t in rows["track_id"]]
        alive_set = set(alive)
        for tid in self._ids:
            if tid not in alive_set:
                del self.additional_info[tid]
        for tid, c in zip(alive[len(alive) - n_new:] if n_new else [], new_cols):
            self.additional_info[tid] = rects[int(c)][1]
        self._ids = alive
        self._last_rows = rows
        out = OrderedDict((int(r["track_id"]), np.array([r["x"], r["y"]])) for r in rows)
        return out, self.additional_info

    @property
    def last_claims(self):
        """(row, column) pairs accepted by the most recent update (diagnostics / parity tests)."""
        n_before = int(self._scal[1].item())
        claim = self._claim[:n_before].cpu().numpy()
        return [(r, int(c)) for r, c in enumerate(claim) if c >= 0]

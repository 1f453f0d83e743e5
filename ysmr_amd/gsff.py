"""Gaussian-sum FIR filter (Pak 2019) -- the reference-compatible class over the device filter.

Horizons and least-squares gains come from the library (``ysmr_gsff_gains``, ``csrc/track.hip``): for the
constant-velocity model the reference always uses (gsff.py:111-126), rows 0/1 of ``(L^T L)^-1 L^T`` are the
closed form c_N[j] = 1/N + t_j ((N + 1) / 2) / sum t^2, t_j = j - (N - 1)/2, exactly decoupled in x and y;
it agrees with the reference's NumPy/LAPACK evaluation to 4e-16 (tests/test_cabi.py compares it with the
oracle's restatement of gsff.py:87-153).  The per-measurement arithmetic (``correct`` / ``predict``,
gsff.py:204-347) runs in ``csrc/track.hip`` (``gsff_step``): inside the tracker for the hot path, and
through a one-track device tracker for this stand-alone class.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

__all__ = ["GaussianSumFIR", "horizons_and_gains"]


def horizons_and_gains(fps, n_min=0, n_max=None, n_f=3):
    """-> (n_i, gains): the filter horizons N_i (gsff.py:87-109; ``n_max=None`` means fps, tracker.py:55-56)
    and, per filter, the two position rows of its gain as a (2, 2 N_i) array -- from ``ysmr_gsff_gains``."""
    L = _lib.lib()
    n_i = (ctypes.c_int32 * int(n_f))()
    top = float(-1 if n_max is None else n_max)
    _lib.check(L.ysmr_gsff_gains(float(fps), int(n_min), top, int(n_f), n_i, None), "ysmr_gsff_gains")
    sizes = list(n_i)
    flat = np.zeros(4 * sum(sizes), np.float64)
    _lib.check(L.ysmr_gsff_gains(float(fps), int(n_min), top, int(n_f), n_i, flat.ctypes.data), "ysmr_gsff_gains")
    gains, off = [], 0
    for n in sizes:
        gains.append(flat[off:off + 4 * n].reshape(2, 2 * n).copy())
        off += 4 * n
    return sizes, gains


class GaussianSumFIR:
    """Reference-compatible filter object (ysmr/gsff.py:28-347).

    ``correct(measurement, **state)`` / ``predict(**state)`` thread a state dict exactly like
    upstream; here the dict carries an opaque device-side filter (key ``'_device'``) next to
    read-only mirrors of ``mode``.  Only 2-D measurements with ``x_hat_array_length == 2`` and the
    identity ``inv_cov`` (what CentroidTracker uses, tracker.py:60-70) are supported.
    """

    def __init__(self, delta_t, n_min=0, n_max=30, n_f=3, a=None, c=None, likelihood_minimum=10 ** -20,
                 inv_cov=None, x_hat_array_length=2, device="cuda:0"):
        if x_hat_array_length != 2:
            raise NotImplementedError("x_hat_array_length must be 2")
        if inv_cov is not None and not np.array_equal(np.asarray(inv_cov), np.eye(2)):
            raise NotImplementedError("only the identity inverse covariance is supported")
        if likelihood_minimum != 10 ** -20:
            raise NotImplementedError("likelihood_minimum is fixed at 1e-20 (tracker.py:67)")
        if a is not None or c is not None:
            raise NotImplementedError("only the constant-velocity model (a = c = None, gsff.py:111-126) is supported")
        self.likelihood_minimum = likelihood_minimum
        self.x_hat_array_length = x_hat_array_length
        self.n_f = n_f
        #: horizons, and per filter the two position rows (2, 2 N) of its gain (upstream keeps all four rows;
        #: rows 2/3, the velocity estimates, are never read: gsff.py:240)
        self.n_i, self.gains = horizons_and_gains(1.0 / delta_t, n_min, n_max, n_f)
        self.inv_cov = np.eye(2) if inv_cov is None else inv_cov
        self._delta_t, self._n_min, self._n_max = delta_t, n_min, n_max
        self._device = device

    def _new_filter(self):
        from .tracker import _SingleFilter
        return _SingleFilter(self)

    def correct(self, measurement, **kwargs):
        filt = kwargs.get("_device")
        if filt is None:
            filt = self._new_filter()
        x_hat = filt.correct(np.asarray(measurement, dtype=np.float64))
        kwargs.update({"_device": filt, "mode": filt.mode})
        return x_hat, kwargs

    def predict(self, **kwargs):
        filt = kwargs.get("_device")
        if filt is None:
            return None, kwargs
        return filt.predict(), kwargs

"""Gaussian-sum FIR filter (Pak 2019) -- host-side set-up and the reference-compatible class.

Set-up (``horizon_sizes``, ``lsf_gain``) runs once per tracker on the host and follows
``GaussianSumFIR.generate_n_i`` / ``compute_lsf_gain`` (ysmr/gsff.py:87-153) so that the device
kernels are handed the very coefficients the reference would use.  The per-measurement arithmetic
(``correct`` / ``predict``, gsff.py:204-347) runs in ``csrc/track.hip`` (``gsff_step``): inside the
tracker for the hot path, and through a one-track device tracker for this stand-alone class.
"""
from __future__ import annotations

import numpy as np

__all__ = ["GaussianSumFIR", "horizon_sizes", "lsf_gain", "lsf_gain_rows"]


def horizon_sizes(n_min=0, n_max=30, n_f=3):
    """Filter horizons N_i = int(n_min + i * (n_max - n_min) / n_f), i = 1..n_f (Pak eq. 17;
    gsff.py:87-109).  With the defaults: [10, 20, 30]; with n_max = 29.97 fps: [9, 19, 29]."""
    step = (n_max - n_min) / n_f
    return [int(n_min + step * k) for k in range(1, n_f + 1)]


def lsf_gain(size, delta_t, a=None, c=None):
    """Least-squares FIR gain (L^T L)^-1 L^T, L = [C; CA; ..; CA^(N-1)] A^-N (Pak eqs. 13-14;
    gsff.py:111-153).  Shape (4, 2N) for the default constant-velocity model."""
    if a is None:
        a = np.array([[1, 0, delta_t, 0],
                      [0, 1, 0, delta_t],
                      [0, 0, 1, 0],
                      [0, 0, 0, 1]], dtype=np.float64)
    if c is None:
        c = np.array([[1, 0, 0, 0],
                      [0, 1, 0, 0]])
    stacked, a_pow = c, a
    for _ in range(size - 1):
        stacked = np.concatenate((stacked, np.dot(c, a_pow)), axis=0)
        a_pow = np.dot(a_pow, a)
    ell = np.dot(stacked, np.linalg.matrix_power(np.linalg.inv(a), size))
    return np.dot(np.linalg.inv(np.dot(ell.T, ell)), ell.T)


def lsf_gain_rows(size, delta_t, a=None, c=None):
    """Rows 0 and 1 (the position estimates) of ``lsf_gain`` -- what the device kernels consume."""
    return np.ascontiguousarray(lsf_gain(size, delta_t, a, c)[:2])


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
        self.likelihood_minimum = likelihood_minimum
        self.x_hat_array_length = x_hat_array_length
        self.n_f = n_f
        self.n_i = horizon_sizes(n_min=n_min, n_max=n_max, n_f=n_f)
        self.gains = [lsf_gain(n, delta_t, a, c) for n in self.n_i]
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

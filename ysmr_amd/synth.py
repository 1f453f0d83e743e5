"""Synthetic microscope video for benchmarks and parity tests.

Bright rod-shaped "bacteria" on a dark, noisy background, moving run-and-tumble, with per-frame
dropout and single-pixel speckle so that the tracker's disappear/register paths are exercised.
Deterministic for a given seed (numpy.random.default_rng).  There is no network in the build or
benchmark environment, so this generator stands in for real recordings; frame geometry and blob
statistics follow the reference's defaults (ysmr/helper_file.py:160-175: 1228x922 @ 30 fps,
area limits 2..50 px^2, rod width/height ratio 0.125..0.67).
"""
from __future__ import annotations

import numpy as np

__all__ = ["SyntheticVideo", "S50", "S500", "S4K"]

_PATCH = 13  # rods are at most ~8 px long: a 13x13 patch around the rounded centre covers them


class SyntheticVideo:
    """Iterable/indexable source of (H, W) uint8 frames.

    Frames must be generated in order (blob state is advanced frame by frame); ``frames(n)``
    returns an (n, H, W) uint8 array and may be called repeatedly to continue the stream.
    """

    def __init__(self, height=922, width=1228, n_blobs=500, seed=0, fps=30.0, background=40.0,
                 noise_sigma=2.0, dropout=0.02, speckle=0.01, tumble=0.03):
        self.h, self.w, self.n = int(height), int(width), int(n_blobs)
        self.fps = float(fps)
        self.bg, self.sigma = float(background), float(noise_sigma)
        self.dropout, self.speckle, self.tumble = dropout, speckle, tumble
        rng = self.rng = np.random.default_rng(seed)
        n = self.n
        self.pos = np.stack([rng.uniform(8, self.w - 8, n), rng.uniform(8, self.h - 8, n)], axis=1)
        self.heading = rng.uniform(0, 2 * np.pi, n)
        self.speed = rng.uniform(0.5, 3.0, n)
        self.length = 6.0 + rng.uniform(-1.0, 1.0, n)
        self.width_px = 2.0 + rng.uniform(-0.3, 0.3, n)
        self.peak = rng.uniform(120.0, 220.0, n)
        self.frame_index = 0
        gy, gx = np.mgrid[0:_PATCH, 0:_PATCH]
        self._gx = (gx - _PATCH // 2).astype(np.float32)[None]
        self._gy = (gy - _PATCH // 2).astype(np.float32)[None]

    # -- motion -------------------------------------------------------------------------------
    def _advance(self):
        rng = self.rng
        n = self.n
        tumbling = rng.random(n) < self.tumble
        self.heading = np.where(tumbling, rng.uniform(0, 2 * np.pi, n), self.heading)
        self.pos[:, 0] += self.speed * np.cos(self.heading)
        self.pos[:, 1] += self.speed * np.sin(self.heading)
        for axis, hi in ((0, self.w - 4.0), (1, self.h - 4.0)):
            low = self.pos[:, axis] < 4.0
            high = self.pos[:, axis] > hi
            self.pos[low, axis] = 8.0 - self.pos[low, axis]
            self.pos[high, axis] = 2 * hi - self.pos[high, axis]
            flip = low | high
            if axis == 0:
                self.heading[flip] = np.pi - self.heading[flip]
            else:
                self.heading[flip] = -self.heading[flip]

    # -- rendering ----------------------------------------------------------------------------
    def _render(self):
        rng = self.rng
        h, w, n = self.h, self.w, self.n
        img = rng.standard_normal(size=(h, w), dtype=np.float32)
        img *= np.float32(self.sigma)
        img += np.float32(self.bg)
        if n:
            visible = rng.random(n) >= self.dropout
            idx = np.nonzero(visible)[0]
            c = self.pos[idx]
            ci = np.rint(c).astype(np.int64)
            frac = (c - ci).astype(np.float32)
            # capsule signed distance: segment half-length (L - W)/2 along the heading, radius W/2
            ux = np.cos(self.heading[idx]).astype(np.float32)[:, None, None]
            uy = np.sin(self.heading[idx]).astype(np.float32)[:, None, None]
            px = self._gx - frac[:, 0, None, None]
            py = self._gy - frac[:, 1, None, None]
            half = ((self.length[idx] - self.width_px[idx]) * 0.5).astype(np.float32)[:, None, None]
            t = np.clip(px * ux + py * uy, -half, half)
            dist = np.sqrt((px - t * ux) ** 2 + (py - t * uy) ** 2)
            cover = np.clip(0.5 - (dist - self.width_px[idx].astype(np.float32)[:, None, None] * 0.5),
                            0.0, 1.0)
            val = cover * (self.peak[idx].astype(np.float32)[:, None, None] - self.bg)
            yy = ci[:, 1, None, None] + self._gy.astype(np.int64)
            xx = ci[:, 0, None, None] + self._gx.astype(np.int64)
            ok = (cover > 0) & (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            signal = np.zeros((h, w), np.float32)
            np.maximum.at(signal, (yy[ok], xx[ok]), val[ok])
            img += signal
        k = int(round(self.speckle * n))
        if k:
            sy = rng.integers(0, h, k)
            sx = rng.integers(0, w, k)
            img[sy, sx] = rng.uniform(120.0, 220.0, k).astype(np.float32)
        np.rint(img, out=img)
        np.clip(img, 0, 255, out=img)
        return img.astype(np.uint8)

    def next_frame(self):
        if self.frame_index:
            self._advance()
        self.frame_index += 1
        return self._render()

    def frames(self, count):
        out = np.empty((count, self.h, self.w), np.uint8)
        for i in range(count):
            out[i] = self.next_frame()
        return out

    def __iter__(self):
        while True:
            yield self.next_frame()


def S50(seed=0):
    """Config 0 geometry: 1228x922, ~50 blobs."""
    return SyntheticVideo(922, 1228, 50, seed)


def S500(seed=0):
    """Configs 1-3: 1228x922, ~500 blobs."""
    return SyntheticVideo(922, 1228, 500, seed)


def S4K(seed=0):
    """Config 4: 3840x2160 dense field, ~5000 blobs."""
    return SyntheticVideo(2160, 3840, 5000, seed)

"""Frame sources for the detect-and-link path.

The reference reads frames with ``cv2.VideoCapture`` (ysmr/track_eval.py:65-93, 159).  OpenCV is
not a dependency of this package, so raw containers are read natively and ``cv2`` is used only when
it happens to be importable:

* ``.npy``  -- uint8 array [T, H, W] (gray) or [T, H, W, 3] (BGR), memory-mapped; fps from a
               ``<name>_meta.json`` side file (key ``fps``) or the tracking.ini value
* ``.y4m``  -- YUV4MPEG2; the luma plane is used as the gray frame
* anything else -- ``cv2.VideoCapture`` if cv2 can be imported, otherwise an error

Every source exposes ``frame_count``, ``fps``, ``height``, ``width``, ``channels`` and
``read(start, count) -> uint8 ndarray [n, H, W(, 3)]`` (fewer than ``count`` frames at the end).
"""
from __future__ import annotations

import json
import os

import numpy as np

__all__ = ["open_video", "NpyVideo", "Y4mVideo", "Cv2Video"]


class NpyVideo:
    def __init__(self, path, default_fps=30.0):
        self.path = path
        self._a = np.load(path, mmap_mode="r")
        if self._a.dtype != np.uint8 or self._a.ndim not in (3, 4):
            raise ValueError(f"{path}: expected uint8 [T,H,W] or [T,H,W,3], got {self._a.dtype} {self._a.shape}")
        if self._a.ndim == 4 and self._a.shape[3] != 3:
            raise ValueError(f"{path}: colour frames must have 3 channels")
        self.frame_count, self.height, self.width = (int(v) for v in self._a.shape[:3])
        self.channels = 1 if self._a.ndim == 3 else 3
        self.fps = float(default_fps)
        meta = os.path.splitext(path)[0] + "_meta.json"
        try:
            with open(meta) as fh:
                self.fps = float(json.load(fh).get("fps", self.fps))
        except (OSError, ValueError, TypeError):
            pass

    def read(self, start, count):
        return np.ascontiguousarray(self._a[start:start + count])

    def close(self):
        self._a = None


class Y4mVideo:
    """Minimal YUV4MPEG2 reader (8-bit; C420*, C422, C444 or Cmono): luma only."""

    def __init__(self, path, default_fps=30.0):
        self.path = path
        self._fh = open(path, "rb")
        header = self._fh.readline()
        if not header.startswith(b"YUV4MPEG2"):
            raise ValueError(f"{path}: not a YUV4MPEG2 file")
        self.fps, chroma = float(default_fps), "420"
        for tok in header.split()[1:]:
            t = tok.decode("ascii", "replace")
            if t[0] == "W":
                self.width = int(t[1:])
            elif t[0] == "H":
                self.height = int(t[1:])
            elif t[0] == "F" and ":" in t:
                num, den = t[1:].split(":")
                if int(den):
                    self.fps = int(num) / int(den)
            elif t[0] == "C":
                chroma = t[1:]
        y = self.width * self.height
        if chroma.startswith("420"):
            cw, ch = (self.width + 1) // 2, (self.height + 1) // 2
            self._frame_bytes = y + 2 * cw * ch
        elif chroma.startswith("422"):
            self._frame_bytes = y + 2 * ((self.width + 1) // 2) * self.height
        elif chroma.startswith("444"):
            self._frame_bytes = 3 * y
        elif chroma.startswith("mono"):
            self._frame_bytes = y
        else:
            raise ValueError(f"{path}: unsupported chroma '{chroma}'")
        self._data0 = self._fh.tell()
        size = os.path.getsize(path) - self._data0
        self._stride = len(b"FRAME\n") + self._frame_bytes   # frames without per-frame parameters
        self.frame_count = size // self._stride
        self.channels = 1

    def read(self, start, count):
        n = max(0, min(count, self.frame_count - start))
        out = np.empty((n, self.height, self.width), np.uint8)
        for i in range(n):
            self._fh.seek(self._data0 + (start + i) * self._stride)
            marker = self._fh.readline()
            if not marker.startswith(b"FRAME"):
                raise ValueError(f"{self.path}: frame {start + i} has no FRAME marker")
            out[i] = np.frombuffer(self._fh.read(self.width * self.height), np.uint8).reshape(self.height, self.width)
        return out

    def close(self):
        self._fh.close()


class Cv2Video:
    """Fallback for compressed containers when OpenCV is installed (decode stays on the host)."""

    def __init__(self, path, default_fps=30.0):
        import cv2  # noqa: F401  (optional dependency)
        self._cv2 = cv2
        self._cap = cv2.VideoCapture(path)
        self.frame_count = int(self._cap.get(cv2.CAP_PROP_FRAME_COUNT))
        self.fps = float(self._cap.get(cv2.CAP_PROP_FPS) or default_fps)
        self.height, self.width = int(self._cap.get(4)), int(self._cap.get(3))
        self.channels = 3
        self._next = 0

    def read(self, start, count):
        if start != self._next:
            self._cap.set(self._cv2.CAP_PROP_POS_FRAMES, start)
        frames = []
        for _ in range(count):
            ok, frame = self._cap.read()
            if not ok:
                break
            frames.append(frame)
        self._next = start + len(frames)
        if not frames:
            return np.empty((0, self.height, self.width, 3), np.uint8)
        return np.stack(frames)

    def close(self):
        self._cap.release()


def open_video(path, default_fps=30.0):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return NpyVideo(path, default_fps)
    if ext == ".y4m":
        return Y4mVideo(path, default_fps)
    try:
        return Cv2Video(path, default_fps)
    except ImportError as exc:
        raise OSError(f"{path}: only .npy and .y4m can be read without OpenCV ({exc})") from exc

"""Frame sources for the detect-and-link path.

The reference reads frames with ``cv2.VideoCapture`` (ysmr/track_eval.py:65-93, 159).  OpenCV is
not a dependency of this package, so raw containers are read natively and ``cv2`` is used only when
it happens to be importable:

* ``.npy``  -- uint8 array [T, H, W] (gray) or [T, H, W, 3] (BGR), memory-mapped; fps from a
               ``<name>_meta.json`` side file (key ``fps``) or the tracking.ini value
* ``.y4m``  -- YUV4MPEG2; the luma plane is used as the gray frame
* ``.avi``  -- uncompressed AVI (what many microscope cameras write): 8-bit gray or 24-bit BGR DIB frames,
               bottom-up or top-down, OpenDML ``AVIX`` extensions included -- ``DeviceFrameFeed`` uploads the stored
               frames as they are and unpacks them on the device (``ysmr_unpack_dib_batch``); Motion-JPEG AVI if
               Pillow can be imported (frames decoded on the reader threads); other compressed streams go to cv2
* anything else -- ``cv2.VideoCapture`` if cv2 can be imported, otherwise an error

Every source exposes ``frame_count`` (what the container reports: cv2's CAP_PROP_FRAME_COUNT), optionally
``frames_available`` (what it holds, when that can differ), ``fps``, ``height``, ``width``, ``channels`` and
``read(start, count) -> uint8 ndarray [n, H, W(, 3)]`` (fewer than ``count`` frames at the end).
"""
from __future__ import annotations

import json
import os

import numpy as np

__all__ = ["open_video", "NpyVideo", "Y4mVideo", "AviVideo", "Cv2Video", "DeviceFrameFeed"]


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
        # positional reads for read_into: where the frames start in the file, and a descriptor of its own
        self._data0 = int(getattr(self._a, "offset", 0)) if self._a.flags["C_CONTIGUOUS"] else None
        self._fd = os.open(path, os.O_RDONLY) if self._data0 is not None else None
        meta = os.path.splitext(path)[0] + "_meta.json"
        try:
            with open(meta) as fh:
                self.fps = float(json.load(fh).get("fps", self.fps))
        except (OSError, ValueError, TypeError):
            pass

    def read(self, start, count):
        return np.ascontiguousarray(self._a[start:start + count])

    def read_into(self, start, count, out, pool=None):
        """Copy frames [start, start + count) into ``out[:n]`` (any writable uint8 array of the frame shape, e.g. pinned
        memory).  Positional reads side by side in ONE native call (``ysmr_file_read``, as many threads as the pool has, no
        Python between the pieces): the kernel copies page-cache pages straight into ``out``.  (Copying out of the memory map faults every 4-KiB page in first, and the faults of all
        threads queue on the one address space: 36 GB/s whatever the number of threads, which was the frame loop of
        ``track_bacteria``.)"""
        n = max(0, min(count, self.frame_count - start))
        if n == 0:
            return 0
        dst = out[:n]
        if self._fd is None or not dst.flags["C_CONTIGUOUS"]:
            np.copyto(dst, self._a[start:start + n])
            return n
        frame_bytes = self.height * self.width * self.channels
        threads = 1 if pool is None else pool._max_workers
        from . import _lib
        _lib.check(_lib.lib().ysmr_file_read(self._fd, dst.ctypes.data, n * frame_bytes, self._data0 + start * frame_bytes, threads),
                   "ysmr_file_read")
        return n

    def close(self):
        self._a = None
        if self._fd is not None:
            os.close(self._fd)
            self._fd = None


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
        self.read_into(start, n, out)
        return out

    def read_into(self, start, count, out, pool=None):
        """Luma planes of frames [start, start + count) straight into ``out[:n]`` (e.g. pinned memory)."""
        n = max(0, min(count, self.frame_count - start))
        for i in range(n):
            self._fh.seek(self._data0 + (start + i) * self._stride)
            marker = self._fh.readline()
            if not marker.startswith(b"FRAME"):
                raise ValueError(f"{self.path}: frame {start + i} has no FRAME marker")
            plane = memoryview(out[i]).cast("B")
            if self._fh.readinto(plane) != self.width * self.height:
                raise ValueError(f"{self.path}: frame {start + i} is truncated")
        return n

    def close(self):
        self._fh.close()


class AviVideo:
    """AVI (RIFF) without OpenCV: stream 0 must be video with BI_RGB 8- or 24-bit frames (or the raw
    gray fourccs Y800 / GREY / Y8), or Motion-JPEG (decoded with Pillow, if installed).  8-bit frames whose palette is the gray ramp are delivered as gray
    [H, W] -- what ``cv2.VideoCapture`` + ``COLOR_BGR2GRAY`` make of them --, other palettes are
    expanded to BGR; 24-bit frames are BGR as stored.  Raises ValueError for compressed streams."""

    _RAW_GRAY = (b"Y800", b"GREY", b"Y8  ")

    def __init__(self, path, default_fps=30.0):
        self.path = path
        self._fh = open(path, "rb")
        try:
            self._parse(path, default_fps)
        except BaseException:
            self._fh.close()           # (open_video falls through to OpenCV on ValueError: do not leak the handle)
            raise

    def _parse(self, path, default_fps):
        import struct
        fh = self._fh
        size = os.path.getsize(path)
        self.fps = float(default_fps)
        self._frames = []          # (offset, size) of every video chunk of stream 0
        declared = [0]
        bih = palette = None
        first_stream = True

        def walk(start, end, depth):
            nonlocal bih, palette, first_stream
            pos = start
            while pos + 8 <= end:
                fh.seek(pos)
                cid, csz = struct.unpack("<4sI", fh.read(8))
                body = pos + 8
                if cid in (b"RIFF", b"LIST"):
                    kind = fh.read(4)
                    if kind == b"movi":
                        walk_movi(body + 4, min(body + csz, end))
                    elif kind in (b"AVI ", b"AVIX", b"hdrl", b"strl"):
                        walk(body + 4, min(body + csz, end), depth + 1)
                elif cid == b"strh":
                    data = fh.read(min(csz, 56))
                    if first_stream and data[:4] == b"vids" and len(data) >= 28:
                        scale, rate = struct.unpack("<II", data[20:28])
                        if scale and rate:
                            self.fps = rate / scale
                        if len(data) >= 36:
                            declared[0] = struct.unpack("<I", data[32:36])[0]     # dwLength: the stream's frames
                elif cid == b"strf" and first_stream:
                    data = fh.read(csz)
                    if len(data) >= 40:
                        bih = struct.unpack("<IiiHHIIiiII", data[:40])
                        palette = data[40:]
                    first_stream = False
                pos = body + csz + (csz & 1)

        def walk_movi(start, end):
            pos = start
            while pos + 8 <= end:
                fh.seek(pos)
                cid, csz = struct.unpack("<4sI", fh.read(8))
                body = pos + 8
                if cid == b"LIST":                       # 'rec ' groups
                    walk_movi(body + 4, min(body + csz, end))
                elif cid[:2] == b"00" and cid[2:] in (b"db", b"dc"):
                    # an EMPTY chunk is AVI's "frame dropped by the capture, show the previous one again":
                    # it still occupies a frame slot (cv2.VideoCapture delivers the repeated frame, so
                    # POSITION_T and the frame count stay aligned with the reference)
                    if csz:
                        self._frames.append((body, csz))
                    elif self._frames:
                        self._frames.append(self._frames[-1])
                pos = body + csz + (csz & 1)

        if fh.read(4) != b"RIFF":
            raise ValueError(f"{path}: not a RIFF file")
        walk(0, size, 0)
        if bih is None:
            raise ValueError(f"{path}: no video stream format found")
        _, width, height, _, bits, compression, *_ = bih
        fourcc = struct.pack("<I", compression)
        self._jpeg = fourcc.upper() in (b"MJPG", b"JPEG")
        if self._jpeg:
            try:
                from PIL import Image
            except ImportError as exc:
                raise ValueError(f"{path}: Motion-JPEG AVI needs Pillow ({exc})") from exc
            self._Image = Image
            self.width, self.height = int(width), abs(int(height))
            self.frames_available = len(self._frames)
            self.frame_count = declared[0] or self.frames_available
            if not self.frames_available:
                raise ValueError(f"{path}: no frames")
            with Image.open(self._chunk(0)) as first:
                self.channels = 1 if first.mode == "L" else 3
            return
        if not (compression == 0 or (fourcc in self._RAW_GRAY and bits == 8)) or bits not in (8, 24):
            raise ValueError(f"{path}: only uncompressed 8/24-bit or Motion-JPEG AVI is read natively "
                             f"(fourcc {fourcc!r}, {bits} bit)")
        self.width, self.height = int(width), abs(int(height))
        self._bottom_up = height > 0 and compression == 0
        self._bytes_pp = bits // 8
        self._stride = (self.width * self._bytes_pp + 3) & ~3 if compression == 0 else self.width
        self._lut = None
        self.channels = 1 if bits == 8 else 3
        if bits == 8 and compression == 0 and len(palette) >= 4:
            pal = np.frombuffer(palette[: 4 * (len(palette) // 4)], np.uint8).reshape(-1, 4)[:256, :3]   # B, G, R
            ramp = np.arange(len(pal), dtype=np.uint8)[:, None]
            if not np.array_equal(pal, np.repeat(ramp, 3, axis=1)):
                full = np.zeros((256, 3), np.uint8)
                full[:len(pal)] = pal
                self._lut, self.channels = full, 3
        need = self._stride * self.height
        short = [i for i, f in enumerate(self._frames) if f[1] < need]
        if short:      # dropping them silently would shift every later frame number
            raise ValueError(f"{path}: video chunk {short[0]} holds {self._frames[short[0]][1]} bytes, a frame needs {need}")
        # what the container REPORTS (the stream header's length, cv2's CAP_PROP_FRAME_COUNT) and what it HOLDS: the
        # reference tolerates a report one above the frames it gets and treats anything else as a read error
        # (track_eval.py:170-178); track_bacteria applies that rule to these two numbers
        self.frames_available = len(self._frames)
        self.frame_count = declared[0] or self.frames_available

    def read(self, start, count):
        n = max(0, min(count, self.frames_available - start))
        out = np.empty((n, self.height, self.width) + ((3,) if self.channels == 3 else ()), np.uint8)
        self.read_into(start, n, out)
        return out

    def _chunk(self, i):
        import io
        off, size = self._frames[i]
        self._fh.seek(off)
        return io.BytesIO(self._fh.read(size))

    def _decode_jpeg(self, blob, dst):
        with self._Image.open(blob) as im:
            if self.channels == 1:
                dst[...] = np.asarray(im.convert("L"))
            else:
                dst[...] = np.asarray(im.convert("RGB"))[:, :, ::-1]     # BGR, what cv2.VideoCapture delivers

    def read_into(self, start, count, out, pool=None):
        n = max(0, min(count, self.frames_available - start))
        if self._jpeg:
            blobs = [self._chunk(start + i) for i in range(n)]           # file access stays on this thread
            if pool is None:
                for i in range(n):
                    self._decode_jpeg(blobs[i], out[i])
            else:
                for job in [pool.submit(self._decode_jpeg, blobs[i], out[i]) for i in range(n)]:
                    job.result()
            return n
        raw = np.empty((self.height, self._stride), np.uint8)
        used = self.width * self._bytes_pp
        for i in range(n):
            self._fh.seek(self._frames[start + i][0])
            if self._fh.readinto(memoryview(raw).cast("B")) != raw.size:
                raise ValueError(f"{self.path}: frame {start + i} is truncated")
            rows = raw[::-1, :used] if self._bottom_up else raw[:, :used]
            if self._lut is not None:
                out[i] = self._lut[rows]
            elif self.channels == 3:
                out[i] = rows.reshape(self.height, self.width, 3)
            else:
                out[i] = rows
        return n

    # ---- frames as they are in the file, for unpacking on the device (DeviceFrameFeed, ysmr_unpack_dib_batch)
    @property
    def raw_layout(self):
        """None, or what ``ysmr_unpack_dib_batch`` needs to turn the chunk bodies of this file into frames:
        (bytes per stored frame, bytes per pixel, row stride, bottom-up, palette u8 [256, 3] or None)."""
        if self._jpeg:
            return None
        return self._stride * self.height, self._bytes_pp, self._stride, bool(self._bottom_up), self._lut

    def read_raw_into(self, start, count, out, pool=None):
        """Chunk bodies of frames [start, start + count) into ``out[:n]`` (u8 [n, bytes per stored frame], e.g.
        pinned memory): file reads only, no per-pixel work on the host; positional reads (``os.preadv``), spread
        over the threads of ``pool`` if one is given."""
        n = max(0, min(count, self.frames_available - start))
        need = self._stride * self.height
        fd = self._fh.fileno()

        def one(i):
            got, view = 0, memoryview(out[i]).cast("B")[:need]
            while got < need:                                   # (a read may return short of a large request)
                k = os.preadv(fd, [view[got:]], self._frames[start + i][0] + got)
                if k <= 0:
                    raise ValueError(f"{self.path}: frame {start + i} is truncated")
                got += k

        if pool is None or n < 2:
            for i in range(n):
                one(i)
        else:
            for job in [pool.submit(one, i) for i in range(n)]:
                job.result()
        return n

    def close(self):
        self._fh.close()


class Cv2Video:
    """Fallback for compressed containers when OpenCV is installed (decode stays on the host)."""

    def __init__(self, path, default_fps=30.0):
        import cv2  # noqa: F401  (optional dependency)
        self._cv2 = cv2
        self._cap = cv2.VideoCapture(path)
        self.frame_count = int(self._cap.get(cv2.CAP_PROP_FRAME_COUNT))   # what the container reports ...
        self.frames_available = 1 << 62                                    # ... read() tells where it really ends
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


_NODE_CPUS = {}      # GPU index -> CPUs of its NUMA node (sysfs; asked once per process)


def cpus_near_gpu(device):
    """The CPUs of the NUMA node the GPU hangs off that this thread may use (None: unknown, or fewer than four)."""
    try:
        import torch
        from . import dist
        index = torch.device(device).index or 0
        if index not in _NODE_CPUS:
            _NODE_CPUS[index] = dist.local_cpus(dist.pci_bus_id(index))
        cpus = _NODE_CPUS[index]
        if not cpus:
            return None
        cpus = cpus & os.sched_getaffinity(0)
        return cpus if len(cpus) >= 4 else None
    except (AttributeError, OSError, ValueError):
        return None


class _ThreadOn:
    """``with _ThreadOn(cpus):`` -- the calling THREAD runs on ``cpus`` inside the block (Linux: sched_setaffinity(0) is the
    calling thread's mask, and the threads it starts inherit it), and on what it was allowed before afterwards."""

    def __init__(self, cpus):
        self.cpus, self.before = cpus, None

    def __enter__(self):
        if self.cpus:
            try:
                self.before = os.sched_getaffinity(0)
                os.sched_setaffinity(0, self.cpus)
            except (AttributeError, OSError):
                self.before = None
        return self

    def __exit__(self, *exc):
        if self.before is not None:
            try:
                os.sched_setaffinity(0, self.before)
            except OSError:
                pass
        return False


class DeviceFrameFeed:
    """Batches of a video as device tensors, read and uploaded ahead of their use.

    A producer thread copies batch i into one of ``depth`` pinned staging buffers (split over
    ``readers`` threads) and uploads it on its own HIP stream into one of ``depth`` device buffers,
    while the consumer works on earlier batches.  Iterating yields ``(frames_dev, first_frame, count,
    slot)``; the consumer's stream already waits for the upload.  When the kernels that read
    ``frames_dev`` have been issued, the consumer calls ``release(slot, event)`` with an event recorded
    behind them: the slot's device buffer is overwritten only after that event.

    ``near_gpu``: the staging buffers are allocated, and the producer and its readers run, on the CPUs of the GPU's own NUMA node
    (the caller's threads are left where they are).  On a two-socket host the copy into pinned memory and the DMA out of it
    otherwise cross the sockets' link half of the time: reads 50-60 GB/s and uploads 35-45 while both run, against 85-90 and
    56 (`scripts/feed_copy_timeline.py`)."""

    def __init__(self, video, batch, device, depth=3, readers=16, pieces=4, near_gpu=True):
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor

        import torch
        self.video, self.B, self.device, self.depth = video, int(batch), torch.device(device), int(depth)
        self.pieces = max(1, int(pieces))
        shape = (self.B, video.height, video.width) + ((3,) if video.channels == 3 else ())
        # uncompressed AVI: the stored frames (bottom-up, padded rows, palette indices) go to the device as they are
        # and are unpacked there; every other source delivers finished frames
        self._raw = getattr(video, "raw_layout", None)
        self._cpus = cpus_near_gpu(self.device) if near_gpu else None
        with _ThreadOn(self._cpus):      # (pinned pages are the calling thread's node's)
            self._allocate(shape)
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self._uploaded = [None] * self.depth     # event: H2D copy out of pinned[slot] finished
        self._released = [None] * self.depth     # event posted by the consumer (or True: never used / free)
        self._cv = threading.Condition()
        for k in range(self.depth):
            self._released[k] = True
        self._q = queue.Queue(maxsize=self.depth)
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(readers))) if readers > 1 else None
        self._stop = False
        self._thread = threading.Thread(target=self._produce, name="ysmr-frame-feed", daemon=True)
        self._thread.start()

    def _allocate(self, shape):
        import torch
        if self._raw is not None:
            raw_bytes, _, _, _, palette = self._raw
            self._pinned = [torch.empty((self.B, raw_bytes), dtype=torch.uint8, pin_memory=True) for _ in range(self.depth)]
            self._raw_dev = [torch.empty((self.B, raw_bytes), dtype=torch.uint8, device=self.device) for _ in range(self.depth)]
            self._palette = None if palette is None else torch.from_numpy(np.ascontiguousarray(palette)).to(self.device)
        else:
            self._pinned = [torch.empty(shape, dtype=torch.uint8, pin_memory=True) for _ in range(self.depth)]
        self._dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(self.depth)]

    def _produce(self):
        import torch
        try:
            torch.cuda.set_device(self.device)
            if self._cpus:
                try:
                    os.sched_setaffinity(0, self._cpus)      # this thread, and the readers it starts
                except OSError:
                    pass
            i = 0
            # (every frame the source HOLDS, which need not be the number it reports: track_bacteria applies the
            # reference's rule to the difference)
            for f0 in range(0, getattr(self.video, "frames_available", self.video.frame_count), self.B):
                slot = i % self.depth
                with self._cv:
                    while self._released[slot] is None and not self._stop:
                        self._cv.wait(0.05)
                    if self._stop:
                        return
                    released, self._released[slot] = self._released[slot], None
                if self._uploaded[slot] is not None:
                    self._uploaded[slot].synchronize()          # staging buffer free again
                host = self._pinned[slot].numpy()
                # A batch is read and uploaded in `pieces` parts: the upload of a part starts when the part has been read, not
                # when the batch has (the first batch of a 1228 x 922 file is 280 MB: 5 ms of reading before the first byte
                # crossed the bus, 4 of the 73 ms a 1920-frame file takes)
                step = max(1, -(-self.B // self.pieces))
                n = 0
                for c0 in range(0, self.B, step):
                    want = min(step, self.B - c0)
                    if self._raw is not None:
                        got = self.video.read_raw_into(f0 + c0, want, host[c0:c0 + want], self._pool)
                    elif hasattr(self.video, "read_into"):
                        got = self.video.read_into(f0 + c0, want, host[c0:c0 + want], self._pool)
                    else:
                        part = self.video.read(f0 + c0, want)
                        got = part.shape[0]
                        host[c0:c0 + got] = part
                    if got:
                        with torch.cuda.stream(self._copy_stream):
                            if released is not True:
                                self._copy_stream.wait_event(released)  # the kernels that read this device buffer are done
                                released = True
                            dst = self._raw_dev[slot] if self._raw is not None else self._dev[slot]
                            dst[c0:c0 + got].copy_(self._pinned[slot][c0:c0 + got], non_blocking=True)
                    n += got
                    if got < want:
                        break
                if n == 0:
                    break
                with torch.cuda.stream(self._copy_stream):
                    if self._raw is not None:
                        from . import _lib
                        raw_bytes, bpp, stride, bottom_up, _ = self._raw
                        _lib.check(_lib.lib().ysmr_unpack_dib_batch(
                            self._copy_stream.cuda_stream, self._raw_dev[slot].data_ptr(), n, raw_bytes, self.video.height,
                            self.video.width, bpp, stride, int(bottom_up),
                            None if self._palette is None else self._palette.data_ptr(), self._dev[slot].data_ptr()),
                            "ysmr_unpack_dib_batch")
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                self._uploaded[slot] = ev
                self._q.put((slot, f0, n, ev))
                i += 1
            self._q.put(None)
        except BaseException as exc:   # hand the failure to the consumer
            self._q.put(exc)

    def __iter__(self):
        import torch
        while True:
            item = self._q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            slot, f0, n, ev = item
            torch.cuda.current_stream(self.device).wait_event(ev)
            yield self._dev[slot][:n], f0, n, slot

    def release(self, slot, event):
        with self._cv:
            self._released[slot] = event
            self._cv.notify_all()

    def close(self):
        self._stop = True
        with self._cv:
            self._cv.notify_all()
        try:
            while True:                # unblock a producer waiting on a full queue
                self._q.get_nowait()
        except Exception:
            pass
        self._thread.join(timeout=5.0)
        if self._pool is not None:
            self._pool.shutdown(wait=False)


def open_video(path, default_fps=30.0):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return NpyVideo(path, default_fps)
    if ext == ".y4m":
        return Y4mVideo(path, default_fps)
    if ext == ".avi":
        try:
            return AviVideo(path, default_fps)
        except ValueError:
            pass                                   # compressed: OpenCV's business
    try:
        return Cv2Video(path, default_fps)
    except ImportError as exc:
        raise OSError(f"{path}: only .npy and .y4m can be read without OpenCV ({exc})") from exc

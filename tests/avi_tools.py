"""Writer of minimal uncompressed AVI files for the frame-source tests."""
import numpy as np


def _riff(fourcc, payload):
    return fourcc + len(payload).to_bytes(4, "little") + payload + (b"\0" if len(payload) & 1 else b"")


def write_avi(path, frames, bits, fps=(30000, 1001), top_down=False, palette=None, split=None, jpeg=None, dropped=(),
              truncated=None, declared=None):
    """Minimal AVI (uncompressed DIB frames, or the given JPEG blobs as Motion-JPEG): hdrl (avih, strl(strh, strf)), movi with 00db chunks (+ an audio chunk that
    must be skipped), optionally continued in a second RIFF AVIX.  ``declared``: the frame count written into the stream
    header when it is to differ from the frames stored."""
    import struct
    n, h, w = frames.shape[:3]
    stride = (w * bits // 8 + 3) & ~3
    def dib(f):
        rows = f.reshape(h, -1)
        rows = rows if top_down else rows[::-1]
        out = np.zeros((h, stride), np.uint8)
        out[:, :rows.shape[1]] = rows
        return out.tobytes()
    strh = struct.pack("<4s4sIHHIIIIIIII4H", b"vids", b"DIB ", 0, 0, 0, 0, fps[1], fps[0], 0, n if declared is None else declared, stride * h, 0, 0, 0, 0, w, h)
    compression = int.from_bytes(b"MJPG", "little") if jpeg is not None else 0
    bih = struct.pack("<IiiHHIIiiII", 40, w, -h if top_down else h, 1, bits, compression, stride * h, 0, 0, 0, 0)
    if bits == 8 and jpeg is None:
        pal = palette if palette is not None else np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, axis=1)
        bih += np.concatenate([pal, np.zeros((256, 1), np.uint8)], axis=1).tobytes()
    hdrl = b"hdrl" + _riff(b"avih", bytes(56)) + _riff(b"LIST", b"strl" + _riff(b"strh", strh) + _riff(b"strf", bih))
    chunks = [_riff(b"00db", dib(f)) for f in frames] if jpeg is None else [_riff(b"00dc", blob) for blob in jpeg]
    for i in dropped:                                             # empty chunk = "repeat the previous frame"
        chunks[i] = _riff(b"00db", b"")
    if truncated is not None:
        chunks[truncated] = _riff(b"00db", dib(frames[truncated])[:-8])
    chunks.insert(1, _riff(b"01wb", b"abc"))                      # odd-sized audio chunk in between
    split = n if split is None else split
    body = _riff(b"RIFF", b"AVI " + _riff(b"LIST", hdrl) + _riff(b"LIST", b"movi" + b"".join(chunks[:split + 1])))
    if split < n:
        body += _riff(b"RIFF", b"AVIX" + _riff(b"LIST", b"movi" + b"".join(chunks[split + 1:])))
    with open(path, "wb") as fh:
        fh.write(body)

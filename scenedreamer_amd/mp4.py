"""Minimal MP4 (ISO base media file) writer for a Motion-JPEG video track.

The reference appends every rendered frame to an `imageio` MP4 writer at 10 fps
(imaginaire/generators/scenedreamer.py:560, :631; imageio/ffmpeg are not part of this image).  This module writes
the same kind of artefact with nothing but the standard library + Pillow's JPEG encoder: one `jpeg` video sample
per frame, `ftyp` + `mdat` + `moov` (mvhd / trak / mdia / minf / stbl with stsd, stts, stsc, stsz, co64).
Players built on ffmpeg (VLC, mpv, browsers via ffmpeg.wasm) read it; the frames are intra-coded, so the file is
also trivially seekable and a frame can be pulled out with `read_frames` below (used by the tests).
"""
import io
import struct

try:
    from PIL import Image
except ImportError:
    Image = None


def _box(kind, payload):
    return struct.pack(">I4s", 8 + len(payload), kind) + payload


def _full(kind, version, flags, payload):
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


_MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


class Mp4MjpegWriter:
    """Append uint8 RGB frames [H,W,3]; `close()` writes the index.  Not thread-safe: call from one thread."""

    def __init__(self, path, fps=10, quality=92):
        if Image is None:
            raise RuntimeError("Mp4MjpegWriter needs Pillow for JPEG encoding")
        self.f = open(path, "wb")
        self.fps, self.quality = int(fps), int(quality)
        self.sizes, self.offsets = [], []
        self.wh = None
        self.f.write(_box(b"ftyp", b"isom" + struct.pack(">I", 0x200) + b"isomiso2mp41"))
        self.mdat_pos = self.f.tell()
        self.f.write(struct.pack(">I4sQ", 1, b"mdat", 0))      # 64-bit size, patched in close()

    def append(self, rgb):
        h, w = int(rgb.shape[0]), int(rgb.shape[1])
        if self.wh is None:
            self.wh = (w, h)
        elif self.wh != (w, h):
            raise ValueError(f"frame size changed: {self.wh} -> {(w, h)}")
        buf = io.BytesIO()
        Image.fromarray(rgb, "RGB").save(buf, format="JPEG", quality=self.quality, subsampling=0)
        data = buf.getvalue()
        self.offsets.append(self.f.tell())
        self.sizes.append(len(data))
        self.f.write(data)

    def close(self):
        if self.f is None:
            return
        end = self.f.tell()
        n = len(self.sizes)
        w, h = self.wh or (0, 0)
        ts = self.fps                     # timescale = fps -> every sample lasts 1 tick
        stsd_entry = (struct.pack(">6xH", 1) + struct.pack(">HHIII", 0, 0, 0, 0, 0) + struct.pack(">HH", w, h) +
                      struct.pack(">IIIH", 0x480000, 0x480000, 0, 1) + bytes(32) + struct.pack(">Hh", 24, -1))
        stbl = _box(b"stbl",
                    _full(b"stsd", 0, 0, struct.pack(">I", 1) + _box(b"jpeg", stsd_entry)) +
                    _full(b"stts", 0, 0, struct.pack(">III", 1, n, 1)) +
                    _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, 1, 1)) +
                    _full(b"stsz", 0, 0, struct.pack(">II", 0, n) + b"".join(struct.pack(">I", s) for s in self.sizes)) +
                    _full(b"co64", 0, 0, struct.pack(">I", n) + b"".join(struct.pack(">Q", o) for o in self.offsets)))
        dinf = _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1) + _full(b"url ", 0, 1, b"")))
        minf = _box(b"minf", _full(b"vmhd", 0, 1, struct.pack(">HHHH", 0, 0, 0, 0)) + dinf + stbl)
        mdia = _box(b"mdia",
                    _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, ts, n, 0x55C4, 0)) +
                    _full(b"hdlr", 0, 0, struct.pack(">I4s12x", 0, b"vide") + b"sdnative video\0") + minf)
        tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, n) + struct.pack(">8xhhhH", 0, 0, 0, 0) + _MATRIX +
                     struct.pack(">II", w << 16, h << 16))
        mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIII", 0, 0, ts, n) + struct.pack(">IH10x", 0x10000, 0x100) + _MATRIX +
                     bytes(24) + struct.pack(">I", 2))
        self.f.write(_box(b"moov", mvhd + _box(b"trak", tkhd + mdia)))
        self.f.seek(self.mdat_pos + 8)
        self.f.write(struct.pack(">Q", end - self.mdat_pos))
        self.f.close()
        self.f = None


# ---------------------------------------------------------------------------------------------------- reader (tests)
def _walk(buf, start, end):
    pos = start
    while pos + 8 <= end:
        size, kind = struct.unpack_from(">I4s", buf, pos)
        hdr = 8
        if size == 1:
            size = struct.unpack_from(">Q", buf, pos + 8)[0]
            hdr = 16
        yield kind, pos + hdr, pos + size
        pos += size


def _find(buf, path, start=0, end=None):
    end = len(buf) if end is None else end
    for kind, a, b in _walk(buf, start, end):
        if kind == path[0]:
            return (a, b) if len(path) == 1 else _find(buf, path[1:], a, b)
    raise KeyError(path)


def read_frames(path):
    """(fps, [uint8 RGB arrays]) of a file written by Mp4MjpegWriter (structure-checking reader for the tests)."""
    import numpy as np
    buf = open(path, "rb").read()
    a, _ = _find(buf, [b"moov", b"trak", b"mdia", b"mdhd"])
    ts, dur = struct.unpack_from(">II", buf, a + 12)
    sa, _ = _find(buf, [b"moov", b"trak", b"mdia", b"minf", b"stbl", b"stsz"])
    n = struct.unpack_from(">I", buf, sa + 8)[0]
    sizes = struct.unpack_from(f">{n}I", buf, sa + 12)
    ca, _ = _find(buf, [b"moov", b"trak", b"mdia", b"minf", b"stbl", b"co64"])
    offs = struct.unpack_from(f">{n}Q", buf, ca + 8)
    da, _ = _find(buf, [b"moov", b"trak", b"mdia", b"minf", b"stbl", b"stsd"])
    assert buf[da + 12:da + 16] == b"jpeg" and dur == n
    frames = [np.asarray(Image.open(io.BytesIO(buf[o:o + s])).convert("RGB")) for o, s in zip(offs, sizes)]
    return ts, frames

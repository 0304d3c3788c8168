"""Minimal mirror of mindaudio.data.io.read (io.py:552-760) for the loader: integer-PCM RIFF/WAVE -> float64 in
[-1, 1) and the sample rate.  Host-side file parsing only; everything numeric happens on the device afterwards."""
import struct

import numpy as np

__all__ = ["read"]

_SCALE = {8: 128.0, 16: 32768.0, 32: 2147483648.0}  # io.py:741-745


def read_pcm16(path):
    """(samples int16 (n,), rate) of a mono 16-bit PCM WAV file without leaving integers - the loader's fast path
    (conformer/dataset.py: the samples go to the device as they lie in the file); None for every other format (the caller then takes
    `read`)."""
    with open(path, "rb") as fh:
        buf = fh.read()
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        return None
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        size = struct.unpack_from("<I", buf, pos + 4)[0]
        tag = buf[pos:pos + 4]
        if tag == b"fmt ":
            fmt = struct.unpack_from("<HHIIHH", buf, pos + 8)
        elif tag == b"data":
            data = (pos + 8, min(size, len(buf) - pos - 8))
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or fmt[0] != 1 or fmt[1] != 1 or fmt[5] != 16:
        return None
    return np.frombuffer(buf, dtype="<i2", count=data[1] // 2, offset=data[0]), fmt[2]


def read(file, offset=0.0, duration=None):
    fh = open(file, "rb") if isinstance(file, (str, bytes)) else file
    try:
        head = fh.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError("not a RIFF/WAVE file")
        fmt = payload = None
        while True:
            hdr = fh.read(8)
            if len(hdr) < 8:
                break
            size = struct.unpack("<I", hdr[4:])[0]
            body = fh.read(size)
            if size & 1:
                fh.read(1)
            if hdr[:4] == b"fmt ":
                fmt = struct.unpack("<HHIIHH", body[:16])
            elif hdr[:4] == b"data":
                payload = body
    finally:
        if fh is not file:
            fh.close()
    if fmt is None or payload is None:
        raise ValueError("missing fmt/data chunk")
    tag, channels, rate, _, _, bits = fmt
    if tag != 1 or bits not in _SCALE:
        raise NotImplementedError("only integer PCM 8/16/32-bit WAV files are on the loader path")
    if bits == 8:
        pcm = np.frombuffer(payload, dtype=np.uint8).astype(np.int16) - 128
    else:
        pcm = np.frombuffer(payload, dtype="<i2" if bits == 16 else "<i4")
    if channels > 1:
        pcm = pcm[:pcm.size // channels * channels].reshape(-1, channels)
    first = int(round(offset * rate))
    last = pcm.shape[0] if duration is None else min(pcm.shape[0], first + int(round(duration * rate)))
    return pcm[first:last] / _SCALE[bits], rate

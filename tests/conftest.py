import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        return cache[name]

    return load


def write_synthetic_nd2(path, frames_yxc, row_pad_bytes: int = 0, loops=None, calibration=None, period_ms=None):
    """Write an uncompressed ND2 container (chunk map + lite-variant attributes) holding
    ``frames_yxc`` = (N, Y, X, C) uint16 frames; ``row_pad_bytes`` pads every pixel row (uiWidthBytes > X*C*2);
    ``loops`` = [(eType, count), ...] outermost first writes an ``ImageMetadataLV!`` chunk with that experiment tree
    (eType 1 time, 2 positions, 4 z stack, 6 spectral, 8 non-equidistant time); ``period_ms`` gives its time loops a
    ``dPeriod``; ``calibration`` (um per pixel) writes an ``ImageMetadataSeqLV|0!`` chunk with ``dCalibration``.
    The reference's fixture files do not travel to the GPU box; their pixels are pinned in tests/golden."""
    import struct

    frames_yxc = np.asarray(frames_yxc, dtype="<u2")
    N, H, W, C = frames_yxc.shape
    stride = W * C * 2 + row_pad_bytes

    def lv(typ, name, payload):
        n = (name + "\x00").encode("utf-16-le")
        return bytes([typ, len(n) // 2]) + n + payload

    items = b"".join([
        lv(3, "uiWidth", struct.pack("<I", W)), lv(3, "uiWidthBytes", struct.pack("<I", stride)),
        lv(3, "uiHeight", struct.pack("<I", H)), lv(3, "uiComp", struct.pack("<I", C)),
        lv(2, "uiBpcInMemory", struct.pack("<i", 16)), lv(3, "uiSequenceCount", struct.pack("<I", N)),
    ])
    name = ("SLxImageAttributes" + "\x00").encode("utf-16-le")
    head = bytes([11, len(name) // 2]) + name
    attrs = head + struct.pack("<IQ", 6, len(head) + 12 + len(items)) + items + b"\x00" * (6 * 8)

    def chunk(cname, payload):
        nm = cname + b"\x00" * (32 - len(cname))
        return struct.pack("<IIQ", 0x0ABECEDA, len(nm), len(payload)) + nm + payload

    chunks = [(b"ImageAttributesLV!", attrs)]

    def level(lname, items):
        nm = (lname + "\x00").encode("utf-16-le")
        hd = bytes([11, len(nm) // 2]) + nm
        body = b"".join(items)
        return hd + struct.pack("<IQ", len(items), len(hd) + 12 + len(body)) + body + b"\x00" * (8 * len(items))

    if calibration is not None:
        chunks.append((b"ImageMetadataSeqLV|0!",
                       level("SLxPictureMetadata", [lv(6, "dCalibration", struct.pack("<d", float(calibration)))])))
    if loops:
        def experiment(rest):
            (etype, count), deeper = rest[0], rest[1:]
            pars = [lv(3, "uiCount", struct.pack("<I", count)), lv(6, "dStart", struct.pack("<d", 0.0))]
            if period_ms is not None and etype in (1, 8):
                pars.append(lv(6, "dPeriod", struct.pack("<d", float(period_ms))))
            items = [lv(3, "eType", struct.pack("<I", etype)),
                     level("uLoopPars", pars),
                     lv(3, "uiNextLevelCount", struct.pack("<I", 1 if deeper else 0))]
            if deeper:
                items.append(level("ppNextLevelEx", [level("", experiment(deeper))]))
            return items

        chunks.append((b"ImageMetadataLV!", level("SLxExperiment", experiment(list(loops)))))
    for i in range(N):
        rows = b"".join(frames_yxc[i, y].tobytes() + b"\xAB" * row_pad_bytes for y in range(H))
        chunks.append((b"ImageDataSeq|%d!" % i, b"\x00" * 8 + rows))
    blob = b""
    entries = []
    for cname, payload in chunks:
        entries.append((cname, len(blob), len(payload)))
        blob += chunk(cname, payload)
    mp = b"".join(c + struct.pack("<QQ", off, size) for c, off, size in entries)
    mp += b"ND2 CHUNK MAP SIGNATURE 0000001!" + struct.pack("<Q", len(blob))
    map_off = len(blob)
    blob += chunk(b"ND2 FILEMAP SIGNATURE NAME 0001!", mp)
    blob += struct.pack("<Q", map_off)
    with open(path, "wb") as f:
        f.write(blob)
    return path

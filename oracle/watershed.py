"""ctypes wrapper for the C watershed oracle (oracle/watershed.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_ws.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/watershed.c with gcc into oracle/_build/ (outputs are git-ignored)."""
    src = os.path.join(_HERE, "watershed.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_watershed.restype = ctypes.c_int
        _lib.oracle_watershed.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    return _lib


def watershed(image, markers, mask=None, connectivity: int = 1) -> np.ndarray:
    """``skimage.segmentation.watershed(image, markers, connectivity, mask=mask)`` -> int32 labels."""
    lib = _load()
    image = np.ascontiguousarray(image, dtype=np.float64)
    markers = np.ascontiguousarray(markers, dtype=np.int32)
    H, W = image.shape
    out = np.zeros((H, W), dtype=np.int32)
    mptr = None
    if mask is not None:
        mask = np.ascontiguousarray(np.asarray(mask, dtype=bool).view(np.uint8))
        mptr = mask.ctypes.data
    rc = lib.oracle_watershed(image.ctypes.data, markers.ctypes.data, mptr, out.ctypes.data, H, W, connectivity)
    if rc != 0:
        raise MemoryError("oracle_watershed failed")
    return out

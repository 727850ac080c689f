"""ctypes wrapper for the C integer-image labelling oracle (oracle/clabel.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_clabel.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/clabel.c with gcc into oracle/_build/ (outputs are git-ignored)."""
    src = os.path.join(_HERE, "clabel.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_label_int64.restype = ctypes.c_int64
        _lib.oracle_label_int64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int]
    return _lib


def label_int(image: np.ndarray, connectivity: int = 2) -> np.ndarray:
    """``skimage.measure.label(int_image, connectivity=...)`` on a 2-D image -> int64 labels."""
    lib = _load()
    image = np.ascontiguousarray(image, dtype=np.int64)
    H, W = image.shape
    out = np.empty((H, W), dtype=np.int64)
    rc = lib.oracle_label_int64(image.ctypes.data, out.ctypes.data, H, W, int(connectivity))
    if rc < 0:
        raise MemoryError("oracle_label_int64 failed")
    return out

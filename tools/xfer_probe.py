"""Host <-> device transfer costs of the reference-level API's arrays (one 2048^2 plane)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import device as dv
from arcadia_microscopy_tools_amd.device import get_context, pinned_empty

ctx = get_context()
lib = ctx._lib


def t(name, fn, n=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print(f"{name:52s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms")


pin = pinned_empty((32 << 20,), np.uint8)
dbuf = ctx.empty((32 << 20,), np.uint8)


def raw_h2d(nb):
    lib.amt_memcpy_h2d(ctx.handle, dbuf.ptr, pin.array.ctypes.data, nb); ctx.synchronize()


def raw_d2h(nb):
    lib.amt_memcpy_d2h(ctx.handle, pin.array.ctypes.data, dbuf.ptr, nb); ctx.synchronize()


for mb in (8, 16, 32):
    t(f"pinned H2D {mb} MB", lambda: raw_h2d(mb << 20))
    t(f"pinned D2H {mb} MB", lambda: raw_d2h(mb << 20))
u16 = np.random.default_rng(0).integers(0, 60000, (2048, 2048)).astype(np.uint16)
i64 = np.random.default_rng(0).integers(0, 1000, (2048, 2048)).astype(np.int64)
t("asarray u16 8 MB", lambda: ctx.asarray(u16))
t("asarray int64 -> int32 (32 MB host, 16 MB bus)", lambda: ctx.asarray(i64, dtype=np.int32))
t("asarray int64 as is (32 MB)", lambda: ctx.asarray(i64))
d32 = ctx.asarray(i64, dtype=np.int32)
t("numpy() int32 16 MB", lambda: d32.numpy())
t("numpy_int64() 16 MB bus -> 32 MB host", lambda: d32.numpy_int64())
t("np.empty + touch 32 MB (1 thread)", lambda: np.empty((2048, 2048), np.int64).fill(0))
t("host astype int64->int32", lambda: i64.astype(np.int32))
f64 = ctx.asarray(np.random.default_rng(1).random((2048, 2048)))
t("numpy() float64 32 MB", lambda: f64.numpy())
dv._result_pool.cap_out = 0
print("-- page-locked result blocks off:")
t("numpy() int32 16 MB", lambda: d32.numpy())
t("numpy_int64() 16 MB bus -> 32 MB host", lambda: d32.numpy_int64())
t("numpy() float64 32 MB", lambda: f64.numpy())
for k in (4, 8):
    dv._PIPE_CHUNKS = k
    t(f"  chunks={k}: asarray int64->int32", lambda: ctx.asarray(i64, dtype=np.int32))
    t(f"  chunks={k}: numpy_int64", lambda: d32.numpy_int64())
    t(f"  chunks={k}: asarray u16", lambda: ctx.asarray(u16))

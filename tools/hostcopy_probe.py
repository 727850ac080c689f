"""Host copy into a staging block: numpy's copyto against amt_host_copy (streaming stores), on 1..N pool threads.
Runs without a GPU (the destination is then ordinary memory); on a GPU box the destination is page-locked."""
import ctypes
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, ".")
from arcadia_microscopy_tools_amd import _hip  # noqa: E402

lib = _hip.load_library()
rng = np.random.default_rng(0)
# correctness on odd sizes and alignments
for n, od, os_ in [(0, 0, 0), (1, 3, 5), (63, 1, 0), (64, 0, 7), (1000003, 5, 9), (1 << 20, 8, 0)]:
    src = rng.integers(0, 256, n + 32, dtype=np.uint8)
    dst = np.zeros(n + 64, np.uint8)
    lib.amt_host_copy(ctypes.c_void_p(dst.ctypes.data + od), ctypes.c_void_p(src.ctypes.data + os_), n)
    assert np.array_equal(dst[od:od + n], src[os_:os_ + n]) and not dst[:od].any() and not dst[od + n:].any(), n
print("amt_host_copy: exact on odd sizes / alignments")

N = 8
IMG = 4 * 2048 * 2048 * 2
srcs = [rng.integers(0, 65535, IMG // 2, dtype=np.uint16).view(np.uint8) for _ in range(N)]
try:
    from arcadia_microscopy_tools_amd.device import pinned_empty

    pin = pinned_empty((N, IMG), np.uint8)
    dst = pin.array
    kind = "page-locked"
except Exception:
    dst = np.zeros((N, IMG), np.uint8)
    kind = "pageable (no GPU here)"


def np_piece(d, s):
    np.copyto(d, s)


def nt_piece(d, s):
    lib.amt_host_copy(ctypes.c_void_p(d.ctypes.data), ctypes.c_void_p(s.ctypes.data), d.nbytes)


for threads in (1, 2, 4, 8, 16):
    with ThreadPoolExecutor(max_workers=threads) as ex:
        for name, fn in (("numpy copyto", np_piece), ("streaming stores", nt_piece)):
            best = 1e9
            for rep in range(4):
                t0 = time.perf_counter()
                futs = []
                for j in range(N):
                    step = IMG // 4
                    futs += [ex.submit(fn, dst[j, o:o + step], srcs[j][o:o + step]) for o in range(0, IMG, step)]
                for f in futs:
                    f.result()
                best = min(best, time.perf_counter() - t0)
            print(f"{kind}, {threads:2d} threads, {name:16s}: {N * IMG / best / 1e9:6.1f} GB/s")
assert all(np.array_equal(dst[j], srcs[j]) for j in range(N))

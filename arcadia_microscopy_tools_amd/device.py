"""Device context and device-resident arrays (HBM buffers addressed through the C ABI).

``Context`` owns one HIP stream + scratch arena on one GPU; it is NOT thread-safe, so every host thread
gets its own via ``get_context()`` (thread-local).  That is what makes the operators re-entrant for
``Pipeline(parallel=True)`` (reference: R/pipeline.py:145-146).

``DeviceArray`` is a typed, shaped view of device memory.  Leading axes are independent planes; the last
two axes are (Y, X).  A (C, Y, X) field of view is channel-major, so ``get_channel_intensities`` on the
device is a pointer offset (reference: R/microscopy.py:279-282 returns a numpy view).
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

from . import _hip

_DTYPE_CODE = {
    np.dtype(np.uint8): _hip.U8,
    np.dtype(np.bool_): _hip.U8,
    np.dtype(np.uint16): _hip.U16,
    np.dtype(np.int32): _hip.I32,
    np.dtype(np.float64): _hip.F64,
    np.dtype(np.int64): _hip.I64,
    np.dtype(np.float32): _hip.F32,
    np.dtype(np.uint32): _hip.I32,
}


_copy_pool = None
_PIPE_MIN = 4 << 20   # device bytes from which a staged transfer is cut into chunks
_PIPE_CHUNKS = 8      # chunks per transfer: the DMA of one overlaps the host-side copy / conversion of the others


def _pool4():
    """The host-copy threads shared by every context of the process (staging copies, conversions, extrema): four when
    one thread drives the GPU, more when the process may run several worker threads with a context each
    (AMT_COPY_THREADS; default = half the cores this process may use, between 4 and 16)."""
    global _copy_pool
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor

        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 4
        nthreads = int(os.environ.get("AMT_COPY_THREADS", str(max(4, min(16, cores // 2)))))
        _copy_pool = ThreadPoolExecutor(max_workers=max(1, nthreads), thread_name_prefix="amt-copy")
    return _copy_pool


def _host_copy(dst: np.ndarray, src: np.ndarray) -> None:
    """dst[:] = src for flat uint8 arrays; large copies are split over four threads (numpy releases the GIL, and the
    first touch of a fresh destination -- the kernel zeroing its pages -- is most of the cost of a 33 MB copy)."""
    n = dst.shape[0]
    if n < (8 << 20):
        dst[:] = src
        return
    step = (n // 4 + 4095) & ~4095
    futs = [_pool4().submit(np.copyto, dst[o:o + step], src[o:o + step]) for o in range(0, n, step)]
    for f in futs:
        f.result()


def host_extrema(a: np.ndarray):
    """(min, max) of a C-contiguous integer array through ONE foreign call (``amt_host_minmax_int``), or None when the
    element type has no such path (floats, uint64, strided views): callers then use numpy."""
    if a.dtype.kind not in "iub" or a.dtype == np.uint64 or not a.flags["C_CONTIGUOUS"] or a.size == 0:
        return None
    out = np.empty(2, np.int64)
    _hip.check(_hip.load_library().amt_host_minmax_int(a.ctypes.data, a.dtype.itemsize, int(a.dtype.kind == "i"),
                                                       a.size, out.ctypes.data), "amt_host_minmax_int")
    return int(out[0]), int(out[1])


def _stage_chunk(dst: np.ndarray, src: np.ndarray, stats: bool):
    """One chunk of a staged upload: convert / copy into the page-locked buffer and, on request, the chunk's extrema
    (the chunk is still in cache: the constructor checks of a label image cost no second pass over it).  The common
    cases are one foreign call each -- a worker thread that makes many short numpy calls queues for the interpreter
    lock between them when other worker threads do the same."""
    lib = _hip.load_library()
    if dst.dtype == np.int32 and src.dtype == np.int64 and dst.nbytes >= (1 << 16):
        mm = np.empty(2, np.int64) if stats else None
        _hip.check(lib.amt_host_narrow_i64_i32(dst.ctypes.data, src.ctypes.data, src.size,
                                               None if mm is None else mm.ctypes.data), "amt_host_narrow_i64_i32")
        return (mm[0], mm[1]) if stats else None
    if dst.dtype == src.dtype and dst.nbytes >= (1 << 16):
        # streaming stores (amt_host_copy): the staging block is read next by the DMA engine, not by this core
        _hip.check(lib.amt_host_copy(dst.ctypes.data, src.ctypes.data, dst.nbytes), "amt_host_copy")
    else:
        np.copyto(dst, src, casting="unsafe")
    if not stats:
        return None
    return host_extrema(src) or (src.min(), src.max())


def _chunks(n: int):
    step = ((-(-n // _PIPE_CHUNKS)) + 4095) & ~4095
    return [(o, min(step, n - o)) for o in range(0, n, step)]


class _PinnedBlock:
    """Page-locked host memory behind a returned numpy array (its ``base``): when the last view dies the block goes
    back to the pool, already faulted in and registered for DMA."""

    __slots__ = ("ptr", "nbytes", "pool", "__array_interface__", "__weakref__")

    def __init__(self, ptr: int, nbytes: int, pool: "_ResultPool"):
        self.ptr, self.nbytes, self.pool = ptr, nbytes, pool
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            self.pool._put(self.ptr, self.nbytes)
        except Exception:
            pass


class _ResultPool:
    """Large results (label images, filtered planes: 17-34 MB at 2048^2) are copied by the DMA engine straight into
    page-locked blocks that become the returned arrays: a fresh pageable array costs 8,192 page faults (2-3 ms,
    more than the 0.6 ms the bus takes) before the copy out of a staging buffer can even land.  At most
    AMT_RESULT_PINNED_BYTES (default 1 GiB; 0 = off) are handed out at any time -- beyond that, and for anything
    below 4 MB, results are ordinary arrays -- and returned blocks are kept for the next call (up to half the budget,
    at least 256 MiB)."""

    def __init__(self):
        # re-entrant: a block's __del__ (-> _put) can run from the garbage collector at any allocation, including
        # one made while this thread holds the lock in get()
        self.lock = threading.RLock()
        self.free: dict[int, list[int]] = {}
        self.kept = 0
        self.out = 0
        self.cap_out = int(os.environ.get("AMT_RESULT_PINNED_BYTES", str(1 << 30)))
        # returned blocks kept for the next call: releasing one (hipHostFree) synchronises the device and allocating
        # the next (hipHostMalloc) costs milliseconds, so a process that cycles through many label images keeps half
        # its budget instead of the 256 MiB that serve one thread
        self.cap_keep = max(256 << 20, self.cap_out // 2)

    def get(self, nbytes: int):
        size = (nbytes + (1 << 20) - 1) & ~((1 << 20) - 1)
        with self.lock:
            lst = self.free.get(size)
            if lst:
                ptr = lst.pop()
                self.kept -= size
                self.out += size
                return _PinnedBlock(ptr, size, self)
            if self.out + size > self.cap_out:
                return None
            self.out += size
        p = ctypes.c_void_p()
        if _hip.load_library().amt_host_alloc(size, ctypes.byref(p)) != 0 or not p.value:
            with self.lock:
                self.out -= size
            return None
        return _PinnedBlock(p.value, size, self)

    def _put(self, ptr: int, size: int):
        with self.lock:
            self.out -= size
            if self.kept + size <= self.cap_keep:
                self.free.setdefault(size, []).append(ptr)
                self.kept += size
                return
        _hip.load_library().amt_host_free(ctypes.c_void_p(ptr))


_result_pool = _ResultPool()


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt not in _DTYPE_CODE:
        raise TypeError(f"dtype {dt} is not supported on the device path")
    return _DTYPE_CODE[dt]


class Context:
    """One GPU + one HIP stream + one scratch arena."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._lib = _hip.load_library()
        h = ctypes.c_void_p()
        if stream is None:
            _hip.check(self._lib.amt_ctx_create(int(device), ctypes.byref(h)), "amt_ctx_create")
        else:
            _hip.check(self._lib.amt_ctx_create_on_stream(int(device), ctypes.c_void_p(stream), ctypes.byref(h)),
                       "amt_ctx_create_on_stream")
        self.handle = h
        self.device = int(device)
        # Freed device buffers are kept for the next allocation of the same size: hipFree synchronises the device and
        # costs ~0.15 ms a call, and the reference-level API (one call = a dozen temporaries) would pay it twenty times
        # per image.  Reuse is safe because every buffer of a context is only ever touched by that context's stream
        # (arrays handed to another context with DeviceArray.on() are excluded and freed the slow way).
        self._pool: dict[int, list[int]] = {}
        self._pool_bytes = 0
        self._pool_cap = int(os.environ.get("AMT_POOL_BYTES", str(4 << 30)))
        self._pool_lock = threading.RLock()  # re-entrant: DeviceArray.__del__ -> _release may run under empty()
        # Host <-> device copies of pageable numpy arrays go through one page-locked staging buffer per context: the
        # runtime otherwise registers every fresh host array for DMA, which costs 1.6 ms or 25 ms per 33 MB array
        # (alternating, measured) against 0.6 ms of PCIe time; a host memcpy through pinned memory is 3 ms and steady
        self._stage_buf = None
        self._stage_cap = int(os.environ.get("AMT_STAGE_BYTES", str(256 << 20)))
        self._chunk_events: list[ctypes.c_void_p] = []

    def _staging(self, nbytes: int):
        """A page-locked uint8 array of at least ``nbytes`` (None if the copy is too small to matter or too large)."""
        if nbytes < (64 << 10) or nbytes > self._stage_cap:
            return None
        if self._stage_buf is None or self._stage_buf.array.nbytes < nbytes:
            if self._stage_buf is not None:
                self.synchronize()
                self._stage_buf.close()
            self._stage_buf = PinnedBuffer((max(nbytes, 64 << 20),), np.uint8)
        return self._stage_buf.array

    def _events_for_chunks(self, n: int):
        while len(self._chunk_events) < n:
            h = ctypes.c_void_p()
            _hip.check(self._lib.amt_event_create(self.handle, ctypes.byref(h)), "amt_event_create")
            self._chunk_events.append(h)
        return self._chunk_events[:n]

    def set_fork(self, enable) -> None:
        """Let independent kernels inside one call use the context's auxiliary streams (True = all three, the default;
        an int = that many) or keep everything on its one stream (False / 0) -- the right choice when several
        contexts already run side by side (amt_ctx_set_fork)."""
        n = 3 if enable is True else int(enable)
        _hip.check(self._lib.amt_ctx_set_fork(self.handle, n), "amt_ctx_set_fork")

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None):
            self.trim()
            if self._stage_buf is not None:
                self.synchronize()
                self._stage_buf.close()
                self._stage_buf = None
            for ev in self._chunk_events:
                self._lib.amt_event_destroy(self.handle, ev)
            self._chunk_events = []
            self._lib.amt_ctx_destroy(self.handle)
            self.handle = None

    def trim(self) -> None:
        """Return every cached buffer to the driver."""
        with self._pool_lock:
            ptrs = [p for lst in self._pool.values() for p in lst]
            self._pool.clear()
            self._pool_bytes = 0
        for p in ptrs:
            self._lib.amt_free(self.handle, p)

    def _release(self, ptr: int, nbytes: int, shared: bool) -> None:
        if not shared and nbytes:
            with self._pool_lock:
                if self._pool_bytes + nbytes <= self._pool_cap:
                    self._pool.setdefault(nbytes, []).append(ptr)
                    self._pool_bytes += nbytes
                    return
        self._lib.amt_free(self.handle, ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ---------------------------------------------------------------------------------
    def empty(self, shape, dtype) -> "DeviceArray":
        shape = tuple(int(s) for s in shape)
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        nalloc = max(256, (nbytes + 255) & ~255)
        with self._pool_lock:
            lst = self._pool.get(nalloc)
            ptr = lst.pop() if lst else 0
            if ptr:
                self._pool_bytes -= nalloc
        if not ptr:
            p = ctypes.c_void_p()
            _hip.check(self._lib.amt_malloc(self.handle, nalloc, ctypes.byref(p)), "amt_malloc")
            ptr = p.value
        a = DeviceArray(self, ptr, shape, dt, owner=True)
        a._alloc = nalloc
        return a

    def zeros(self, shape, dtype) -> "DeviceArray":
        a = self.empty(shape, dtype)
        _hip.check(self._lib.amt_memset(self.handle, a.ptr, 0, a.nbytes), "amt_memset")
        return a

    def asarray(self, arr: np.ndarray, dtype=None, stats: bool = False, out: "DeviceArray | None" = None):
        """Host -> device copy (bool becomes uint8 0/1).  ``dtype``: the element type on the device, converted while
        the data moves into the page-locked staging buffer (int64 label images travel as int32).  ``stats=True``
        returns ``(array, (min, max))`` of the HOST values, computed chunk by chunk in the same pass.  Transfers of
        4 MB and more are cut into chunks: four host threads fill the staging buffer while the DMA engine drains
        the chunks already there.  ``out``: an existing device array (or a slot of one) to fill instead of a new one."""
        a = np.ascontiguousarray(arr)
        if a.dtype == np.bool_:
            a = a.view(np.uint8)
        if out is not None:
            dt = out.dtype
            if out.size != a.size:
                raise ValueError("asarray(out=...): the destination has a different number of elements")
            d = out
        else:
            dt = np.dtype(dtype) if dtype is not None else a.dtype
            d = self.empty(a.shape, dt)
        mm = None
        if a.size:
            flat = a.reshape(-1)
            isz = dt.itemsize
            nbytes = flat.shape[0] * isz
            stage = self._staging(nbytes)
            if stage is None or nbytes < _PIPE_MIN:
                if stats:
                    mm = (flat.min(), flat.max())
                src_arr = flat if flat.dtype == dt else flat.astype(dt)
                if stage is not None:
                    self.synchronize()  # the staging buffer may still feed an earlier copy
                    stage[:nbytes] = src_arr.view(np.uint8)
                    src = stage.ctypes.data
                else:
                    src = src_arr.ctypes.data
                _hip.check(self._lib.amt_memcpy_h2d(self.handle, d.ptr, src, nbytes), "amt_memcpy_h2d")
                self.synchronize()  # the host buffer may be a temporary
            else:
                self.synchronize()
                st = stage[:nbytes].view(dt)
                parts = _chunks(flat.shape[0])
                pool = _pool4()
                futs = [pool.submit(_stage_chunk, st[o:o + m], flat[o:o + m], stats) for o, m in parts]
                base = stage.ctypes.data
                for (o, m), f in zip(parts, futs):
                    r = f.result()
                    if r is not None:
                        mm = r if mm is None else (min(mm[0], r[0]), max(mm[1], r[1]))
                    _hip.check(self._lib.amt_memcpy_h2d(self.handle, d.ptr + o * isz, base + o * isz, m * isz),
                               "amt_memcpy_h2d")
                self.synchronize()
        if a.dtype == np.uint8 and getattr(arr, "dtype", None) == np.bool_ and out is None:
            d.is_bool = True
        return (d, mm) if stats else d

    def synchronize(self):
        _hip.check(self._lib.amt_sync(self.handle), "amt_sync")

    def copy_from_host_async(self, dst: "DeviceArray", src: np.ndarray):
        """Enqueue a host -> device copy on this context's stream WITHOUT waiting for it.  ``src`` must stay
        alive and unchanged until the stream has passed the copy; it only overlaps with other streams when it
        is page-locked (``pinned_empty``)."""
        if not src.flags["C_CONTIGUOUS"] or src.nbytes != dst.nbytes:
            raise ValueError("copy_from_host_async needs a C-contiguous source of the destination's size")
        _hip.check(self._lib.amt_memcpy_h2d(self.handle, dst.ptr, src.ctypes.data, src.nbytes), "amt_memcpy_h2d")

    def copy_to_host_async(self, dst: np.ndarray, src: "DeviceArray", nbytes: int | None = None):
        """Enqueue a device -> host copy of the first ``nbytes`` of ``src`` on this context's stream WITHOUT waiting
        for it (``dst``: page-locked, C-contiguous, large enough; read it after an event recorded behind the copy)."""
        nbytes = src.nbytes if nbytes is None else int(nbytes)
        if not dst.flags["C_CONTIGUOUS"] or nbytes > dst.nbytes or nbytes > src.nbytes:
            raise ValueError("copy_to_host_async needs a C-contiguous destination of at least nbytes")
        if nbytes:
            _hip.check(self._lib.amt_memcpy_d2h(self.handle, dst.ctypes.data, src.ptr, nbytes), "amt_memcpy_d2h")

    @property
    def stream_ptr(self) -> int:
        """The context's hipStream_t as an integer (0 = the null stream)."""
        p = ctypes.c_void_p()
        _hip.check(self._lib.amt_ctx_stream(self.handle, ctypes.byref(p)), "amt_ctx_stream")
        return int(p.value or 0)

    def wait_for(self, other: "Context"):
        """Order this context's stream after the work enqueued so far on ``other``'s stream (no host sync)."""
        _hip.check(self._lib.amt_stream_wait(self.handle, other.handle), "amt_stream_wait")

    def event(self) -> "Event":
        """A HIP event bound to this context's device (record on any context of the device, wait on any other)."""
        return Event(self)

    def device_name(self) -> str:
        buf = ctypes.create_string_buffer(256)
        _hip.check(self._lib.amt_device_name(self.handle, buf, 256), "amt_device_name")
        return buf.value.decode()

    # -- timing (HIP events on this context's stream) ---------------------------------------------
    def timer(self) -> "Timer":
        return Timer(self)


class Event:
    """Ordering point between contexts: ``ev.record(a)`` marks a's stream, ``ev.wait(b)`` orders b's next work
    after that mark (and not after what a's stream was given later, unlike ``Context.wait_for``)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        h = ctypes.c_void_p()
        _hip.check(ctx._lib.amt_event_create(ctx.handle, ctypes.byref(h)), "amt_event_create")
        self.h = h

    def record(self, ctx: Context | None = None):
        c = ctx or self.ctx
        _hip.check(c._lib.amt_event_record(c.handle, self.h), "amt_event_record")

    def wait(self, ctx: Context):
        _hip.check(ctx._lib.amt_event_wait(ctx.handle, self.h), "amt_event_wait")

    def synchronize(self):
        """The HOST waits for the last record of this event."""
        _hip.check(self.ctx._lib.amt_event_sync(self.ctx.handle, self.h), "amt_event_sync")

    def __del__(self):
        try:
            if self.ctx.handle:
                self.ctx._lib.amt_event_destroy(self.ctx.handle, self.h)
        except Exception:
            pass


class Timer:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        h = ctypes.c_void_p()
        _hip.check(ctx._lib.amt_timer_create(ctx.handle, ctypes.byref(h)), "amt_timer_create")
        self.h = h

    def start(self):
        _hip.check(self.ctx._lib.amt_timer_start(self.ctx.handle, self.h), "amt_timer_start")

    def stop(self):
        _hip.check(self.ctx._lib.amt_timer_stop(self.ctx.handle, self.h), "amt_timer_stop")

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        _hip.check(self.ctx._lib.amt_timer_elapsed_ms(self.ctx.handle, self.h, ctypes.byref(ms)), "amt_timer_elapsed")
        return float(ms.value)

    def __del__(self):
        try:
            if self.ctx.handle:
                self.ctx._lib.amt_timer_destroy(self.ctx.handle, self.h)
        except Exception:
            pass


class DeviceArray:
    """A typed view of device memory; frees the allocation when the owning object dies."""

    __slots__ = ("ctx", "ptr", "shape", "dtype", "_owner", "_base", "is_bool", "_alloc", "_shared")

    def __init__(self, ctx: Context, ptr: int, shape, dtype, owner: bool = False, base=None):
        self.ctx = ctx
        self.ptr = int(ptr) if ptr else 0
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner
        self._base = base  # keeps the owning array alive for views
        self.is_bool = False
        self._alloc = 0       # bytes of the allocation (owners made by Context.empty)
        self._shared = False  # handed to another context: not eligible for the context's buffer cache

    # -- geometry -------------------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def nplanes(self):
        return int(np.prod(self.shape[:-2], dtype=np.int64)) if self.ndim >= 2 else 1

    @property
    def plane_shape(self):
        if self.ndim < 2:
            raise ValueError("device ops need at least 2-D arrays")
        return self.shape[-2:]

    def __getitem__(self, idx) -> "DeviceArray":
        """Integer / contiguous-slice indexing of the LEADING axis only (a pointer offset, no copy)."""
        if self.ndim < 1:
            raise IndexError("0-d DeviceArray")
        inner = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        if isinstance(idx, (int, np.integer)):
            i = int(idx)
            if i < 0:
                i += self.shape[0]
            if not 0 <= i < self.shape[0]:
                raise IndexError(idx)
            v = DeviceArray(self.ctx, self.ptr + i * inner, self.shape[1:], self.dtype, base=self._base if self._base is not None else self)
        elif isinstance(idx, slice):
            start, stop, step = idx.indices(self.shape[0])
            if step != 1:
                raise IndexError("only contiguous slices of the leading axis are device views")
            n = max(0, stop - start)
            v = DeviceArray(self.ctx, self.ptr + start * inner, (n,) + self.shape[1:], self.dtype,
                            base=self._base if self._base is not None else self)
        else:
            raise IndexError("DeviceArray supports int or slice indexing of the leading axis only")
        v.is_bool = self.is_bool
        return v

    def on(self, ctx: Context) -> "DeviceArray":
        """The same memory as an array of another context of the same device.  Operators run on the stream of the
        context their input belongs to, so a batch part that was uploaded by one context and is processed by
        another (bench.py's stream split, the feeder's device buffers) must be re-bound, and ordered against its
        producer with ``Context.wait_for`` / an ``Event`` if that producer may still be running."""
        if ctx is self.ctx:
            return self
        if ctx.device != self.ctx.device:
            raise ValueError(f"cannot bind an array of device {self.ctx.device} to a context of device {ctx.device}")
        owner = self._base if self._base is not None else self
        owner._shared = True
        v = DeviceArray(ctx, self.ptr, self.shape, self.dtype, base=owner)
        v.is_bool = self.is_bool
        return v

    def reshape(self, *shape) -> "DeviceArray":
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        if int(np.prod(shape, dtype=np.int64)) != self.size:
            raise ValueError("cannot reshape DeviceArray: size mismatch")
        v = DeviceArray(self.ctx, self.ptr, shape, self.dtype, base=self._base if self._base is not None else self)
        v.is_bool = self.is_bool
        return v

    # -- transfers ------------------------------------------------------------------------------
    def numpy(self, dtype=None) -> np.ndarray:
        """Device -> host copy (synchronises the stream). uint8 masks flagged as bool come back as bool.  ``dtype``:
        the element type of the returned array.  Results of 4 MB and more land directly in a page-locked block that
        backs the returned array (_ResultPool; int32 -> int64 is widened on the device first).  When the pool's
        budget is spent the copy arrives in chunks in the staging buffer and four host threads move / convert the
        finished chunks into an ordinary array while the DMA engine delivers the next ones."""
        odt = np.dtype(dtype) if dtype is not None else self.dtype
        ctx = self.ctx
        lib = ctx._lib
        nbytes = self.nbytes
        if self.size * odt.itemsize >= _PIPE_MIN and (odt == self.dtype or (self.dtype == np.int32 and odt == np.int64)):
            block = _result_pool.get(self.size * odt.itemsize)
            if block is not None:
                src = self
                if odt != self.dtype:  # widened on the device: the DMA engine writes the final array
                    src = ctx.empty(self.shape, odt)
                    _hip.check(lib.amt_cast_i32_i64(ctx.handle, self.ptr, src.ptr, self.size), "amt_cast_i32_i64")
                _hip.check(lib.amt_memcpy_d2h(ctx.handle, block.ptr, src.ptr, src.nbytes), "amt_memcpy_d2h")
                ctx.synchronize()
                out = np.asarray(block)[: src.nbytes].view(odt).reshape(self.shape)
                if self.is_bool and self.dtype == np.uint8 and dtype is None:
                    return out.view(np.bool_)
                return out
        out = np.empty(self.shape, dtype=odt)
        if out.size:
            isz = self.dtype.itemsize
            stage = ctx._staging(nbytes)
            if stage is None or nbytes < _PIPE_MIN:
                raw = out if odt == self.dtype else np.empty(self.shape, dtype=self.dtype)
                dst = stage.ctypes.data if stage is not None else raw.ctypes.data
                _hip.check(lib.amt_memcpy_d2h(ctx.handle, dst, self.ptr, nbytes), "amt_memcpy_d2h")
                ctx.synchronize()
                if stage is not None:
                    raw.reshape(-1).view(np.uint8)[:] = stage[:nbytes]
                if raw is not out:
                    np.copyto(out, raw, casting="unsafe")
            else:
                st = stage[:nbytes].view(self.dtype)
                flat = out.reshape(-1)
                parts = _chunks(flat.shape[0])
                evs = ctx._events_for_chunks(len(parts))
                base = stage.ctypes.data
                for (o, m), ev in zip(parts, evs):
                    _hip.check(lib.amt_memcpy_d2h(ctx.handle, base + o * isz, self.ptr + o * isz, m * isz),
                               "amt_memcpy_d2h")
                    _hip.check(lib.amt_event_record(ctx.handle, ev), "amt_event_record")
                pool = _pool4()
                futs = []
                for (o, m), ev in zip(parts, evs):
                    _hip.check(lib.amt_event_sync(ctx.handle, ev), "amt_event_sync")
                    futs.append(pool.submit(np.copyto, flat[o:o + m], st[o:o + m], "unsafe"))
                for f in futs:
                    f.result()
        if self.is_bool and self.dtype == np.uint8 and dtype is None:
            return out.view(np.bool_)
        return out

    def numpy_int64(self) -> np.ndarray:
        """int32 labels as the int64 array the reference's API returns (R/model.py:215, R/masks.py:63-65)."""
        return self.numpy(dtype=np.int64)

    def copy(self) -> "DeviceArray":
        d = self.ctx.empty(self.shape, self.dtype)
        _hip.check(self.ctx._lib.amt_memcpy_d2d(self.ctx.handle, d.ptr, self.ptr, self.nbytes), "amt_memcpy_d2d")
        d.is_bool = self.is_bool
        return d

    def __del__(self):
        try:
            if self._owner and self.ptr and self.ctx.handle:
                # back to the context's cache (same-stream reuse is ordered); shared or uncached buffers go through
                # hipFree, whose implicit synchronisation orders them against every stream
                self.ctx._release(self.ptr, self._alloc, self._shared)
                self.ptr = 0
        except Exception:
            pass

    def __repr__(self):
        return f"<DeviceArray shape={self.shape} dtype={self.dtype} device={self.ctx.device}>"


class PinnedBuffer:
    """Page-locked host memory (hipHostMalloc) exposed as a numpy array: the source of asynchronous H2D copies."""

    def __init__(self, shape, dtype):
        self._lib = _hip.load_library()
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = ctypes.c_void_p()
        _hip.check(self._lib.amt_host_alloc(nbytes, ctypes.byref(p)), "amt_host_alloc")
        self._ptr = p
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape, dtype=np.int64))).reshape(
            self.shape)

    def close(self):
        if getattr(self, "_ptr", None):
            self.array = None
            self._lib.amt_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pinned_empty(shape, dtype) -> PinnedBuffer:
    return PinnedBuffer(shape, dtype)


_tls = threading.local()
_default_device = 0


def set_default_device(device: int):
    """Select the GPU used by contexts created after this call (one process per GPU: LOCAL_RANK)."""
    global _default_device
    _default_device = int(device)
    _tls.__dict__.pop("ctx", None)


def get_context() -> Context:
    """The calling thread's context (created on first use).  Raises HipUnavailableError without a GPU."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None or ctx.handle is None or ctx.device != _default_device:
        ctx = Context(_default_device)
        if threading.current_thread() is not threading.main_thread():
            # a worker thread's context runs beside its siblings' (Pipeline(parallel=True), R/pipeline.py:145-146): the
            # threads provide the overlap, every call stays on its context's one stream (amt_ctx_set_fork)
            ctx.set_fork(False)
        _tls.ctx = ctx
    return ctx


def is_device_array(x) -> bool:
    return isinstance(x, DeviceArray)

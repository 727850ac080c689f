"""Host images -> ``SegmentationMask`` objects in ONE pass over the bus (the throughput route behind the reference-level
API: R/model.py:217-290 followed by R/masks.py:118-328).

The reference's two calls hand an int64 label image from ``batch_segment`` back into ``SegmentationMask``; on a device
that is 33.5 MB down and 50 MB up again per 2048^2 field of view, plus the validation and narrowing passes over it on
the host.  ``SegmentationModel.batch_masks`` keeps what the first call computed where the second one needs it: every
image crosses the bus once (all channels, page-locked double buffer, the copy of chunk i + 1 overlaps the kernels of
chunk i), the chain of ``FovSegmenter`` leaves labels + feature tables in HBM, one pack kernel compacts the rows that
exist and only those come home.  The masks it returns are ordinary ``SegmentationMask`` objects whose label plane lives
on the device and whose ``mask_image`` / ``label_image`` are downloaded when somebody asks for them."""
from __future__ import annotations

import numpy as np

from . import _hip, hipops
from .device import Context, _pool4


def _copy_piece(dst: np.ndarray, src: np.ndarray) -> None:
    if dst.dtype == src.dtype:  # streaming stores: the block is read next by the DMA engine, not by this core
        _hip.check(_hip.load_library().amt_host_copy(dst.ctypes.data, src.ctypes.data, dst.nbytes), "amt_host_copy")
    else:
        np.copyto(dst, src, casting="unsafe")


def _fill(dst: np.ndarray, src: np.ndarray) -> list:
    """dst[...] = src on the host-copy threads (page-locked destination; the GIL is released).  Returns futures."""
    flat_d, flat_s = dst.reshape(-1), np.ascontiguousarray(src).reshape(-1)
    n = flat_d.shape[0]
    step = max(1 << 20, -(-n // 4))
    return [_pool4().submit(_copy_piece, flat_d[o:o + step], flat_s[o:o + step]) for o in range(0, n, step)]


class MaskBatcher:
    """Chunks of ``B`` images of shape (C, H, W) uint16 -> per image (label plane on the device, cell count, morphology
    rows, intensity rows).  One per thread and context; buffers are reused from call to call."""

    times: dict | None = None  # set to {} to collect where a call's host time goes (tools/masks_profile.py)

    def __init__(self, B: int, C: int, H: int, W: int, *, nuclear: int, sigma: float, radius: int, min_distance: int,
                 ctx: Context):
        from .feeder import FovFeeder
        from .device import pinned_empty
        from .plate import packed_ncols
        from .segment import FovSegmenter

        self.B, self.C, self.H, self.W, self.ctx = B, C, H, W, ctx
        self.K = max(4096, (H * W) // 256)
        self.seg = FovSegmenter(B, C, H, W, sigma=sigma, radius=radius, min_distance=min_distance, max_cells=self.K,
                                dapi_index=nuclear, ctx=ctx, props=True, fused=True)
        self.feeder = FovFeeder((B, C, H, W), ctx.device)
        for slot in range(2):
            self.feeder.host(slot)[...] = 0  # a short chunk leaves the rest of its block as it was: blank at first
        self.ncols = packed_ncols(C)
        self.rows = [ctx.empty((B * self.K, self.ncols), np.float64) for _ in range(2)]
        self.nrows = [ctx.zeros((1,), np.int64) for _ in range(2)]
        # per slot, page-locked: row count, cell counts, marker counts
        self.small = [(pinned_empty((1,), np.int64), pinned_empty((B,), np.int32), pinned_empty((B,), np.int32))
                      for _ in range(2)]
        self.done = [ctx.event() for _ in range(2)]
        self.copied = [self.feeder.copy_ctx.event() for _ in range(2)]
        self._copy_pending = [False, False]

    # -- one chunk ------------------------------------------------------------------------------------------
    def _send(self, slot: int, part) -> None:
        """Chunk ``part`` (<= B images) into the page-locked block of ``slot`` and onto the copy stream."""
        f = self.feeder
        if self._copy_pending[slot]:
            self.copied[slot].synchronize()  # the DMA engine has read this page-locked block
        host = f.host(slot)
        futs = []
        for j, image in enumerate(part):
            futs += _fill(host[j], image)
        for fu in futs:
            fu.result()
        f.submit(slot)
        self.copied[slot].record(f.copy_ctx)
        self._copy_pending[slot] = True

    def _compute(self, slot: int):
        """Segment + measure + pack the chunk in ``slot``; returns the chunk's label block."""
        f, seg, ctx = self.feeder, self.seg, self.ctx
        d = f.acquire(slot, [ctx])
        labels = ctx.empty((self.B, self.H, self.W), np.int32)  # owned by the masks of this chunk
        seg.labels = labels
        seg.run_c3(d)
        f.release(slot, [ctx])
        hipops.pack_plate_rows(seg.table, seg.itable, seg.ncells, out=self.rows[slot], nrows_out=self.nrows[slot])
        for pin, dev in zip(self.small[slot], (self.nrows[slot], seg.ncells, seg.nmarkers)):
            ctx.copy_to_host_async(pin.array, dev)
        self.done[slot].record(ctx)
        return labels

    def _collect(self, slot: int, labels, n_images: int):
        """-> [(label plane, k, morph rows, intensity rows) | None] for the chunk's first ``n_images`` images; None
        for the whole chunk when a table or the marker list overflowed (the caller takes the general route)."""
        self.done[slot].synchronize()
        n = int(self.small[slot][0].array[0])
        ncells, nmark = self.small[slot][1].array.copy(), self.small[slot][2].array
        if n < 0 or (nmark < 0).any() or (nmark > self.K).any():
            return None
        rows = self.rows[slot][:max(n, 1)].numpy()[:n] if n else np.zeros((0, self.ncols))
        out, o = [], 0
        r0 = 2 + _hip.RP_NCOLS
        for j in range(n_images):
            k = int(ncells[j])
            blk = rows[o:o + k]
            out.append((labels[j], k, blk[:, 2:r0], blk[:, r0:].reshape(k, self.C, 4)))
            o += k
        return out

    def run(self, images) -> list | None:
        """All images, chunk by chunk: while chunk i - 1 is segmented the host fills and sends chunk i, then takes
        the rows of chunk i - 1 (its kernels finished during the fill) and enqueues the kernels of chunk i."""
        B = self.B
        chunks = [images[i:i + B] for i in range(0, len(images), B)]
        out: list = []
        prev = None
        clock = self._clock
        for i, part in enumerate(chunks):
            t = clock()
            self._send(i % 2, part)
            t = clock("send", t)
            if prev is not None:
                got = self._collect(*prev)
                if got is None:
                    self.ctx.synchronize()
                    self.feeder.copy_ctx.synchronize()
                    return None
                out += got
            t = clock("collect", t)
            prev = (i % 2, self._compute(i % 2), len(part))
            clock("enqueue", t)
        t = clock()
        got = self._collect(*prev) if prev is not None else []
        clock("collect_last", t)
        if got is None:
            return None
        return out + got

    def _clock(self, name=None, t0=0.0):
        if self.times is None:
            return 0.0
        import time

        now = time.perf_counter()
        if name is not None:
            self.times[name] = self.times.get(name, 0.0) + now - t0
        return now

    def close(self):
        self.ctx.synchronize()
        self.feeder.close()
        for trio in self.small:
            for p in trio:
                p.close()

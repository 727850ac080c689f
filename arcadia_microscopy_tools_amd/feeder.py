"""Host -> HBM feeder for fields of view (second half): double-buffered,
page-locked staging so that the PCIe transfer of the next batch overlaps the segmentation of the current one.

The reference reads a file into a numpy array and hands it to the pipeline (R/nikon.py:41 ``nd2f.asarray()``,
R/microscopy.py:284-308); on the device path the same hand-over is an asynchronous H2D copy on a copy stream,
ordered against the compute streams with ``amt_stream_wait`` (no host synchronisation in steady state)."""
from __future__ import annotations

import numpy as np

from .device import Context, DeviceArray, pinned_empty


class FovFeeder:
    """Two device buffers of ``shape`` uint16 fed from two pinned host buffers by a dedicated copy context.

    Protocol per batch ``i`` (slot = i % 2)::

        host = feeder.host(slot)            # fill it (file read, decode, ...)
        feeder.submit(slot)                 # async H2D on the copy stream
        d = feeder.acquire(slot, consumers) # consumers' streams wait for that copy; returns the DeviceArray
        ... enqueue the segmentation of d on the consumers ...
        feeder.release(slot, consumers)     # the copy stream will not overwrite d before they are done
    """

    def __init__(self, shape, device: int = 0):
        self.copy_ctx = Context(device)
        self.shape = tuple(int(s) for s in shape)
        self._pinned = [pinned_empty(self.shape, np.uint16) for _ in range(2)]
        self._dev = [self.copy_ctx.empty(self.shape, np.uint16) for _ in range(2)]

    def host(self, slot: int) -> np.ndarray:
        return self._pinned[slot].array

    def device(self, slot: int) -> DeviceArray:
        return self._dev[slot]

    def submit(self, slot: int):
        self.copy_ctx.copy_from_host_async(self._dev[slot], self._pinned[slot].array)

    def acquire(self, slot: int, consumers) -> DeviceArray:
        """The consumers' streams wait for the copy into ``slot``.  The returned array belongs to the COPY context;
        operators run on the stream of their input's context, so bind it (or its parts) to the consuming context
        with ``DeviceArray.on`` before use -- ``FovSegmenter`` does that itself."""
        for c in consumers:
            c.wait_for(self.copy_ctx)
        return self._dev[slot]

    def release(self, slot: int, consumers):
        for c in consumers:
            self.copy_ctx.wait_for(c)

    def close(self):
        self.copy_ctx.synchronize()
        for p in self._pinned:
            p.close()

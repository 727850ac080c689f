"""Does the host link carry both directions at once?  H2D alone, D2H alone, then both from two threads / contexts."""
import sys, threading, time
sys.path.insert(0, ".")
import numpy as np
from arcadia_microscopy_tools_amd.device import Context, pinned_empty

N = 64 << 20
REP = 30


def run(ctx, dev, pin, h2d):
    for _ in range(REP):
        if h2d:
            ctx.copy_from_host_async(dev, pin.array)
        else:
            ctx.copy_to_host_async(pin.array, dev)
    ctx.synchronize()


a, b = Context(0), Context(0)
da, db = a.empty((N,), np.uint8), b.empty((N,), np.uint8)
pa, pb = pinned_empty((N,), np.uint8), pinned_empty((N,), np.uint8)
pa.array[:] = 1
pb.array[:] = 2
run(a, da, pa, True); run(b, db, pb, False)
for name, jobs in (("H2D alone", [(a, da, pa, True)]), ("D2H alone", [(b, db, pb, False)]),
                   ("H2D + D2H together", [(a, da, pa, True), (b, db, pb, False)]),
                   ("H2D + H2D together", [(a, da, pa, True), (b, db, pb, True)]),
                   ("D2H + D2H together", [(a, da, pa, False), (b, db, pb, False)])):
    ts = [threading.Thread(target=run, args=j) for j in jobs]
    t0 = time.perf_counter()
    [t.start() for t in ts]; [t.join() for t in ts]
    el = time.perf_counter() - t0
    print(f"{name}: {len(jobs) * REP * N / el / 1e9:.1f} GB/s in total")

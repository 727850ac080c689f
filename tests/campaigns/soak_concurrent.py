"""Repeatability soak of the timed configuration: several contexts (HIP streams) run config 3 side by side on
resident FOVs, step after step; every step's labels and feature tables must equal step 0's bit for bit, and step 0's
must equal the CPU oracle for the distinct FOVs.  A data race between streams, a stale scratch plane or a marker
plane that was not cleared shows up here as a difference.  Usage: soak_concurrent.py [steps] [contexts] [fovs/context]."""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.device import Context
from arcadia_microscopy_tools_amd.segment import FovSegmenter
from oracle import chains

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per = int(sys.argv[3]) if len(sys.argv) > 3 else 12
S = int(os.environ.get("SOAK_SIZE", "1024"))
NU = 6
uniq = [synth.synth_fov(40 + i, size=S) for i in range(NU)]
bad = 0
for fork in (0, 3):
    ctxs = [Context(0) for _ in range(nctx)]
    for c in ctxs:
        c.set_fork(fork)
    # every context sees the distinct FOVs in a different order
    orders = [[(3 * k + j) % NU for j in range(per)] for k in range(nctx)]
    parts = [c.asarray(np.stack([uniq[i] for i in o])) for c, o in zip(ctxs, orders)]
    segs = [FovSegmenter(per, 4, S, S, ctx=c, max_cells=int(os.environ.get("SOAK_MAX_CELLS", "1024" if S <= 1024 else "2048"))) for c in ctxs]
    ref = None
    t0 = time.time()
    for it in range(steps):
        for sg, p in zip(segs, parts):
            sg.run_c3(p)
        for c in ctxs:
            c.synchronize()
        snap = []
        for sg in segs:
            nc = sg.ncells.numpy()
            snap.append((nc.copy(), zlib.crc32(sg.labels.numpy().tobytes()), sg.table.numpy().copy(),
                         sg.itable.numpy().copy()))
        if ref is None:
            ref = snap
            for sg in segs:
                sg.result()  # raises when a FOV overflowed max_cells or the sparse-labelling capacity
            # step 0 against the oracle, one FOV per distinct input
            lab0 = segs[0].labels.numpy()
            for j, i in enumerate(orders[0][:NU]):
                want, _ = chains.c3_labels(uniq[i][1])
                ok = np.array_equal(lab0[j], want)
                bad += not ok
                print(f"fork {fork} oracle fov {i}: labels {'ok' if ok else 'DIFFER'} ({int(want.max())} cells)")
            # the same FOV must give the same answer in every context and slot
            first = {}
            for k, sg in enumerate(segs):
                lab = sg.labels.numpy()
                for j, i in enumerate(orders[k]):
                    c = zlib.crc32(lab[j].tobytes())
                    if first.setdefault(i, c) != c:
                        bad += 1
                        print(f"fork {fork}: FOV {i} differs between slots (context {k}, slot {j})")
            continue
        for k, (a, b) in enumerate(zip(ref, snap)):
            same = np.array_equal(a[0], b[0]) and a[1] == b[1]
            for f in range(per):
                n = int(a[0][f])
                same = same and np.array_equal(a[2][f, :n], b[2][f, :n]) and np.array_equal(a[3][f, :n], b[3][f, :n])
            if not same:
                bad += 1
                print(f"fork {fork} step {it} context {k}: DIFFERS from step 0")
        if it % 10 == 9:
            print(f"fork {fork} step {it + 1}/{steps} ok so far, {time.time() - t0:.0f} s", flush=True)
    del segs, parts
    for c in ctxs:
        c.close()
print("BAD", bad)
sys.exit(1 if bad else 0)

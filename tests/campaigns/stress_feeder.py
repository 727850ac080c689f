"""Stress of the host -> HBM feeder hand-over: fresh context + segmenter + feeder per round, three batches streamed
while the previous one is segmented, every stage compared with the oracle's intermediates (computed once).
Found in round 2: operators run on the stream of their INPUT's context, so the feeder's buffers (owned by the copy
context) must be re-bound with DeviceArray.on(); FovSegmenter now does that itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.device import Context
from arcadia_microscopy_tools_amd.feeder import FovFeeder
from arcadia_microscopy_tools_amd.segment import FovSegmenter
from oracle import chains

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
use_feeder = (sys.argv[2] != "resident") if len(sys.argv) > 2 else True
batches = [np.stack([synth.synth_fov(10 * b + i, size=192) for i in range(2)]) for b in range(3)]
refs = [[chains.c3_labels(f[1]) for f in b] for b in batches]
bad = 0
t0 = time.time()
for r in range(rounds):
    ctx = Context(0)
    seg = FovSegmenter(2, 4, 192, 192, ctx=ctx, max_cells=128, fused=False)
    feeder = FovFeeder(batches[0].shape) if use_feeder else None
    if feeder:
        feeder.host(0)[...] = batches[0]
        feeder.submit(0)
    for i in range(3):
        slot = i % 2
        if feeder:
            d = feeder.acquire(slot, [ctx])
            if i + 1 < 3:
                feeder.host(1 - slot)[...] = batches[i + 1]
                feeder.submit(1 - slot)
        else:
            d = ctx.asarray(batches[i])
        lab = seg.run_c3(d)
        if feeder:
            feeder.release(slot, [ctx])
        L = lab.numpy().copy()
        for j in range(2):
            ref, inter = refs[i][j]
            if not np.array_equal(L[j].astype(np.int64), ref):
                bad += 1
                st = {
                    "input": np.array_equal(d.numpy()[j], batches[i][j]),
                    "mask": np.array_equal(seg.mask_a.numpy()[j].astype(bool), inter["mask"]),
                    "d2": np.array_equal(np.sqrt(seg.d2.numpy()[j].astype(np.float64)), inter["edt"]),
                    "markers": np.array_equal(seg.markers.numpy()[j], inter["markers"]),
                    "ws": np.array_equal(seg.ws.numpy()[j], inter["watershed"]),
                }
                print(f"round {r} batch {i} fov {j}: MISMATCH {st} ndiff {int((L[j] != ref).sum())} "
                      f"nmarkers {seg.nmarkers.numpy()} ncells {seg.ncells.numpy()} thr {seg.thr.numpy()}", flush=True)
    if feeder:
        feeder.close()
    if time.time() - t0 > 200:
        print("time budget reached at round", r); break
print(f"done: {bad} mismatches in {r + 1} rounds ({'feeder' if use_feeder else 'resident'})")

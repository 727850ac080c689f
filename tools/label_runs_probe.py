"""amt_label on 0 / 1 masks (run tables, AMT_CCL_RUNS=1, against the parent plane, =0): differential check against the
oracle on widths that take the path (multiples of 16), then the time per 48 planes of 2048^2.
usage: python tools/label_runs_probe.py [cases] [seed] [planes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import ndimage as ndi
from arcadia_microscopy_tools_amd import hipops, synth
from arcadia_microscopy_tools_amd.device import get_context
from oracle import skops

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 48
ctx = get_context()
bad = 0
for case in range(ncases):
    H, W = int(rng.integers(1, 420)), 16 * int(rng.integers(1, 34))
    kind = int(rng.integers(0, 5))
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == 0:
        m = rng.random((H, W)) < rng.uniform(0.3, 0.65)
    elif kind == 1:
        m = ndi.gaussian_filter(rng.random((H, W)), rng.uniform(1.5, 6.0)) > 0.5
    elif kind == 2:
        m = np.zeros((H, W), bool)
        for _ in range(int(rng.integers(2, 12))):
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(3, 90)
            d = np.hypot(yy - cy, xx - cx)
            m |= (d <= r) & (d >= r - rng.integers(1, 6)) if rng.random() < 0.5 else d <= r
        for _ in range(int(rng.integers(0, 4))):
            o = int(rng.integers(-W, W))
            m |= np.abs(yy - xx - o) <= rng.integers(0, 2)
            m |= np.abs(yy + xx - abs(o)) <= rng.integers(0, 2)
    elif kind == 3:
        m = ndi.gaussian_filter(rng.random((H, W)), 2.0) > 0.47
    else:  # checkerboards and stripes: the most runs a tile can hold, diagonal-only contacts
        p = int(rng.integers(1, 4))
        m = ((yy // p + xx // p) % 2 == 0) if rng.random() < 0.5 else (xx % 2 == 0) & (rng.random((H, W)) < 0.9)
    nb = int(rng.integers(1, 4))
    stack = np.stack([m] + [rng.random((H, W)) < 0.5 for _ in range(nb - 1)])
    # uint8 bytes take amt_label (other byte values would be noticed), bool arrays amt_label_mask (truth value)
    d = ctx.asarray(stack.astype(np.uint8)) if case % 2 else ctx.asarray(stack)
    for conn in (1, 2):
        lab, cnt = hipops.label(d, connectivity=conn)
        got, gc = lab.numpy(), cnt.numpy()
        for b in range(nb):
            ref = skops.label(stack[b], conn)
            if gc[b] != ref.max() or not np.array_equal(got[b], ref):
                bad += 1
                print("MISMATCH case", case, "kind", kind, (H, W), "conn", conn, "plane", b, "count", gc[b], ref.max(),
                      "px", int((got[b] != ref).sum()))
print("label fuzz:", ncases, "cases,", bad, "mismatches, runs =", os.environ.get("AMT_CCL_RUNS", "1"))

fov = synth.synth_fov(0)
for ch, name in ((1, "DAPI nuclei mask"), (0, "brightfield noise mask")):
  g = hipops.gaussian(ctx.asarray(np.stack([fov[ch]] * B)), 2.0)
  m = hipops.greater_than(g, hipops.threshold_otsu(g))
  lab, cnt = hipops.label(m)
  ref = skops.label(m.numpy()[0].astype(bool), 2)
  print(name, "2048^2 equal to the oracle:", np.array_equal(lab.numpy()[0], ref), "labels", int(cnt.numpy()[0]))
  for conn in (2, 1):
    hipops.label(m, connectivity=conn, out=lab, count=cnt); ctx.synchronize()
    t = ctx.timer(); t.start()
    for _ in range(10):
        hipops.label(m, connectivity=conn, out=lab, count=cnt)
    t.stop(); ms = t.elapsed_ms() / 10
    print(f"  label conn={conn}: {ms:.3f} ms per {B} planes = {5 * m.size / ms / 1e6:.0f} GB/s algorithmic = {5 * m.size / ms / 8e7:.1f} % of 8 TB/s")

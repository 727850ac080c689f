"""Cellpose-style flow -> mask post-processing (config 5) on random synthetic flow fields: sizes, cell counts, flow
noise (which breaks cells up and makes float32 trajectories sensitive), iteration counts, probability thresholds and
size filters, against the CPU restatement (oracle/cellpose_dynamics.py; parity with the cellpose package itself is
unpinned).  Usage: fuzz_dynamics.py [cases] [seed]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops, synth
from arcadia_microscopy_tools_amd.device import get_context
from oracle import cellpose_dynamics as cd

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = get_context()
bad = 0
for case in range(ncases):
    H, W = int(rng.integers(34, 260)), int(rng.integers(34, 300))
    ncells = int(rng.integers(0, max(1, H * W // 900)))
    noise = float(rng.choice([0.0, 0.0, 0.1, 0.3, 1.0]))
    niter = int(rng.choice([1, 10, 60, 200]))
    thr = float(rng.choice([0.0, 0.0, -7.0, 5.0, 7.0]))
    min_size = int(rng.choice([0, 15, 60]))
    frac = float(rng.choice([0.4, 0.4, 0.02, 1.0]))
    dP, prob, _ = synth.synthetic_flows((H, W), ncells, seed=int(rng.integers(0, 1 << 30)), noise=noise)
    if rng.random() < 0.2:
        prob = prob + rng.normal(0, 4, prob.shape).astype(np.float32)  # ragged cell-probability mask
    ref = cd.compute_masks(dP, prob, cellprob_threshold=thr, niter=niter, min_size=min_size, max_size_fraction=frac)
    lab, cnt = hipops.cellpose_masks(ctx.asarray(dP[None]), ctx.asarray(prob[None]), cellprob_threshold=thr, niter=niter,
                                     min_size=min_size, max_size_fraction=frac)
    got = lab.numpy()[0]
    ok = np.array_equal(got, ref) and int(cnt.numpy()[0]) == int(ref.max())
    if not ok:
        bad += 1
        print("MISMATCH", case, (H, W), ncells, noise, niter, thr, min_size, frac, int((got != ref).sum()),
              int(cnt.numpy()[0]), int(ref.max()), flush=True)
    if case % 20 == 19:
        print(f"{case + 1}/{ncases}, bad {bad}", flush=True)
print("BAD", bad)
sys.exit(1 if bad else 0)

"""What the N = 1 delivery of the feature table costs per step (plate48 arrangement: 4 contexts x 12 FOVs, or argv[1] FOVs):
A no delivery, B tables written into the staging rings, E the full delivery (per-context pack kernel + count + rows).

Round-3 findings with the FIRST design (one side stream ordered behind the four compute streams; 48 FOVs, ms per step):
no delivery 4.25 | the four stream waits alone 5.02 | events recorded on the compute streams, nobody waiting 4.27 |
pack kernel unordered on the side stream 4.26 | one pack kernel per context on its OWN stream 4.27 | waits + pack 5.46.
Cross-stream waits were the whole cost, so plate.HostTables keeps everything on the producing stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.device import Context, set_default_device
from arcadia_microscopy_tools_amd.plate import HostTables
from arcadia_microscopy_tools_amd.segment import FovSegmenter

set_default_device(0)
B, NS = int(sys.argv[1]) if len(sys.argv) > 1 else 48, 4
uniq = [synth.synth_fov(i) for i in range(4)]
fovs = np.stack([uniq[i % 4] for i in range(B)])
ctxs = [Context(0) for _ in range(NS)]
for c in ctxs:
    c.set_fork(0)
d = ctxs[0].asarray(fovs)
bounds = [round(i * B / NS) for i in range(NS + 1)]
parts = [d[bounds[i]:bounds[i + 1]] for i in range(NS)]
segs = [FovSegmenter(bounds[i + 1] - bounds[i], 4, 2048, 2048, ctx=ctxs[i], max_cells=2048) for i in range(NS)]
ht = HostTables(segs, slots=4, lag=int(os.environ.get("AMT_DELIVER_LAG", "2")))


def sync():
    for c in ctxs:
        c.synchronize()


def run(mode, steps=30):
    def step(i):
        if mode >= 1:
            ht.point(i)
        for sg, p in zip(segs, parts):
            sg.run_c3(p)
        if mode == 2:
            ht.deliver_step(i)
    for i in range(4):
        step(i)
    if mode == 2:
        ht.flush()
    sync()
    th = []
    t0 = time.perf_counter()
    for i in range(4, 4 + steps):
        h0 = time.perf_counter()
        step(i)
        th.append(time.perf_counter() - h0)
    if mode == 2:
        ht.flush()
    sync()
    el = time.perf_counter() - t0
    print(f"mode {'ABE'[mode]}: {el / steps * 1e3:.3f} ms per step ({B * steps / el:.0f} FOV/s), host enqueue {np.mean(th) * 1e3:.3f} ms per step")


for m in (0, 1, 2, 0, 2):
    run(m)

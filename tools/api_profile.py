"""Where the time of the reference-level calls goes under N worker threads (one context each):
    python tools/api_profile.py WORKERS [CHUNK]
prints the mean wall time per FOV of each phase as the worker threads saw it, and the aggregate FOV/s."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("AMT_RESULT_PINNED_BYTES", str(4 << 30))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.channels import BRIGHTFIELD, DAPI, FITC, TRITC
from arcadia_microscopy_tools_amd.masks import SegmentationMask
from arcadia_microscopy_tools_amd.model import SegmentationModel

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 48
uniq = [synth.synth_fov(i) for i in range(4)]
fovs = [uniq[i % 4] for i in range(B)]
chans = (BRIGHTFIELD, DAPI, FITC, TRITC)
model = SegmentationModel(backend="classical")
acc = {"segment": 0.0, "mask_ctor": 0.0, "label_image": 0.0, "props": 0.0}
lock = threading.Lock()

def work(part):
    t = {}
    t0 = time.perf_counter()
    masks = model.batch_segment([f[1] for f in part], batch_size=len(part), show_progress=False)
    t["segment"] = time.perf_counter() - t0
    t["mask_ctor"] = t["label_image"] = t["props"] = 0.0
    for f, m in zip(part, masks):
        t0 = time.perf_counter()
        sm = SegmentationMask(m, {c: f[i] for i, c in enumerate(chans)})
        t1 = time.perf_counter()
        sm._labels_device
        t2 = time.perf_counter()
        sm.cell_properties
        t3 = time.perf_counter()
        t["mask_ctor"] += t1 - t0; t["label_image"] += t2 - t1; t["props"] += t3 - t2
    with lock:
        for k in acc:
            acc[k] += t[k]

parts = [fovs[i:i + chunk] for i in range(0, B, chunk)]
if os.environ.get("AMT_API_CPROFILE") == "1":  # one thread under cProfile: which calls the time goes to
    import cProfile, pstats
    for part in parts[:4]:
        work(part)
    pr = cProfile.Profile()
    pr.enable()
    for part in parts[:12]:
        work(part)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
    sys.exit(0)
samples = {}
stop_sampling = threading.Event()


def sampler():
    """Every 2 ms: the innermost frames of all worker threads (where they are, or wait)."""
    import traceback
    me = threading.get_ident()
    while not stop_sampling.is_set():
        for tid, frame in sys._current_frames().items():
            if tid == me:
                continue
            stack = traceback.extract_stack(frame)
            names = [f"{os.path.basename(fr.filename)}:{fr.name}:{fr.lineno}" for fr in stack[-3:]]
            key = " < ".join(reversed(names))
            samples[key] = samples.get(key, 0) + 1
        time.sleep(0.002)


if os.environ.get("AMT_API_SAMPLE") == "1":
    threading.Thread(target=sampler, daemon=True).start()
with ThreadPoolExecutor(max_workers=workers) as ex:
    for _ in range(2):
        list(ex.map(work, parts))
    for k in acc:
        acc[k] = 0.0
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        list(ex.map(work, parts))
    el = time.perf_counter() - t0
print(f"workers {workers} chunk {chunk}: {B * reps / el:.1f} FOV/s; per-FOV thread time ms: " +
      ", ".join(f"{k} {v / (B * reps) * 1e3:.2f}" for k, v in acc.items()))
stop_sampling.set()
if samples:
    tot = sum(samples.values())
    for k, v in sorted(samples.items(), key=lambda kv: -kv[1])[:25]:
        print(f"{100.0 * v / tot:5.1f} %  {k}")

"""Per-operator device timings at 2048 x 2048 (HIP events), for the reference-level operations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arcadia_microscopy_tools_amd import hipops, synth
from arcadia_microscopy_tools_amd.device import get_context

ctx = get_context()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fov = synth.synth_fov(0)
u = ctx.asarray(np.stack([fov[1]] * B))
n = u.size


def timeit(name, fn, bytes_per_px, reps=5):
    fn(); ctx.synchronize()
    t = ctx.timer(); t.start()
    for _ in range(reps):
        fn()
    t.stop(); ms = t.elapsed_ms() / reps
    print(f"{name:34s} {ms / B * 1e3:9.1f} us/plane   {bytes_per_px * n / (ms * 1e-3) / 1e9:8.1f} GB/s algorithmic")


g = hipops.gaussian(u, 2.0)
dog = hipops.difference_of_gaussians(u, 0.6, 16.0)
timeit("gaussian sigma=2 (u16->f64)", lambda: hipops.gaussian(u, 2.0, out=g), 10)
timeit("gaussian sigma=16 (u16->f64)", lambda: hipops.gaussian(u, 16.0, out=g), 10)
timeit("DoG(0.6, 16)", lambda: hipops.difference_of_gaussians(u, 0.6, 16.0, out=dog), 10)
p = hipops.percentile(dog, (1, 99))
timeit("percentile f64 (1,99)", lambda: hipops.percentile(dog, (1, 99), out=p), 8)
p1 = hipops.percentile(dog, 0.0)
timeit("percentile f64 (0)", lambda: hipops.percentile(dog, 0.0, out=p1), 8)
pu = hipops.percentile(u, (0.1, 99.9))
timeit("percentile u16 (0.1,99.9)", lambda: hipops.percentile(u, (0.1, 99.9), out=pu), 2)
r = hipops.rescale(dog, p)
timeit("rescale f64", lambda: hipops.rescale(dog, p, out=r), 16)
timeit("sub_clip0", lambda: hipops.sub_clip0(dog, p1, out=r), 16)
th = hipops.threshold_otsu(u)
timeit("otsu u16 (hist+select)", lambda: hipops.threshold_otsu(u, out=th), 2)
timeit("otsu f64", lambda: hipops.threshold_otsu(g, out=th), 8)
m = hipops.greater_than(g, th)
lab, cnt = hipops.label(m)
timeit("label 8-conn", lambda: hipops.label(m, out=lab, count=cnt), 5)
se = hipops.disk(2)
o = hipops.erosion(u, se)
timeit("grey erosion disk(2) u16", lambda: hipops.erosion(u, se, out=o), 4)
timeit("median disk(2) u16", lambda: hipops.median(u, se, out=o), 4)
timeit("white_tophat disk(7) u16", lambda: hipops.white_tophat(u, hipops.disk(7), out=o), 14, reps=2)
d2, e = hipops.edt(m)
timeit("edt (d2 + f64)", lambda: hipops.edt(m, d2_out=d2, edt_out=e), 9)

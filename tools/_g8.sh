timeout -k 10 300 python3 tools/deliver_probe.py 48 2>&1 | grep mode
python3 - <<'PY'
import numpy as np
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context
ctx = get_context()
rng = np.random.default_rng(0)
for name, arr in (("random full range", rng.integers(0, 65536, (1, 2048, 2048)).astype(np.uint16)),
                  ("random 12 bit", rng.integers(0, 4096, (1, 2048, 2048)).astype(np.uint16)),
                  ("constant", np.full((1, 2048, 2048), 777, np.uint16)),
                  ("odd size", rng.integers(0, 65536, (3, 1023, 777)).astype(np.uint16)),
                  ("8 planes full range", rng.integers(0, 65536, (8, 2048, 2048)).astype(np.uint16)),
                  ("32 planes full range", rng.integers(0, 65536, (32, 2048, 2048)).astype(np.uint16))):
    d = ctx.asarray(arr)
    h = hipops.histogram_u16(d).numpy()
    assert all(np.array_equal(h[i], np.bincount(arr[i].ravel(), minlength=65536)) for i in range(arr.shape[0])), name
    tm = ctx.timer(); ts = []
    for _ in range(5):
        tm.start(); hipops.histogram_u16(d); tm.stop(); ctx.synchronize(); ts.append(tm.elapsed_ms())
    print(f"hist_u16 {name}: {np.median(ts) * 1e3 / arr.shape[0]:.1f} us per plane")
PY
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_api.py -m gpu -x -q -k "hist or otsu or percentile or threshold" 2>&1 | tail -2

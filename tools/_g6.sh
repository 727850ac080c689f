set -o pipefail
O=gpurun_out/r3; mkdir -p $O
for cfg in "1 2" "4 2" "8 2"; do set -- $cfg; timeout -k 10 200 python3 tools/api_profile.py $1 $2 2>&1 | tail -1; done
echo "--- copy threads 4"; AMT_COPY_THREADS=4 timeout -k 10 200 python3 tools/api_profile.py 8 2 2>&1 | tail -1
echo "--- copy threads 16"; AMT_COPY_THREADS=16 timeout -k 10 200 python3 tools/api_profile.py 8 2 2>&1 | tail -1
python3 - <<'PY'
import numpy as np, time
from arcadia_microscopy_tools_amd import hipops
from arcadia_microscopy_tools_amd.device import get_context
ctx = get_context()
rng = np.random.default_rng(0)
for name, arr in (("random full range", rng.integers(0, 65536, (1, 2048, 2048)).astype(np.uint16)),
                  ("random 12 bit", rng.integers(0, 4096, (1, 2048, 2048)).astype(np.uint16)),
                  ("32 planes full range", rng.integers(0, 65536, (32, 2048, 2048)).astype(np.uint16))):
    d = ctx.asarray(arr)
    h = hipops.histogram_u16(d)
    assert all(np.array_equal(h.numpy()[i], np.bincount(arr[i].ravel(), minlength=65536)) for i in range(min(2, arr.shape[0])))
    tm = ctx.timer()
    ts = []
    for _ in range(5):
        tm.start(); hipops.histogram_u16(d); tm.stop(); ctx.synchronize(); ts.append(tm.elapsed_ms())
    print(f"hist_u16 {name}: {np.median(ts) * 1e3 / arr.shape[0]:.1f} us per plane")
PY
timeout -k 10 300 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "hist or otsu or percentile" 2>&1 | tail -2

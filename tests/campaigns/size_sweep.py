"""One-off robustness sweep: config 3 end to end at odd image sizes (every non-aligned fallback path: widths that are
not multiples of 8 / 64, partial tiles, strips and row blocks) through the HIP path and the CPU oracle; labels must be
bit-identical, features within 1e-5.  Run on the GPU box: python tests/campaigns/size_sweep.py [seed]"""
import sys

sys.path.insert(0, '.')
import numpy as np

from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.segment import segment_fovs
from oracle import chains

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
sizes = sorted({int(s) for s in rng.integers(96, 700, 20)} | {128, 256, 257, 320, 511, 512, 513, 640})
bad = 0
for k, s in enumerate(sizes):
    fovs = np.stack([synth.synth_fov(1000 + 7 * k + j, size=s) for j in range(2)])
    res = segment_fovs(fovs, max_cells=1024)
    lab, tabs = res.labels_numpy(), res.feature_tables()
    for j in range(2):
        rl, rp = chains.c3_chain(fovs[j])
        ok = np.array_equal(lab[j], rl)
        pk = all(np.allclose(tabs[j][c], rp[c], rtol=1e-5, atol=1e-8) for c in rp if c != 'orientation')
        bad += (not ok) + (not pk)
        print(s, j, 'labels', ok, 'props', pk, 'cells', int(rl.max()), flush=True)
print('BAD', bad)

"""One-off parity campaign at BASELINE size: config 3 on 12 synthetic 4 x 2048 x 2048 fields of view (indices 2..13)
through the HIP path and through the CPU oracle (12 threads); labels must be bit-identical, features within 1e-5.
Run on the GPU box: python tests/campaigns/parity_campaign.py  (CAMPAIGN_FIRST / CAMPAIGN_LAST select the FOV indices; last runs: 72 / 72 identical)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from arcadia_microscopy_tools_amd import synth
from arcadia_microscopy_tools_amd.segment import segment_fovs
from oracle import chains
import os
idx = list(range(int(os.environ.get("CAMPAIGN_FIRST", "2")), int(os.environ.get("CAMPAIGN_LAST", "14"))))
fovs = np.stack([synth.synth_fov(i) for i in idx])
t0 = time.time()
res = segment_fovs(fovs, max_cells=2048)
lab = res.labels_numpy()
tabs = res.feature_tables()
print('gpu done', time.time() - t0, flush=True)
with ThreadPoolExecutor(max_workers=12) as ex:
    refs = list(ex.map(chains.c3_chain, list(fovs)))
bad = 0
for k, (rl, rp) in enumerate(refs):
    ok = np.array_equal(lab[k], rl)
    pk = all(np.allclose(tabs[k][c], rp[c], rtol=1e-5, atol=1e-8) for c in rp if c != 'orientation')
    print(idx[k], 'labels', ok, 'props', pk, 'cells', int(rl.max()), flush=True)
    bad += (not ok) + (not pk)
print('BAD', bad)

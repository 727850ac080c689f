set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py -m gpu -x -q -k "watershed or chain or c3 or fused" 2>&1 | tail -2
timeout -k 10 300 python3 tests/campaigns/fuzz_watershed.py 2>&1 | tail -1
B="--streams 1 --batch 48 --steps 8 --warmup 2 --no-cpu --no-h2d --no-sublines"
AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/b48_xf.json 2> $O/b48_xf.err && python3 -c "
import json;j=json.load(open('$O/b48_xf.json'));print('b48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
cd /tmp && export TMPDIR=/tmp
AMT_FORK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_xf -o k -- python3 $R/bench.py $B > $O/ks_xf.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks_xf/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "ws_stats" in n or "ccl_tile" in n or "ws_final" in n: print("   %-60s %8.1f us"%(n.split("(")[0][:60],float(r["AverageNs"])/1e3))
PY

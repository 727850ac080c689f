set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py tests/test_gpu_api.py tests/test_gpu_plate.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python3 tests/campaigns/fuzz_labels_props.py 2>&1 | tail -1
timeout -k 10 300 python3 tests/campaigns/fuzz_watershed.py 2>&1 | tail -1
B="--streams 1 --batch 48 --steps 8 --warmup 2 --no-cpu --no-h2d --no-sublines"
for bits in 0 1; do
AMT_CCL_BITS=$bits AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/b48_bits$bits.json 2> $O/b48_bits$bits.err && python3 -c "
import json;j=json.load(open('$O/b48_bits$bits.json'));print('bits $bits b48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
AMT_CCL_BITS=$bits AMT_FORK=0 timeout -k 10 300 python3 bench.py --workload c2 $B > $O/c2_bits$bits.json 2> $O/c2_bits$bits.err && python3 -c "
import json;j=json.load(open('$O/c2_bits$bits.json'));print('bits $bits c2 single ctx', round(j['value']), 'label8', round(j['roofline']['stage_ms']['label8'],3))"
done
timeout -k 10 300 python3 bench.py --no-cpu --no-h2d --no-sublines --steps 10 --warmup 2 > $O/def_bits.json 2> $O/def_bits.err && python3 -c "
import json;j=json.load(open('$O/def_bits.json'));print('default', round(j['value']))"
cd /tmp && export TMPDIR=/tmp
AMT_FORK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_bits -o k -- python3 $R/bench.py $B > $O/ks_bits.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("$O/ks_bits/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    if "ccl_" in n: print("   %-60s %8.1f us x %s"%(n.split("(")[0][:60],float(r["AverageNs"])/1e3, r["Calls"]))
PY

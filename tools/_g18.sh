set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python3 tests/campaigns/fuzz_watershed.py 2>&1 | tail -1
B="--streams 1 --batch 48 --steps 8 --warmup 2 --no-cpu --no-h2d --no-sublines"
AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/b48_p16.json 2> $O/b48_p16.err && python3 -c "
import json;j=json.load(open('$O/b48_p16.json'));print('b48', round(j['value']), {k:round(v,3) for k,v in j['roofline']['stage_ms'].items()})"
timeout -k 10 300 python3 bench.py --no-cpu --no-h2d --no-sublines --steps 10 --warmup 2 > $O/def_p16.json 2> $O/def_p16.err && python3 -c "
import json;j=json.load(open('$O/def_p16.json'));print('default', round(j['value']))"
cd /tmp && export TMPDIR=/tmp
AMT_FORK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_p16 -o k -- python3 $R/bench.py $B > $O/ks_p16.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_p16_c2 -o k -- python3 $R/bench.py --workload c2 --streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d > $O/ks_p16_c2.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
for d in ("ks_p16","ks_p16_c2"):
    f=glob.glob("$O/"+d+"/**/*kernel_stats.csv",recursive=True)[0]
    print(d)
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("ws_","ccl_","roots_","apply_rank","presence","drop_flagged")):
            us=float(r["AverageNs"])/1e3
            if us>8: print("   %-60s %8.1f us x %s"%(n.split("(")[0][:60],us,r["Calls"]))
PY

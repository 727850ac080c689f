set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r3; mkdir -p $O
B="--streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d --no-sublines"
cd /tmp && export TMPDIR=/tmp
AMT_FORK=0 AMT_WS_ANYORDER=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_xs_inorder -o k -- python3 $R/bench.py $B > $O/ks_xs_inorder.log 2>&1 || exit 1
AMT_FORK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_xs_any -o k -- python3 $R/bench.py $B > $O/ks_xs_any.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
for d in ("ks_xs_inorder","ks_xs_any"):
    f=glob.glob("$O/"+d+"/**/*kernel_stats.csv",recursive=True)[0]
    print(d)
    tot=0
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("ws_","ccl_","roots_","presence","drop_flagged")):
            us=float(r["AverageNs"])/1e3
            if us>8: print("   %-60s %8.1f us x %s"%(n.split("(")[0][:60],us,r["Calls"]))
            tot+=float(r["TotalDurationNs"])/float(r["Calls"])*1 if False else 0
PY

set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3/pmc_prep; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES -d $O -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload prep --no-sublines --no-cpu --steps 2 --warmup 1 > $O.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
ls $O
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3/pmc_prep/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "conv_" in k or "gauss" in k:
        print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY

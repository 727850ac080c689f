set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3/pmc_pq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $O -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload prep --no-sublines --no-cpu --steps 2 --warmup 1 > $O.log 2>&1; echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/b -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload prep --no-sublines --no-cpu --steps 2 --warmup 1 > $O.b.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r3/pmc_pq/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:34]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "pq_" in k or "hist" in k or "sub_clip" in k:
            print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "(millions)")
PY

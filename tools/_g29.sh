set -o pipefail
mkdir -p gpurun_out/r3/prof_prep
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_prep -o prep -- python3 $GRAFT_REPO_ROOT/bench.py --workload prep --no-sublines --no-cpu --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/r3/prof_prep.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r3/prof_prep -name "*kernel_stats.csv" | head -1); echo $f
python3 tools/kernel_stats_summary.py $f "prep" | head -16

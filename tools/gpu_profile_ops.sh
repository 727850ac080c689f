#!/bin/bash
# rocprofv3 kernel stats of a bench.py operator workload (prep | filters) -> gpurun_out/<tag>/kernel_stats_<workload>.csv
W=${1:-prep}; TAG=${2:-prof_ops}; O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ARGS="--workload $W --steps 5 --warmup 1 --no-cpu"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt_$W -o w --output-format csv -- python3 bench.py $ARGS > $O/kt_$W.json 2> $O/kt_$W.err; echo "kernel-trace rc=$?"
STATS=$(find $O/kt_$W -name '*kernel_stats.csv' | head -1)
python tools/kernel_stats_summary.py "$STATS" "$ARGS" > $O/kernel_stats_$W.csv && head -30 $O/kernel_stats_$W.csv | sed "s/(.*)\"/\"/"
rm -rf $O/kt_$W

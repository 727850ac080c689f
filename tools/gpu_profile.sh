#!/bin/bash
# rocprofv3 kernel stats (+ optional PMC passes) of the single-stream 32-FOV config-3 chain -> gpurun_out/<tag>/ ;
# usage: bash tools/gpu_profile.sh <tag> [pmc]
TAG=${1:-prof}; O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ARGS="--steps 5 --warmup 1 --no-cpu --no-h2d --streams 1 --batch 32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o c3 --output-format csv -- python3 bench.py $ARGS > $O/kt.json 2> $O/kt.err; echo "kernel-trace rc=$?"
STATS=$(find $O/kt -name '*kernel_stats.csv' | head -1)
python tools/kernel_stats_summary.py "$STATS" "$ARGS" > $O/kernel_stats.csv && head -40 $O/kernel_stats.csv
if [ "$2" = "pmc" ]; then
  PARGS="--steps 2 --warmup 1 --no-cpu --no-h2d --streams 1 --batch 32"
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 bench.py $PARGS > $O/pf.json 2> $O/pf.err; echo "pmc fetch rc=$?"
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 bench.py $PARGS > $O/pw.json 2> $O/pw.err; echo "pmc write rc=$?"
  python tools/pmc_summary.py $O/pmc_f $O/pmc_w auto "$PARGS" > $O/hbm_pmc.csv && cat $O/hbm_pmc.csv
  rm -rf $O/pmc_f $O/pmc_w
fi
rm -rf $O/kt

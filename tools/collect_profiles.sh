#!/bin/bash
# Collect the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   stage 1 = kernel stats: the timed region's launch (one context, 48 FOVs, no auxiliary streams), the default run itself
#             (4 contexts), config 2, prep, filters;
#   stage 2 = the two PMC passes of the 32-FOV chain (FETCH_SIZE / WRITE_SIZE, separate passes);
#   stage 3 = the bench lines (run after the summaries of stages 1-2 are in profiles/: bench.py reads them for `traffic`).
# usage: tools/collect_profiles.sh 1|2|3   -> writes gpurun_out/final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu --no-h2d --no-sublines"
if [ "$1" = "1" ]; then
  AMT_FORK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_b48 -o b48 -- python3 $R/bench.py --streams 1 --batch 48 --steps 5 --warmup 1 $Q > $O/ks_b48.log 2>&1 || exit 1
  echo "48-FOV single-context stats done"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_default -o c3d -- python3 $R/bench.py --steps 5 --warmup 1 $Q > $O/ks_default.log 2>&1 || exit 1
  echo "default-config stats done"
  AMT_FORK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c2 -o c2 -- python3 $R/bench.py --workload c2 --streams 1 --batch 48 --steps 5 --warmup 1 $Q > $O/ks_c2.log 2>&1 || exit 1
  echo "c2 stats done"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_prep -o prep -- python3 $R/bench.py --workload prep --steps 5 --warmup 1 --no-cpu > $O/ks_prep.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_filters -o filters -- python3 $R/bench.py --workload filters --steps 5 --warmup 1 --no-cpu > $O/ks_filters.log 2>&1 || exit 1
  echo "prep / filters stats done"
elif [ "$1" = "2" ]; then
  P="--steps 2 --warmup 1 --streams 1 --batch 32 $Q"
  AMT_FORK=0 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/bench.py $P > $O/pmc_f.log 2>&1 || exit 1
  echo "pmc fetch done"
  AMT_FORK=0 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/bench.py $P > $O/pmc_w.log 2>&1 || exit 1
  echo "pmc write done"
else
  cd $R
  python3 bench.py 2> $O/bench.err > $O/bench.json && echo "default bench (with sublines) done" || exit 1
  python3 bench.py --workload c5 --no-cpu 2> $O/bench_c5.err > $O/bench_c5.json && echo "c5 done" || exit 1
  AMT_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python3 bench.py --no-cpu --no-h2d 2> $O/bench_dist1.err > $O/bench_dist1.json && echo "dist1 done" || exit 1
fi

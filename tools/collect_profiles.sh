#!/bin/bash
# Collect the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   stage 1 = kernel stats (c3 / c2 / prep / filters, one 32-plane launch per kernel) and the two PMC passes of c3;
#   stage 2 = the bench lines (run after stage 1's summaries are in profiles/, bench.py reads them for `traffic`).
# usage: tools/collect_profiles.sh 1|2   -> writes gpurun_out/final/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C3="--steps 5 --warmup 1 --no-cpu --no-h2d --streams 1 --batch 32"
if [ "$1" = "1" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c3 -o c3 -- python3 $R/bench.py $C3 > $O/ks_c3.log 2>&1 || exit 1
  echo "c3 stats done"
  # the default bench configuration itself (4 contexts x 48 FOVs, every call on its context's one stream)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_default -o c3d -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-h2d > $O/ks_default.log 2>&1 || exit 1
  echo "default-config stats done"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c2 -o c2 -- python3 $R/bench.py --workload c2 $C3 > $O/ks_c2.log 2>&1 || exit 1
  echo "c2 stats done"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_prep -o prep -- python3 $R/bench.py --workload prep --steps 5 --warmup 1 --no-cpu > $O/ks_prep.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_filters -o filters -- python3 $R/bench.py --workload filters --steps 5 --warmup 1 --no-cpu > $O/ks_filters.log 2>&1 || exit 1
  echo "prep / filters stats done"
  P="--steps 2 --warmup 1 --no-cpu --no-h2d --streams 1 --batch 32"
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_f -o f --output-format csv -- python3 $R/bench.py $P > $O/pmc_f.log 2>&1 || exit 1
  echo "pmc fetch done"
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_w -o w --output-format csv -- python3 $R/bench.py $P > $O/pmc_w.log 2>&1 || exit 1
  echo "pmc write done"
elif [ "$1" = "5" ]; then
  # one context alone with the timed region's launch size and settings: what bench.py's profiled pass measures
  AMT_FORK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_b48 -o b48 -- python3 $R/bench.py --streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d > $O/ks_b48.log 2>&1 || exit 1
  echo "48-FOV single-context stats done"
elif [ "$1" = "4" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_default -o c3d -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-h2d > $O/ks_default.log 2>&1 || exit 1
  echo "default-config stats done"
elif [ "$1" = "3" ]; then
  cd $R
  AMT_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python3 bench.py --no-cpu --no-h2d 2> $O/bench_dist1.err > $O/bench_dist1.json && echo "dist1 done" || exit 1
  python3 bench.py --unique 64 --tail-reps 24 --no-cpu --no-h2d 2> $O/bench_u64.err > $O/bench_u64.json && echo "u64 done" || exit 1
else
  cd $R
  python3 bench.py 2> $O/bench.err > $O/bench.json && echo "default bench done" || exit 1
  python3 bench.py --workload c2 2> $O/bench_c2.err > $O/bench_c2.json && echo "c2 done" || exit 1
  python3 bench.py --workload prep 2> $O/bench_prep.err > $O/bench_prep.json && echo "prep done" || exit 1
  python3 bench.py --workload filters 2> $O/bench_filters.err > $O/bench_filters.json && echo "filters done" || exit 1
  python3 bench.py --workload c5 --no-cpu 2> $O/bench_c5.err > $O/bench_c5.json && echo "c5 done" || exit 1
  python3 bench.py --plate 48 --no-cpu --no-h2d 2> $O/bench_plate48.err > $O/bench_plate48.json && echo "plate48 done" || exit 1
  AMT_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python3 bench.py --no-cpu --no-h2d 2> $O/bench_dist1.err > $O/bench_dist1.json && echo "dist1 done" || exit 1
  python3 bench.py --unique 64 --tail-reps 24 --no-cpu --no-h2d 2> $O/bench_u64.err > $O/bench_u64.json && echo "u64 done" || exit 1
fi

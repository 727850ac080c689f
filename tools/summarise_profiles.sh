#!/bin/bash
# gpurun_out/final (tools/collect_profiles.sh 1 and 2) -> the summaries committed under profiles/ ($1 = tag, e.g. r03)
T=${1:-r03}; F=gpurun_out/final
python tools/kernel_stats_summary.py $(find $F/ks_b48 -name "*kernel_stats.csv") "--streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d --no-sublines with AMT_FORK=0 (ONE context, the launch size and settings of the default run's contexts)" "# one launch = one stage kernel over a batch of 48 planes of 2048x2048" > profiles/${T}_kernel_stats_b48.csv
python tools/kernel_stats_summary.py $(find $F/ks_default -name "*kernel_stats.csv") "--steps 5 --warmup 1 --no-cpu --no-h2d --no-sublines (the default run: 4 contexts x 48 FOVs side by side; durations are stretched by the neighbours)" > profiles/${T}_kernel_stats_default.csv
python tools/kernel_stats_summary.py $(find $F/ks_c2 -name "*kernel_stats.csv") "--workload c2 --streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d --no-sublines with AMT_FORK=0" "# one launch = one stage kernel over a batch of 48 planes of 2048x2048" > profiles/${T}_kernel_stats_c2.csv
python tools/kernel_stats_summary.py $(find $F/ks_prep -name "*kernel_stats.csv") "--workload prep --steps 5 --warmup 1 --no-cpu" "# one launch = one stage kernel over a batch of 32 planes of 2048x2048" > profiles/${T}_kernel_stats_prep.csv
python tools/kernel_stats_summary.py $(find $F/ks_filters -name "*kernel_stats.csv") "--workload filters --steps 5 --warmup 1 --no-cpu" "# one launch = one stage kernel over a batch of 32 planes of 2048x2048" > profiles/${T}_kernel_stats_filters.csv
python tools/pmc_summary.py $F/pmc_f $F/pmc_w auto "--steps 2 --warmup 1 --streams 1 --batch 32 --no-cpu --no-h2d --no-sublines (AMT_FORK=0)" > profiles/${T}_hbm_pmc.csv

#!/bin/bash
# round-2 record run: the bench lines and rocprofv3 summaries that are committed under profiles/
O=gpurun_out/r2rec; mkdir -p $O
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --tail-reps 24 --unique 64 > $O/bench_u64.json 2> $O/bench_u64.err; echo "bench(unique 64, tail) rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --workload c2 --no-h2d > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 rc=$?"
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu --no-h2d --plate 48 > $O/bench_plate48.json 2> $O/bench_plate48.err; echo "bench plate48 rc=$?"
AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-h2d > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "bench dist1 rc=$?"
for w in prep filters; do timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"; done
timeout -k 10 400 python bench.py --workload c5 --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 rc=$?"
bash tools/gpu_profile.sh r2rec pmc > $O/profile.log 2>&1; echo "profile rc=$?"
bash tools/gpu_profile_ops.sh prep r2rec > /dev/null 2>&1; bash tools/gpu_profile_ops.sh filters r2rec > /dev/null 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], round(d["value"],1), d["unit"], "frac", round(d["roofline"]["frac"],3), d["roofline"].get("kernel"))
    except Exception as e: print(f, "ERR", e)
PY

#!/bin/bash
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_plate.py -x -q > $O/t3.log 2>&1; echo "plate tests rc=$?"; tail -3 $O/t3.log
D="AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
env $D MASTER_PORT=29511 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu > $O/b_dist1.json 2> $O/b_dist1.err; echo "bench(dist1) rc=$?"
for s in 2 3 4 6; do
 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-h2d --plate 48 --streams $s > $O/b_p48_s$s.json 2> $O/b_p48_s$s.err; echo "bench(48/GPU, $s streams) rc=$?"
done
env $D MASTER_PORT=29513 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --plate 48 > $O/b_p48_dist.json 2> $O/b_p48_dist.err; echo "bench(48 dist) rc=$?"
for f in b_dist1 b_p48_s2 b_p48_s3 b_p48_s4 b_p48_s6 b_p48_dist; do python - <<PY
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"]), d["n_gpus"], d["scaling"], d["config"]["streams_per_gpu"], d["ms_per_step"])
except Exception as e:
    print("$f", "no json", e)
PY
done

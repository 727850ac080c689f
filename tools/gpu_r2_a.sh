#!/bin/bash
# round-2 GPU session A: plate tests, runtime comparison, distributed rehearsals
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_plate.py -x -q > $O/t2.log 2>&1; echo "plate tests rc=$?"; tail -3 $O/t2.log
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-h2d > $O/b_torchrt.json 2> $O/b_torchrt.err; echo "bench(torch rt) rc=$?"
AMT_HIP_RUNTIME=system timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-h2d > $O/b_sysrt.json 2> $O/b_sysrt.err; echo "bench(system rt) rc=$?"
AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu > $O/b_dist1.json 2> $O/b_dist1.err; echo "bench(dist1) rc=$?"
AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --plate 384 > $O/b_plate1.json 2> $O/b_plate1.err; echo "bench(plate384 dist1) rc=$?"
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --no-h2d --plate 48 > $O/b_p48.json 2> $O/b_p48.err; echo "bench(48/GPU) rc=$?"
AMT_BENCH_BACKEND=gloo AMT_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu --plate 96 --unique 4 > $O/b_gloo2.json 2> $O/b_gloo2.err; echo "bench(2 ranks gloo, shared GPU) rc=$?"
for f in b_torchrt b_sysrt b_dist1 b_plate1 b_p48 b_gloo2; do python - <<PY
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"]), d["n_gpus"], d["scaling"], d["config"]["streams_per_gpu"], d["config"].get("feature_table_exchange"))
except Exception as e:
    print("$f", "no json", e)
PY
done

#!/bin/bash
O=gpurun_out/r2; mkdir -p $O
timeout -k 10 250 python tools/stress_feeder.py 200 feeder 2>&1 | tail -1
D="AMT_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > $O/b2.json 2> $O/b2.err; echo "bench rc=$?"
env $D MASTER_PORT=29511 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu > $O/b2_dist1.json 2> $O/b2_dist1.err; echo "bench(dist1) rc=$?"
for s in 2 4; do
 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu --no-h2d --plate 48 --streams $s > $O/b2_p48_s$s.json 2> $O/b2_p48_s$s.err; echo "bench(48/GPU, $s streams) rc=$?"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu --no-h2d --workload c2 > $O/b2_c2.json 2> $O/b2_c2.err; echo "bench c2 rc=$?"
for f in b2 b2_dist1 b2_p48_s2 b2_p48_s4 b2_c2; do python - <<PY
import json
try:
    d=json.load(open("$O/$f.json")); print("$f", round(d["value"]), d["n_gpus"], d["scaling"], d["config"]["streams_per_gpu"], round(d["ms_per_step"],3), d.get("pcie_inclusive",{}).get("value"), {k:round(v,3) for k,v in d["roofline"]["stage_ms"].items()})
except Exception as e:
    print("$f", "no json", e)
PY
done

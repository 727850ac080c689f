set -o pipefail
mkdir -p gpurun_out/r3
AMT_BENCH_BACKEND=gloo AMT_BENCH_SHARE_GPU=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --batch 48 > gpurun_out/r3/gloo2.json 2> gpurun_out/r3/gloo2.err; echo "rc=$?"
tail -5 gpurun_out/r3/gloo2.err
python3 - <<'PY'
import json
ls=[l for l in open("gpurun_out/r3/gloo2.json") if l.startswith("{")]
d=json.loads(ls[-1]); print({k:d[k] for k in ("value","n_gpus","ms_per_step","scaling")}, d.get("config",{}).get("parallelism"), list(d.keys()))
PY
timeout -k 10 400 python3 tests/campaigns/fuzz_filters.py 150 > gpurun_out/r3/fuzz_filters.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r3/fuzz_filters.log

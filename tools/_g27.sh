set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_plate.py -x -q > gpurun_out/r3/t_api3.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r3/t_api3.log
timeout -k 10 300 python3 bench.py --workload api --steps 5 --warmup 2 > gpurun_out/r3/api_onecall.json 2> gpurun_out/r3/api_onecall.err; echo "rc=$?"
python3 - <<PY
import json
d=json.load(open("gpurun_out/r3/api_onecall.json")); print("two calls", round(d["value"],1), "one call", d["one_call"])
PY

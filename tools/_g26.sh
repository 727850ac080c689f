set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 300 python3 -m pytest tests/test_gpu_api.py -x -q -k "batch_masks or segmentation" > gpurun_out/r3/t_masks.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r3/t_masks.log
for cfg in "4 4" "4 8" "2 6" "8 3"; do set -- $cfg
AMT_API_MASK_WORKERS=$1 AMT_API_MASK_CHUNK=$2 timeout -k 10 300 python3 bench.py --workload api --steps 4 --warmup 2 > gpurun_out/r3/api_masks_$1_$2.json 2> gpurun_out/r3/api_masks_$1_$2.err; echo "rc=$?"
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3/api_masks_$1_$2.json")); print("workers $1 chunk $2: two calls", round(d["value"],1), "one call", round(d["one_call"]["value"],1))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r3/api_masks_$1_$2.err").read()[-1500:])
PY
done

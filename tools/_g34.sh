set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_api.py -x -q -k "percentile or rescale or dog or operators or pipeline or readme" > gpurun_out/r3/t_pq.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_pq.log
timeout -k 10 300 python3 bench.py --workload prep --no-sublines --no-cpu > gpurun_out/r3/prep_pq.json 2> gpurun_out/r3/prep_pq.err; echo "rc=$?"
grep "stage ms" gpurun_out/r3/prep_pq.err | tail -1
python3 -c "
import json; d=json.load(open('gpurun_out/r3/prep_pq.json')); print(round(d['value'],1), d['unit'])"

set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_api.py -x -q -k "percentile or rescale or dog or operators or pipeline or readme or threshold" > gpurun_out/r3/t_pq.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_pq.log
timeout -k 10 400 python3 tests/campaigns/fuzz_api.py 120 > gpurun_out/r3/fuzz_api.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r3/fuzz_api.log
bash tools/_g29.sh | awk -F'",' '{split($1,a,"("); print a[1], $2}' | head -12

set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py -m gpu -x -q -k "watershed or plain" 2>&1 | tail -2
bash tools/collect_profiles.sh 3

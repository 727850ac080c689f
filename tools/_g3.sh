set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_cellpose.py -m gpu -x -q > $O/t_cellpose.log 2>&1; echo "cellpose tests rc=$?"; tail -30 $O/t_cellpose.log

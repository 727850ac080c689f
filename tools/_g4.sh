set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_api.py -m gpu -x -q -k "classical or stream_rule or other_dtypes" > $O/t_api.log 2>&1; echo "api tests rc=$?"; tail -5 $O/t_api.log
timeout -k 10 900 python3 bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"; tail -40 $O/bench_full.err

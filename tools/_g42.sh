set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all5.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_all5.log
timeout -k 10 400 python3 tests/campaigns/fuzz_labels_props.py 120 > gpurun_out/r3/fuzz_lp.log 2>&1; echo "fuzz_lp rc=$?"; tail -2 gpurun_out/r3/fuzz_lp.log
timeout -k 10 400 python3 tests/campaigns/fuzz_watershed.py 150 > gpurun_out/r3/fuzz_ws.log 2>&1; echo "fuzz_ws rc=$?"; tail -2 gpurun_out/r3/fuzz_ws.log
AMT_FORK=0 timeout -k 10 300 python3 bench.py --workload c2 --streams 1 --batch 48 --steps 5 --warmup 2 --no-sublines --no-cpu --no-h2d > gpurun_out/r3/c2_s1.json 2> gpurun_out/r3/c2_s1.err; grep "stage ms" gpurun_out/r3/c2_s1.err | tail -1
AMT_FORK=0 timeout -k 10 300 python3 bench.py --streams 1 --batch 48 --steps 5 --warmup 2 --no-sublines --no-cpu --no-h2d > gpurun_out/r3/c3_s1.json 2> gpurun_out/r3/c3_s1.err; grep "stage ms" gpurun_out/r3/c3_s1.err | tail -1

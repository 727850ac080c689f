set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_api.py -x -q -k "percentile or rescale or dog or operators or pipeline or readme or threshold" > gpurun_out/r3/t_pq.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_pq.log
for g in 0 64 128 256 512; do
AMT_PQ_GRID=$g timeout -k 10 300 python3 bench.py --workload prep --no-sublines --no-cpu > gpurun_out/r3/prep_g$g.json 2> gpurun_out/r3/prep_g$g.err; echo "grid $g rc=$?"
grep "stage ms" gpurun_out/r3/prep_g$g.err | tail -1
done

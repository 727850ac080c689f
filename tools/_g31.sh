set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_api.py -x -q > gpurun_out/r3/t_filt2.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_filt2.log
timeout -k 10 400 python3 tests/campaigns/fuzz_filters.py --cases 150 > gpurun_out/r3/fuzz_filters.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r3/fuzz_filters.log
for v in new norc; do
  case $v in new) envs="";; norc) envs="AMT_GAUSS_NO_RC=1";; esac
  env $envs timeout -k 10 300 python3 bench.py --workload prep --no-sublines --no-cpu > gpurun_out/r3/prep_$v.json 2> gpurun_out/r3/prep_$v.err; echo "$v rc=$?"
  grep "stage ms" gpurun_out/r3/prep_$v.err | tail -1
done

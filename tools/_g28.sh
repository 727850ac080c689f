set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py -x -q -k "gauss or dog or filter" > gpurun_out/r3/t_filt.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_filt.log
for v in new norc old; do
  case $v in
    new) envs="";;
    norc) envs="AMT_GAUSS_NO_RC=1";;
    old) envs="AMT_HIP_LIB=$PWD/tools/variants/libamt_oldfilters.so";;
  esac
  env $envs timeout -k 10 300 python3 bench.py --workload prep --no-sublines --no-cpu > gpurun_out/r3/prep_$v.json 2> gpurun_out/r3/prep_$v.err; echo "$v rc=$?"
  grep "stage ms" gpurun_out/r3/prep_$v.err | tail -1
done

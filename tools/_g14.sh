set -o pipefail
O=gpurun_out/r3; mkdir -p $O
R=$PWD
B="--streams 1 --batch 48 --steps 8 --warmup 2 --no-cpu --no-h2d --no-sublines"
run() { # name, env...
  name=$1; shift
  env "$@" AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/v_$name.json 2> $O/v_$name.err && python3 -c "
import json;j=json.load(open('$O/v_$name.json'));print('$name b48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
  env "$@" timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d --steps 20 --warmup 3 > $O/vp_$name.json 2> $O/vp_$name.err && python3 -c "
import json;j=json.load(open('$O/vp_$name.json'));print('$name plate48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
}
run classes AMT_WS_PERSIST=0
run w16su6 AMT_WS_PERSIST=1
run w12su8 AMT_HIP_LIB=$R/tools/variants/libamt_w12.so
run w8su8 AMT_HIP_LIB=$R/tools/variants/libamt_w8.so
run w16su6_again AMT_WS_PERSIST=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py -m gpu -x -q -k "watershed or chain or c3 or fused" 2>&1 | tail -2

set -o pipefail
O=gpurun_out/r3; mkdir -p $O
B="--streams 1 --batch 48 --steps 8 --warmup 2 --no-cpu --no-h2d --no-sublines"
run() { # name, env...
  name=$1; shift
  env "$@" AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/w_$name.json 2> $O/w_$name.err && python3 -c "
import json;j=json.load(open('$O/w_$name.json'));print('$name b48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
  env "$@" timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d --steps 20 --warmup 3 > $O/wp_$name.json 2> $O/wp_$name.err && python3 -c "
import json;j=json.load(open('$O/wp_$name.json'));print('$name plate48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
  env "$@" timeout -k 10 300 python3 bench.py --no-cpu --no-h2d --no-sublines --steps 10 --warmup 2 > $O/wd_$name.json 2> $O/wd_$name.err && python3 -c "
import json;j=json.load(open('$O/wd_$name.json'));print('$name default', round(j['value']))"
}
run classes AMT_WS_PERSIST=0
run full8 AMT_WS_PERSIST=1
run half8 AMT_WS_PERSIST=2
run half6 AMT_WS_PERSIST=2 AMT_WS_PF_WAVES=6
run half12 AMT_WS_PERSIST=2 AMT_WS_PF_WAVES=12
run classes_again AMT_WS_PERSIST=0
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py tests/test_gpu_api.py -m gpu -x -q -k "watershed or chain or c3 or fused or other_dtypes" 2>&1 | tail -2
timeout -k 10 600 python3 tests/campaigns/fuzz_watershed.py 2>&1 | tail -1

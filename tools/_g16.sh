set -o pipefail
O=gpurun_out/r3; mkdir -p $O
B="--streams 1 --batch 48 --steps 8 --warmup 2 --no-cpu --no-h2d --no-sublines"
run() { # name, env...
  name=$1; shift
  env "$@" AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/x_$name.json 2> $O/x_$name.err && python3 -c "
import json;j=json.load(open('$O/x_$name.json'));print('$name b48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
  env "$@" timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d --steps 20 --warmup 3 > $O/xp_$name.json 2> $O/xp_$name.err && python3 -c "
import json;j=json.load(open('$O/xp_$name.json'));print('$name plate48', round(j['value']), 'ws', round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
  env "$@" timeout -k 10 300 python3 bench.py --no-cpu --no-h2d --no-sublines --steps 10 --warmup 2 > $O/xd_$name.json 2> $O/xd_$name.err && python3 -c "
import json;j=json.load(open('$O/xd_$name.json'));print('$name default', round(j['value']))"
}
run classes AMT_WS_PERSIST=0
run full63 AMT_WS_PERSIST=1
run s46 AMT_WS_PERSIST=1 AMT_WS_PF_SLOTS=46
run s40w6 AMT_WS_PERSIST=1 AMT_WS_PF_SLOTS=40 AMT_WS_PF_WAVES=6
run full63w6 AMT_WS_PERSIST=1 AMT_WS_PF_WAVES=6
run full63w10 AMT_WS_PERSIST=1 AMT_WS_PF_WAVES=10
run classes_again AMT_WS_PERSIST=0

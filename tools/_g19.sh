set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python3 bench.py --workload a8 --steps 1 --warmup 1 > $O/a8_v2.json 2> $O/a8_v2.err && python3 -c "
import json;j=json.load(open('$O/a8_v2.json'));print('a8 48 planes', round(j['value'],2),'FOV/s', round(j['ms_per_step']), 'ms/step', j['roofline']['stage_ms'])"
timeout -k 10 300 python3 bench.py --workload api --steps 3 --warmup 2 > $O/api4.json 2> $O/api4.err && python3 -c "
import json;j=json.load(open('$O/api4.json'));print('api', round(j['value'],1))"
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu > $O/c5_v2.json 2> $O/c5_v2.err && python3 -c "
import json;j=json.load(open('$O/c5_v2.json'));print('c5', round(j['value'],1), 'tiles/s; fwd frac', round(j['roofline']['frac'],4), 'post ms/tile', round(j['roofline']['postprocessing_ms_per_tile'],3), 'masks', j['config']['masks_per_tile_mean'])"

set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_chain.py tests/test_gpu_ops.py -m gpu -x -q > $O/t_chain.log 2>&1; echo "chain+ops tests rc=$?"; tail -4 $O/t_chain.log
B="--streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d --no-sublines"
AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/b48_xs.json 2> $O/b48_xs.err && python3 -c "
import json;j=json.load(open('$O/b48_xs.json'));print('b48', round(j['value']), {k:round(v,3) for k,v in j['roofline']['stage_ms'].items()})"
timeout -k 10 300 python3 bench.py --no-cpu --no-h2d --no-sublines > $O/def_xs.json 2> $O/def_xs.err && python3 -c "
import json;j=json.load(open('$O/def_xs.json'));print('default', round(j['value']), round(j['ms_per_step'],3))"
timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d --steps 20 --warmup 3 > $O/p48_xs.json 2> $O/p48_xs.err && python3 -c "
import json;j=json.load(open('$O/p48_xs.json'));print('plate48', round(j['value']), {k:round(v,3) for k,v in j['roofline']['stage_ms'].items()})"
timeout -k 10 200 python3 bench.py --workload api --steps 3 --warmup 2 > $O/api2.json 2> $O/api2.err && python3 -c "
import json;j=json.load(open('$O/api2.json'));print('api', round(j['value'],1))"
AMT_API_WORKERS=6 AMT_API_CHUNK=2 timeout -k 10 200 python3 bench.py --workload api --steps 3 --warmup 2 > $O/api3.json 2> $O/api3.err && python3 -c "
import json;j=json.load(open('$O/api3.json'));print('api 6 workers', round(j['value'],1))"

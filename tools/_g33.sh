set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r3/t_all4.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_all4.log
for w in 1 4 8; do timeout -k 10 200 python3 tools/api_profile.py $w 2>&1 | grep -v amdgpu.ids | tail -1; done
for w in 1 4 8; do AMT_GIL=release timeout -k 10 200 python3 tools/api_profile.py $w 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout -k 10 300 python3 bench.py --workload api --steps 5 --warmup 2 > gpurun_out/r3/api_b.json 2> gpurun_out/r3/api_b.err; echo "rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r3/api_b.json')); print('two calls', round(d['value'],1), 'one call', round(d['one_call']['value'],1))"

set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py -x -q -k "watershed or plain or ties or a8" > gpurun_out/r3/t_wsg.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/t_wsg.log
timeout -k 10 600 python3 bench.py --workload a8 --batch 8 --steps 1 --warmup 1 > gpurun_out/r3/a8_8.json 2> gpurun_out/r3/a8_8.err; echo "rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r3/a8_8.json')); print('a8 8 planes', round(d['value'],2), 'FOV/s', round(d['ms_per_step']), 'ms/step tied', d.get('tied_plane_fraction'))"

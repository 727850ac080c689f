set -o pipefail
O=gpurun_out/r3; mkdir -p $O
./tools/probes/anyorder_probe2.bin 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py -m gpu -x -q -k "watershed or plain or chain or c3" > $O/t_wsg.log 2>&1; echo "ws tests rc=$?"; tail -4 $O/t_wsg.log
timeout -k 10 300 python3 bench.py --workload a8 --steps 1 --warmup 1 --batch 8 > $O/a8_8.json 2> $O/a8_8.err && python3 -c "
import json;j=json.load(open('$O/a8_8.json'));print('a8 8 planes', round(j['value'],2),'FOV/s', round(j['ms_per_step']), 'ms/step tied', j['tied_plane_fraction'])"
timeout -k 10 300 python3 bench.py --workload a8 --steps 1 --warmup 1 --batch 48 > $O/a8_48.json 2> $O/a8_48.err && python3 -c "
import json;j=json.load(open('$O/a8_48.json'));print('a8 48 planes', round(j['value'],2),'FOV/s', round(j['ms_per_step']), 'ms/step')"

set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 300 python3 bench.py --workload c2 --steps 10 --warmup 2 --no-cpu --no-h2d > $O/c2_p16.json 2> $O/c2_p16.err && python3 -c "
import json;j=json.load(open('$O/c2_p16.json'));print('c2', round(j['value']), {k:round(v,3) for k,v in j['roofline']['stage_ms'].items()})"
timeout -k 10 300 python3 tests/campaigns/fuzz_labels_props.py 2>&1 | tail -2

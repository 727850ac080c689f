set -o pipefail
mkdir -p gpurun_out/r3
for lt in 0 1; do
AMT_BENCH_LOW_TRAFFIC=$lt timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-sublines --no-cpu --no-h2d > gpurun_out/r3/lt_$lt.json 2> gpurun_out/r3/lt_$lt.err; echo "rc=$?"
grep "stage ms" gpurun_out/r3/lt_$lt.err | tail -1
python3 -c "
import json; d=json.load(open('gpurun_out/r3/lt_$lt.json')); print('low_traffic $lt:', round(d['value']), 'FOV/s', round(d['ms_per_step'],2), 'ms')"
done

set -o pipefail
mkdir -p gpurun_out/r3
for cfg in "4 192" "3 192" "6 192" "8 192" "4 256" "4 128" "5 240"; do set -- $cfg
timeout -k 10 300 python3 bench.py --streams $1 --batch $2 --steps 10 --warmup 3 --no-sublines --no-cpu --no-h2d > gpurun_out/r3/sw_$1_$2.json 2> gpurun_out/r3/sw_$1_$2.err; 
python3 -c "
import json; d=json.load(open('gpurun_out/r3/sw_$1_$2.json')); print('streams $1 batch $2:', round(d['value']), 'FOV/s', round(d['ms_per_step'],2), 'ms')"
done

set -o pipefail
O=gpurun_out/r3; mkdir -p $O
for cfg in "1 1" "4 2" "8 4" "12 4" "8 2" "16 3"; do
  set -- $cfg
  AMT_API_WORKERS=$1 AMT_API_CHUNK=$2 timeout -k 10 200 python3 bench.py --workload api --steps 3 --warmup 2 > $O/api_$1_$2.json 2> $O/api_$1_$2.err || { tail -5 $O/api_$1_$2.err; exit 1; }
  python3 -c "import json;j=json.load(open('$O/api_$1_$2.json'));print('workers $1 chunk $2:', round(j['value'],1), 'FOV/s')"
done
timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d --steps 20 --warmup 3 > $O/p48_deliver.json 2> $O/p48_deliver.err && python3 -c "import json;j=json.load(open('$O/p48_deliver.json'));print('plate48 deliver', round(j['value']))"
timeout -k 10 300 python3 bench.py --no-cpu --no-h2d --no-sublines > $O/def_deliver.json 2> $O/def_deliver.err && python3 -c "import json;j=json.load(open('$O/def_deliver.json'));print('default deliver', round(j['value']))"
timeout -k 10 300 python3 -m pytest tests/test_gpu_api.py -m gpu -x -q > $O/t_api2.log 2>&1; echo "api tests rc=$?"; tail -3 $O/t_api2.log

set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_chain.py -m gpu -x -q -k "watershed or plain or chain or c3 or fused or host_tables" > $O/t_persist.log 2>&1; echo "ws tests rc=$?"; tail -4 $O/t_persist.log
B="--streams 1 --batch 48 --steps 5 --warmup 1 --no-cpu --no-h2d --no-sublines"
for P in 0 1; do
AMT_WS_PERSIST=$P AMT_FORK=0 timeout -k 10 300 python3 bench.py $B > $O/b48_p$P.json 2> $O/b48_p$P.err && python3 -c "
import json;j=json.load(open('$O/b48_p$P.json'));print('persist $P b48', round(j['value']), {k:round(v,3) for k,v in j['roofline']['stage_ms'].items()})"
AMT_WS_PERSIST=$P timeout -k 10 300 python3 bench.py --no-cpu --no-h2d --no-sublines > $O/def_p$P.json 2> $O/def_p$P.err && python3 -c "
import json;j=json.load(open('$O/def_p$P.json'));print('persist $P default', round(j['value']), round(j['ms_per_step'],3))"
AMT_WS_PERSIST=$P timeout -k 10 300 python3 bench.py --plate 48 --no-cpu --no-h2d --steps 20 --warmup 3 > $O/p48_p$P.json 2> $O/p48_p$P.err && python3 -c "
import json;j=json.load(open('$O/p48_p$P.json'));print('persist $P plate48', round(j['value']), round(j['roofline']['stage_ms']['watershed_clear_relabel'],3))"
done
timeout -k 10 600 python3 tests/campaigns/fuzz_watershed.py 2>&1 | tail -3

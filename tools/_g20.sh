set -o pipefail
O=gpurun_out/r3; mkdir -p $O
for st in 1 2 3 4 6; do
timeout -k 10 300 python3 bench.py --plate 48 --streams $st --no-cpu --no-h2d --steps 20 --warmup 3 > $O/ps_$st.json 2> $O/ps_$st.err && python3 -c "
import json;j=json.load(open('$O/ps_$st.json'));print('plate48 streams $st', round(j['value']))"
done
for st in 3 4 6; do
timeout -k 10 300 python3 bench.py --streams $st --no-cpu --no-h2d --no-sublines --steps 10 --warmup 2 > $O/ds_$st.json 2> $O/ds_$st.err && python3 -c "
import json;j=json.load(open('$O/ds_$st.json'));print('default streams $st', round(j['value']))"
done

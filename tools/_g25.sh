set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python3 bench.py > $O/bench_final2.json 2> $O/bench_final2.err; echo "bench rc=$?"; tail -12 $O/bench_final2.err
python3 - <<PY
import json
j=json.load(open("$O/bench_final2.json"))
print("value", round(j["value"]))
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["all_cores"], j["cpu_baseline"]["all_cores_processes"])
print({k:(round(v.get("value",-1),1), v.get("error","")) for k,v in j["sublines"].items()})
PY

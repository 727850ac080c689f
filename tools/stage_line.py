"""Print the per-stage device times and the throughput of a bench.py JSON line (stdin)."""
import json, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j["roofline"]
st = r.get("stage_ms") or {k: v["ms"] for k, v in r.get("stages", {}).items()}
print(f'{j["value"]:.0f} {j["unit"]} | ' + " ".join(f'{k}={v:.3f}' for k, v in st.items()) + f' | chain {sum(st.values()):.3f} ms')

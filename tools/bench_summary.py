"""the flat scalars of a bench line (value + roofline) on a few lines:  python tools/bench_summary.py FILE"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", round(d["value"], 1), "ms_per_step", round(d["ms_per_step"], 3), "kernel", r.get("kernel"))
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (dict, list, str))})
g = d.get("mixed", {}).get("cli_gzip", {})
print("cli_gzip", {k: g.get(k) for k in ("value", "stream_value", "wall_s")}, g.get("phases_s"))
print("cpu_baseline", d.get("cpu_baseline", {}).get("value"))

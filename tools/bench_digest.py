#!/usr/bin/env python3
"""Print the numbers of a bench.py JSON line that a session log should show (the full line stays in the file)."""
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("bench line unreadable:", e); sys.exit(0)
def g(x, *ks):
    for k in ks:
        x = x.get(k) if isinstance(x, dict) else None
    return x
print("value", round(d["value"]), d["unit"], "| ms/step", round(d["ms_per_step"], 3), "| steps", d["steps"], "| p50", d.get("p50_latency_ms"))
print("roofline", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (d.get("roofline") or {}).items() if not isinstance(v, (dict, list, str))})
for k in ("stage_ms_per_call", "result_check", "grid_config4", "demo_scale", "large_live", "mid_live", "value_incl_h2d", "decision_sensitivity", "u0_stability"):
    if d.get(k) is not None:
        print(k, json.dumps(d[k])[:900])
cb = d.get("cpu_baseline")
if cb:
    print("cpu_baseline", {k: cb[k] for k in cb if k in ("value", "cores", "kind", "value_pruned", "value_pruned_1thread", "identical_to_gpu")}, g(cb, "pruned_pair_parallel"))

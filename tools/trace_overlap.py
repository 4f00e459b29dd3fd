#!/usr/bin/env python3
"""Kernel-trace overlap analysis: python tools/trace_overlap.py <kernel_trace.csv> [skip_first_n_solves]
Reports, for the steady-state part of a bench run, the wall span per step, the time during which exactly the
listed sets of kernels are resident, and GPU idle time."""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("roman::", "")
    n = re.sub(r"<.*", "", n)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
solves = [e for e in ev if e[2] in ("k_solve_up", "k_solve_stream")]
big = [e for e in solves if e[1] - e[0] > 1_000_000]
if len(big) < 4:
    print("too few batch launches"); sys.exit()
t0, t1 = big[2][0], big[-1][0]          # steady state: from the 3rd big solve start to the last big solve start
nsteps = len(big) - 3
pts = []
for s, e, n in ev:
    if e <= t0 or s >= t1 or not n.startswith("k_"): continue
    pts.append((max(s, t0), 1, n)); pts.append((min(e, t1), -1, n))
pts.sort()
active = collections.Counter(); last = t0; acc = collections.Counter()
for t, d, n in pts:
    key = "+".join(sorted(k for k, v in active.items() if v > 0)) or "(idle)"
    acc[key] += t - last; last = t
    active[n] += d
acc["+".join(sorted(k for k, v in active.items() if v > 0)) or "(idle)"] += t1 - last
span = t1 - t0
print(f"steps {nsteps}  span/step {span/nsteps/1e6:.3f} ms")
for k, v in acc.most_common(14):
    print(f"  {v/nsteps/1e6:7.3f} ms/step  {100*v/span:5.1f}%  {k}")

#!/usr/bin/env python3
"""Timeline of the last kernels of a rocprofv3 kernel trace: python tools/trace_timeline.py <kernel_trace.csv> [last_ms=4]
One line per kernel launch (start, duration, stream/queue id, name) — do the calls of a pipelined loop overlap?"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
ev = []
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("roman::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
ev = [e for e in ev if e[2].startswith("k_")]
t_end = ev[-1][1]
ev = [e for e in ev if e[0] >= t_end - last_ms * 1e6]
t0 = ev[0][0]
for s, e, n, q, st in ev:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q} s{st}  {n}")

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter CSVs (one directory per pass) into a per-kernel table:
mean counter value per dispatch, for the dispatches of the LARGEST grid of each kernel (the
B=256 bench launches, not the B=1 latency probes)."""
import collections, csv, glob, os, re, sys

root = sys.argv[1]
table = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        by = collections.defaultdict(list)
        for r in rows:
            name = re.sub(r"\(.*", "", r.get("Kernel_Name", "")).replace("void ", "").replace("roman::", "")
            by[(name, r["Counter_Name"])].append((int(r.get("Grid_Size", 0) or 0), int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for (name, ctr), vals in by.items():
            gmax = max(v[0] for v in vals)
            per_disp = collections.defaultdict(float)
            for g, disp, v in vals:
                if g == gmax:
                    per_disp[disp] += v
            table[name][ctr] = sum(per_disp.values()) / max(len(per_disp), 1)
ctrs = sorted({c for k in table.values() for c in k})
names = [n for n in table if n.startswith("k_")]
for n in sorted(names, key=lambda x: -table[x].get("SQ_WAVE_CYCLES", 0)):
    print(f"== {n}")
    for c in ctrs:
        if c in table[n]:
            print(f"   {c:28s} {table[n][c]:16.1f}")
    t = table[n]
    if "SQ_WAVE_CYCLES" in t and t["SQ_WAVE_CYCLES"] > 0:
        wc = t["SQ_WAVE_CYCLES"]
        print(f"   -> wait_any/wave_cycles={t.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst_any={t.get('SQ_WAIT_INST_ANY',0)/wc:.2f} active_inst_any={t.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}"
              f" valu_insts/wave={t.get('SQ_INSTS_VALU',0)/max(t.get('SQ_WAVES',1),1):.0f} salu/wave={t.get('SQ_INSTS_SALU',0)/max(t.get('SQ_WAVES',1),1):.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in t and t.get("SQ_BUSY_CYCLES", 0) > 0:
        print(f"   -> matrix pipe busy {t['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles per SIMD and launch (1024 SIMDs; divide by the launch's duration in cycles for the busy fraction)")
    if "SQ_LDS_IDX_ACTIVE" in t and t["SQ_LDS_IDX_ACTIVE"] > 0:
        print(f"   -> lds_bank_conflict/lds_idx_active={t.get('SQ_LDS_BANK_CONFLICT', 0) / t['SQ_LDS_IDX_ACTIVE']:.3f} lds_insts/wave={t.get('SQ_INSTS_LDS', 0) / max(t.get('SQ_WAVES', 1), 1):.0f}")
    if "FETCH_SIZE" in t or "WRITE_SIZE" in t:
        print(f"   -> FETCH_SIZE={t.get('FETCH_SIZE',0)/1024:.1f} MB (x2 on gfx950 for wide streams) WRITE_SIZE={t.get('WRITE_SIZE',0)/1024:.1f} MB (KB units)")

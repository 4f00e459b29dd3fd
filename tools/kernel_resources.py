#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/SGPR/scratch/LDS/occupancy) for libroman_hip's
kernels, from hipcc -Rpass-analysis=kernel-resource-usage (cross-compiles without a GPU)."""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "roman_amd", "csrc", "roman_hip.hip")
with tempfile.TemporaryDirectory() as td:
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
           "-o", os.path.join(td, "r.o"), "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        name = re.sub(r"\(.*", "", name).replace("roman::", "").replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
cols = ["VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"]
print(f"{'kernel':60s} " + " ".join(f"{c.split(' ')[0]:>10s}" for c in cols))
for r in rows:
    print(f"{r['name'][:60]:60s} " + " ".join(f"{r.get(c,'?'):>10s}" for c in cols))

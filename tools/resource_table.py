#!/usr/bin/env python3
"""Per-kernel registers / scratch / LDS from `make -C qwen3-rs_amd asm` (ASMDIR/resource_usage.txt).
usage: resource_table.py [filter-substring] [--spec]   (--spec: only shape-specialised k_gemv instantiations)"""
import re, subprocess, sys
f = "/tmp/q3_asm/resource_usage.txt"
flt = [a for a in sys.argv[1:] if not a.startswith("--")]
spec = "--spec" in sys.argv
rows, cur = [], None
for line in open(f):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, tag in (("VGPRs:", "vgpr"), ("AGPRs:", "agpr"), ("ScratchSize", "scratch"), ("Occupancy", "occ"), ("LDS Size", "lds"), ("SGPRs:", "sgpr")):
        if key in line:
            cur[tag] = line.split(":")[-1].strip().split()[0]
names = "\n".join(r["name"] for r in rows)
dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"\(q3::\w+\)", "", d).replace("q3::", "")
    d = re.sub(r"\(.*\)$", "", d).replace("void ", "")
    if flt and not all(x in d for x in flt):
        continue
    if spec:
        m = re.search(r"k_gemv<(.*)>", d)
        if not m:
            continue
        parts = [x.strip() for x in m.group(1).split(",")]
        if len(parts) < 8 or parts[7] == "0":
            continue
    print(f"{d:70s} vgpr {r.get('vgpr','?'):>4} agpr {r.get('agpr','?'):>3} sgpr {r.get('sgpr','?'):>4} scratch {r.get('scratch','?'):>5} lds {r.get('lds','?'):>6} occ {r.get('occ','?')}")

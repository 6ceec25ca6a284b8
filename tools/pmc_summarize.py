#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc FETCH_SIZE run: avg HBM read bytes per dispatch per kernel (gfx950: KiB units, x2).

    python tools/pmc_summarize.py <dir with *counter_collection.csv> <out.json> "<note>"
"""
import csv, glob, json, sys, collections
d, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != "FETCH_SIZE":
        continue
    a = acc[r["Kernel_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
res = {"note": note, "kernels": {}}
for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    res["kernels"][k] = {"dispatches": n, "avg_FETCH_SIZE_KiB": round(tot / n, 1),
                         "avg_hbm_read_bytes_corrected": int(2 * 1024 * tot / n)}
json.dump(res, open(out, "w"), indent=1)
for k, v in list(res["kernels"].items())[:12]:
    print(k[:70].ljust(70), v)

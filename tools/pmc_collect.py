#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run: per kernel, dispatch count and the average of every collected counter.

    python tools/pmc_collect.py <counter_collection.csv> <out.json>
FETCH_SIZE is in KiB and on gfx950 reports half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), so
avg_hbm_read_bytes_corrected = 2 * 1024 * FETCH_SIZE is added when that counter is present."""
import collections, csv, json, sys
src, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(src)):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
res = {"kernels": {}}
for k, cs in acc.items():
    n = max(v[0] for v in cs.values())
    row = {"dispatches": n}
    for c, (cnt, tot) in cs.items():
        row["avg_" + c] = round(tot / cnt, 1)
    if "FETCH_SIZE" in cs:
        row["avg_hbm_read_bytes_corrected"] = int(2 * 1024 * cs["FETCH_SIZE"][1] / cs["FETCH_SIZE"][0])
    res["kernels"][k] = row
res["kernels"] = dict(sorted(res["kernels"].items(), key=lambda kv: -kv[1]["dispatches"]))
json.dump(res, open(out, "w"), indent=1)
for k, v in list(res["kernels"].items())[:14]:
    print(k[:64].ljust(64), v)

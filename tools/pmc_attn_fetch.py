#!/usr/bin/env python3
"""Summarise `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 tools/bench_kernels.py attn` into the JSON bench.py
reads for `roofline.traffic` (HBM bytes per algorithmic byte of attn_partial_kernel):  python tools/pmc_attn_fetch.py DIR OUT.json
bench_kernels.py attn walks positions 255 / 1023 / 2679 / 5239 with the same number of launches each (Bp 32, H 20, hd 64, bf16); dispatches are
grouped in order.  FETCH_SIZE counts KiB; gfx950 reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section): x2."""
import csv
import glob
import json
import os
import sys

d, out = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f, newline="")):
        if "attn_partial_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"]))
rows.sort()
pos = [255, 1023, 2679, 5239]
n = len(rows) // len(pos)
assert n > 0 and n * len(pos) == len(rows), (len(rows), "dispatches do not split into %d equal groups" % len(pos))
Bp, H, hd = 32, 20, 64
res = []
for i, p in enumerate(pos):
    grp = rows[i * n:(i + 1) * n]
    raw = sum(v for _, v, _ in grp) / n * 1024.0
    alg = 2.0 * Bp * H * hd * (p + 1) * 2
    res.append({"pos": p, "launches": n, "fetch_size_bytes_raw": raw, "hbm_read_bytes_corrected": 2 * raw, "algorithmic_bytes": alg,
                "traffic_over_algorithmic": 2 * raw / alg})
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 tools/bench_kernels.py attn (MI355X, ROCm 7.2, round 4)",
           "kernel": rows[0][2] + "  Bp=32 H=20 hd=64 S=5240",
           "unit": "FETCH_SIZE is reported in KiB; gfx950 correction x2 for wide coalesced streaming reads (MI355X_MICROARCH.md HBM section)",
           "rows": res}, open(out, "w"), indent=1)
print(json.dumps(res))

#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc output (counter_collection.csv of one or more passes) per kernel: mean counter value per dispatch.
    python tools/pmc_summarize.py OUT.json KERNEL_SUBSTRING[,...] DIR [DIR ...]
Kernel names are shortened to the part before the first '(' with the vlg:: template arguments kept."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    i = name.find("(")
    return name[:i] if i > 0 else name


def main():
    out, subs, dirs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for row in csv.DictReader(open(f, newline="")):
                k = short(row["Kernel_Name"])
                if not any(s in k for s in subs):
                    continue
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                key = (row["Dispatch_Id"], f)
                if key not in seen:
                    seen.add(key)
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    res = {}
    for k, cs in acc.items():
        res[k] = {"dispatches_per_pass": {c: len(v) for c, v in cs.items()}, "mean_per_dispatch": {c: sum(v) / len(v) for c, v in cs.items()},
                  "mean_duration_us_under_pmc": sum(dur[k]) / max(1, len(dur[k]))}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(re.sub(r"\s+", " ", k)[:150])
        for c, x in sorted(v["mean_per_dispatch"].items()):
            print("    %-32s %16.1f  (n=%d)" % (c, x, v["dispatches_per_pass"][c]))


if __name__ == "__main__":
    main()

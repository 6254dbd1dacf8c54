#!/usr/bin/env python3
"""Prints the top rows of a rocprofv3 --stats kernel_stats.csv with shortened kernel names: python tools/kstats_top.py FILE [n]"""
import csv
import re
import sys

csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*", "", name)[:90]
    print("%-92s calls %8s  avg %10.1f ns  total %8.1f ms  %5s%%" % (name, r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))

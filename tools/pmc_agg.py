#!/usr/bin/env python3
"""Sums the counters of a `rocprofv3 --pmc ... --output-format csv -d DIR` run per kernel: pmc_agg.py DIR [kernel-substring].
A kernel is named by its full name without the argument list; `dispatches` counts its launches in the run, so
per-launch figures are the sums divided by it."""
import collections
import csv
import glob
import json
import re
import sys


def kernel_key(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"^(.*?)\((?!anonymous)", name)
    return (m.group(1) if m else name).strip()


out = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = kernel_key(r["Kernel_Name"])
        out[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
res = {}
for k, v in out.items():
    if len(sys.argv) < 3 or sys.argv[2] in k:
        res[k] = dict(v)
        res[k]["dispatches"] = len(seen[k])
print(json.dumps(res, indent=1))

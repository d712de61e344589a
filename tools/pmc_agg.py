#!/usr/bin/env python3
"""Sums the counters of a `rocprofv3 --pmc ... --output-format csv -d DIR` run per kernel: pmc_agg.py DIR [kernel-substring]."""
import csv, glob, json, sys, collections
out = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        out[k][r['Counter_Name']] += float(r['Counter_Value'])
print(json.dumps({k: dict(v) for k, v in out.items() if len(sys.argv) < 3 or sys.argv[2] in k}, indent=1))

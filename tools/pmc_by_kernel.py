#!/usr/bin/env python3
"""Average of each PMC counter per kernel name from a rocprofv3 --pmc run: python3 tools/pmc_by_kernel.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()})

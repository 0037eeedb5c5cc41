#!/usr/bin/env python3
"""FETCH_SIZE (and any other counter) of the first launches of the last recurrent pass: python3 tools/pmc_head.py <dir>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lstm_w" in r["Kernel_Name"] or "lstm_thin" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:40], r["Counter_Name"], float(r["Counter_Value"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)))
rows.sort()
# last pass = last 616 dispatches
ids = sorted({r[0] for r in rows})
last = set(ids[-616:])
sel = [r for r in rows if r[0] in last]
for i in list(range(0, 12)) + list(range(250, 256)) + list(range(300, 304)) + list(range(400, 403)) + list(range(600, 603)):
    if i < len(sel):
        r = sel[i]
        print(i, r[1], "blocks", r[4], r[2], r[3], "-> MB %.1f" % (r[3] * 2 * 1024 / 1e6))

#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run: for every kernel whose name contains <pattern>, the gap
between the end of the previous kernel on the device and its start.  usage: tools/trace_gaps.py <trace dir> <pattern>"""
import csv
import glob
import sys

import numpy as np

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
pat = sys.argv[2]
gaps, durs = [], []
for (s0, e0, n0), (s1, e1, n1) in zip(rows[:-1], rows[1:]):
    if pat in n0 and pat in n1:
        gaps.append(s1 - e0)
        durs.append(e1 - s1)
gaps, durs = np.array(gaps) / 1e3, np.array(durs) / 1e3
keep = gaps < 1000.0   # (the pauses between passes are not launch gaps)
print("kernels %d  busy %.2f ms  gaps %.2f ms  (mean %.2f us, median %.2f, p90 %.2f, max %.1f)" % (
    keep.sum(), durs[keep].sum() / 1e3, gaps[keep].sum() / 1e3, gaps[keep].mean(), np.median(gaps[keep]), np.percentile(gaps[keep], 90), gaps[keep].max()))
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1000)):
    sel = keep & (gaps >= lo) & (gaps < hi)
    print("  gap %3d-%4d us: %5d launches, %.2f ms; mean duration of the kernel behind it %.1f us" % (lo, hi, sel.sum(), gaps[sel].sum() / 1e3, durs[sel].mean() if sel.any() else 0.0))

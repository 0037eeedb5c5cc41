#!/usr/bin/env python3
"""Kernel statistics table (the columns of rocprofv3 --stats) from a plain --kernel-trace run; rocprofv3's own --stats pass
crashed on the two-phase cells' long launch sequences.  usage: tools/trace_stats.py <trace dir> <out csv>"""
import csv
import glob
import statistics
import sys

rows = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
total = sum(sum(v) for v in rows.values())
with open(sys.argv[2], "w", newline="") as fo:
    w = csv.writer(fo, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(v), sum(v), sum(v) / len(v), round(100.0 * sum(v) / total, 2), min(v), max(v), statistics.pstdev(v) if len(v) > 1 else 0.0])

#!/usr/bin/env python3
"""Per-launch picture of the recurrent path from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lstm_trace -- python3 tools/bench_configs.py --which lstm --steps 2
   python3 tools/lstm_trace.py gpurun_out/lstm_trace
Prints, for pass argv[2] (default: the last), the kernel time and the gap to the previous kernel per group of 32 wavefront launches."""
import csv
import glob
import sys


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 1) or 1)))
    rows.sort()
    # passes: maximal runs of lstm_wavefront launches
    runs, cur = [], []
    for i, r in enumerate(rows):
        if "lstm_w" in r[2] or "lstm_thin" in r[2] or "rec_step" in r[2]:
            cur.append(i)
        elif cur:
            runs.append(cur)
            cur = []
    if cur:
        runs.append(cur)
    print("passes:", len(runs), "launches per pass:", [len(r) for r in runs])
    run = runs[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
    tot_k = sum(rows[i][1] - rows[i][0] for i in run)
    tot_g = sum(rows[i][0] - rows[i - 1][1] for i in run[1:])
    print("last pass: kernel %.3f ms, gaps %.3f ms, span %.3f ms" % (tot_k / 1e6, tot_g / 1e6, (rows[run[-1]][1] - rows[run[0]][0]) / 1e6))
    step = 32
    for s in range(0, len(run), step):
        grp = run[s:s + step]
        k = sum(rows[i][1] - rows[i][0] for i in grp) / len(grp)
        g = sum(rows[i][0] - rows[i - 1][1] for i in grp) / len(grp)
        blocks = sum(rows[i][3] // rows[i][4] for i in grp) / len(grp)
        print("launch %4d..%4d  blocks %6.1f  kernel %6.2f us  gap %5.2f us" % (s, s + len(grp) - 1, blocks, k / 1e3, g / 1e3))
    # the other kernels of the pass window
    lo, hi = rows[run[0]][0] - 3_000_000, rows[run[-1]][1] + 3_000_000
    other = {}
    for r in rows:
        if lo <= r[0] <= hi and not ("lstm_w" in r[2] or "lstm_thin" in r[2] or "rec_step" in r[2]):
            other.setdefault(r[2][:60], []).append(r[1] - r[0])
    for k, v in other.items():
        print("  near the pass: %-60s n=%d total %.3f ms" % (k, len(v), sum(v) / 1e6))


if __name__ == "__main__":
    main()

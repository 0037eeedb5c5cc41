#!/usr/bin/env python3
"""Condense a tools/prof.sh output directory (gpurun_out/prof_<tag>/) into the committed summaries under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table, verbatim
  profiles/<tag>_pmc_summary.json   per kernel (name + grid): launches, average duration and counter averages, plus
                                    derived figures following /opt/skills/guides/MI355X_MICROARCH.md:
                                      clock_ghz   = GRBM_GUI_ACTIVE / 8 XCDs / duration
                                      mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
                                      hbm_bytes   = (2 x FETCH_SIZE + WRITE_SIZE) x 1024   (gfx950: FETCH_SIZE counts half of
                                                    a 16-B-per-lane streaming read; WRITE_SIZE is exact; separate --pmc passes)
usage: tools/summarise_profiles.py gpurun_out/prof_r1b r1b
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copyfile(stats[0], os.path.join(out_dir, tag + "_kernel_stats.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0] + " grid=" + r["Grid_Size"]
                agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for k, counters in agg.items():
        o = {}
        for c, vals in counters.items():
            o[c] = {"launches": len(vals), "avg": sum(v[0] for v in vals) / len(vals), "avg_dur_us": sum(v[1] for v in vals) / len(vals) / 1e3}
        if "GRBM_GUI_ACTIVE" in o:
            cyc = o["GRBM_GUI_ACTIVE"]["avg"] / 8.0
            o["derived_clock_ghz"] = cyc / (o["GRBM_GUI_ACTIVE"]["avg_dur_us"] * 1e3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in o and cyc > 0:
                o["derived_mfma_util"] = o["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (1024.0 * cyc)
        if "FETCH_SIZE" in o and "WRITE_SIZE" in o:
            o["derived_hbm_bytes_per_launch"] = (2.0 * o["FETCH_SIZE"]["avg"] + o["WRITE_SIZE"]["avg"]) * 1024.0
        out[k] = o
    json.dump(out, open(os.path.join(out_dir, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    for k in sorted(out):
        o = out[k]
        print("%-60s clock %.3f GHz  mfma_util %s  hbm %s MB" % (k[:60], o.get("derived_clock_ghz", float("nan")),
              "%.3f" % o["derived_mfma_util"] if "derived_mfma_util" in o else "-",
              "%.0f" % (o["derived_hbm_bytes_per_launch"] / 1e6) if "derived_hbm_bytes_per_launch" in o else "-"))


if __name__ == "__main__":
    main()

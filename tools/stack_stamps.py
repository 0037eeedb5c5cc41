#!/usr/bin/env python3
"""Timeline of the single-launch Linear stack (gemm_stack_kernel; run with AM_FF_STAMPS=1, set below): per workgroup its
start / end, tiles computed, time spent in dependency waits and tiles taken from other XCDs' queues.  Diagnostic only."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AM_FF_STAMPS", "1")
os.environ.setdefault("AM_FF_STACK", "1")
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline  # noqa: E402
from nnacousticmodeling_amd import synthetic as syn  # noqa: E402
from nnacousticmodeling_amd._lib import check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=57919)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--units", type=int, default=1024)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((a.frames, 40)).astype(np.float32)
    eng = Engine(0)
    m = Model(eng, "ff", a.layers, syn.make_mlp_params(rng, 440, a.units, a.layers, 1909))
    p = Pipeline(eng)
    p.set_frontend(40, list(range(-5, 6)))
    p.set_members([m])
    p.ff_upload(x, None, a.frames, 0, 0)
    for _ in range(3):
        p.ff_run()
    eng.synchronize()
    st = np.zeros((256, 16), dtype=np.uint64)
    check(p._lib.am_ff_stack_stamps(p._h, st.ctypes.data_as(C.c_void_p)))
    st = st.astype(np.int64)
    live = st[:, 1] > 0
    t0 = st[live, 0].min()
    start, end, last = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, (st[:, 6] - t0) / 100.0
    print("workgroups %d  span %.1f us  starts within %.1f us" % (live.sum(), end[live].max(), start[live].max()))
    print("end of workgroups: min %.1f  p25 %.1f  median %.1f  p75 %.1f  max %.1f us" % tuple(np.percentile(end[live], [0, 25, 50, 75, 100])))
    print("start of the last tile: min %.1f median %.1f max %.1f us" % tuple(np.percentile(last[live], [0, 50, 100])))
    print("tiles per workgroup: min %d median %d max %d; total %d" % (st[live, 2].min(), np.median(st[live, 2]), st[live, 2].max(), st[live, 2].sum()))
    print("dependency waits: %d in all, %.1f us summed over workgroups (max %.1f us in one)" % (st[live, 4].sum(), st[live, 3].sum() / 100.0, st[live, 3].max() / 100.0))
    print("tiles taken from another XCD's queue: %d" % st[live, 5].sum())
    idle = (end[live].max() - end[live]).sum() / live.sum()
    print("mean idle tail per workgroup: %.1f us (%.1f %% of the span)" % (idle, 100.0 * idle / end[live].max()))
    ph = st[live, 8:12].sum(axis=0) / 100.0 / max(st[live, 2].sum(), 1)
    print("per tile (mean over all tiles): pull + dependency check %.2f us, cold start %.2f us, product loop %.2f us, epilogue %.2f us" % tuple(ph))
    for x_ in range(8):
        sel = live & (st[:, 7] == x_)
        if sel.any():
            print("  XCC %d: %3d workgroups, tiles %5d, end median %.1f max %.1f us, stolen %d" % (x_, sel.sum(), st[sel, 2].sum(), np.median(end[sel]), end[sel].max(), st[sel, 5].sum()))


if __name__ == "__main__":
    main()

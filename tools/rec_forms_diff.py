#!/usr/bin/env python3
"""Persistent cell kernel: whole phases (AM_REC_SPLIT=0) against split phases (AM_REC_SPLIT=2) on one input -- where the outputs differ, and by how much.
Diagnostic only.  usage: rec_forms_diff.py [net] [layers units n_utts]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AM_REC_STREAM"] = "2"
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline  # noqa: E402
from nnacousticmodeling_amd import synthetic as syn  # noqa: E402


def main():
    net = sys.argv[1] if len(sys.argv) > 1 else "gru"
    layers, units, n_utts = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3, 512, 13)
    rng = np.random.default_rng(7)
    n = 60 * n_utts
    off = syn.make_offsets(rng, n_utts, n, lo=5, hi=150)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    eng = Engine(0)
    if net == "peepholelstm":
        params = syn.make_peephole_params(rng, 40, units, layers, 1909)
    else:
        params = syn.make_gru_params(rng, 40, units, layers, 1909, reset_gate=net != "mgrurelu")
    p = Pipeline(eng)
    p.set_frontend(40, [0])
    p.set_members([Model(eng, net, layers, params)])
    outs = {}
    for sp in ("0", "2", "2"):
        os.environ["AM_REC_SPLIT"] = sp
        outs.setdefault(sp, []).append(p.forward_lstm(x, off))
    a, b, b2 = outs["0"][0], outs["2"][0], outs["2"][1]
    print("split run twice equal:", np.array_equal(b, b2))
    d = np.abs(a - b).max(axis=1)
    bad = np.where(d > 0)[0]
    print("rows", n, "differing rows", len(bad), "max |diff|", float(d.max()) if len(d) else 0.0)
    lens = np.diff(off)
    order = np.argsort(-lens, kind="stable")
    for u in range(min(n_utts, 8)):
        du = d[off[u]:off[u + 1]]
        nz = np.where(du > 0)[0]
        print("utt %d (len %d, slot %d): first differing frame %s, max %.3g" % (u, lens[u], int(np.where(order == u)[0][0]), nz[0] if len(nz) else None, du.max()))


if __name__ == "__main__":
    main()

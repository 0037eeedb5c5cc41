#!/usr/bin/env python3
"""Timeline of the persistent cell kernel (rec_stream_kernel; run with AM_LSTM_STAMPS=1, AM_REC_STREAM=2): per role and time step,
where the step's time goes (polls, products, element-wise part, publish), summarised by the number of active utterances.
Diagnostic only.  usage: rec_stamps.py [gru|mgrurelu|mgrurelur|peepholelstm] [layers units]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AM_LSTM_STAMPS", "1")
os.environ.setdefault("AM_REC_STREAM", "2")
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline  # noqa: E402
from nnacousticmodeling_amd import synthetic as syn  # noqa: E402
from nnacousticmodeling_amd._lib import check  # noqa: E402


def main():
    net = sys.argv[1] if len(sys.argv) > 1 else "mgrurelu"
    layers, units = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 512)
    rng = np.random.default_rng(20181806)
    N = 57919
    off = syn.make_offsets(rng, 192, N)
    x = rng.standard_normal((N, 40)).astype(np.float32)
    eng = Engine(0)
    if net == "peepholelstm":
        params = syn.make_peephole_params(rng, 40, units, layers, 1909)
    else:
        params = syn.make_gru_params(rng, 40, units, layers, 1909, reset_gate=net != "mgrurelu")
    m = Model(eng, net, layers, params)
    p = Pipeline(eng)
    p.set_frontend(40, [0])
    p.set_members([m])
    p.lstm_upload(x, None, off, 0)
    for _ in range(3):
        p.lstm_run()
    eng.synchronize()
    ng, T = C.c_int(), C.c_int()
    check(p._lib.am_lstm_stamps(p._h, None, 0, C.byref(ng), C.byref(T)))
    raw = np.zeros(ng.value * T.value * 8 + 64 * ng.value, dtype=np.uint64)
    check(p._lib.am_lstm_stamps(p._h, raw.ctypes.data_as(C.c_void_p), raw.size, C.byref(ng), C.byref(T)))
    st = raw[:ng.value * T.value * 8].reshape(ng.value, T.value, 8).astype(np.int64)
    trace = raw[ng.value * T.value * 8:].reshape(ng.value, 16, 4).astype(np.int64)
    if int(os.environ.get("AM_REC_DBG", "0")) & 128:   # lane 0's first accumulator quad per role and step (split phases: the input part stores it, the state part loads it)
        f = raw[:ng.value * T.value * 8].reshape(ng.value, T.value, 8)[:, :, 2:4].copy().view(np.float32).reshape(ng.value, T.value, 4)
        for r in range(ng.value):
            print("role", r, "steps 0..2:", f[r, :3].tolist())
        for r in range(1, ng.value - 1):
            neq = np.where((f[r] != f[r + 1]).any(axis=1))[0]
            print("roles %d / %d: steps where the quads differ: %d of %d, first %s" % (r, r + 1, len(neq), T.value, neq[:8].tolist()))
        y0 = p.lstm_download()
        for k in range(3):   # launch again: do the stored quads, and the outputs, repeat?
            p.lstm_run()
            eng.synchronize()
            raw2 = np.zeros_like(raw)
            check(p._lib.am_lstm_stamps(p._h, raw2.ctypes.data_as(C.c_void_p), raw2.size, C.byref(ng), C.byref(T)))
            f2 = raw2[:ng.value * T.value * 8].reshape(ng.value, T.value, 8)[:, :, 2:4].copy().view(np.float32).reshape(ng.value, T.value, 4)
            y = p.lstm_download()
            dy = np.abs(y - y0).max(axis=1)
            print("run %d: roles whose quads changed: %s ; output rows that changed: %d (first frame index %s)" % (
                k, [(r, int((f2[r] != f[r]).any(axis=1).sum())) for r in range(ng.value)], int((dy > 0).sum()), np.where(dy > 0)[0][:5].tolist()))
        return
    if trace.any():   # AM_REC_DBG = step << 8: the batch loop of that step, per role: stage wait, barrier, request of the stage ahead, (products)
        for r in range(ng.value):
            tr = trace[r]
            nb = int((tr[:, 0] > 0).sum())
            if nb == 0:
                continue
            print("role %d, step %d: batch loop, us per batch [wait for the stage | barrier | poll + request | products .. next batch]" % (r, int(os.environ.get("AM_REC_DBG", "0")) >> 8))
            for b in range(nb):
                nxt = tr[b + 1, 0] if b + 1 < nb else 0
                print("   batch %2d: %.2f | %.2f | %.2f | %s" % (b, (tr[b, 1] - tr[b, 0]) / 100.0, (tr[b, 2] - tr[b, 1]) / 100.0, (tr[b, 3] - tr[b, 2]) / 100.0,
                                                             ("%.2f" % ((nxt - tr[b, 3]) / 100.0)) if nxt else "-"))
    lens = np.diff(off)
    nact = np.array([(lens > t).sum() for t in range(T.value)])
    abs_ = st[:, :, [0, 4, 5, 6]]
    print(net, "roles", ng.value, "steps", T.value, "span of the launch: %.1f us" % ((abs_[abs_ > 0].max() - abs_[abs_ > 0].min()) / 100.0))
    # slots: 0 step start, 1 / 2 / 3 time polling (x rows / previous-step rows / other), 4 products done, 5 stores issued, 6 published,
    # 7 end of the poll for the previous step's rows
    for r in range(ng.value):
        s = st[r]
        live = s[:, 6] > 0
        if live.sum() < 3:
            continue
        print("role %d: %d steps, period mean %.2f us" % (r, live.sum(), np.diff(s[:, 6][live]).mean() / 100.0))
        for lo, hi in ((161, 192), (113, 160), (65, 112), (33, 64), (17, 32), (1, 16)):
            sel = live & (nact >= lo) & (nact <= hi)
            sel[0] = False
            if sel.sum() < 2:
                continue
            i = np.where(sel)[0]
            per = (s[i, 6] - s[i - 1, 6]) / 100.0
            tot = (s[i, 6] - s[i, 0]) / 100.0
            poll_x, poll_h, poll_o = s[i, 1] / 100.0, s[i, 2] / 100.0, s[i, 3] / 100.0
            prod = (s[i, 4] - s[i, 0]) / 100.0 - poll_x - poll_h
            epi = (s[i, 5] - s[i, 4]) / 100.0 - poll_o
            pub = (s[i, 6] - s[i, 5]) / 100.0
            tail = np.where(s[i, 7] > 0, (s[i, 6] - s[i, 7]) / 100.0, np.nan)   # last lateral poll satisfied -> this step published
            print("  n_active %3d-%3d (%3d steps): period %.2f  in-step %.2f = poll x %.2f + poll h %.2f + poll other %.2f + products %.2f + element-wise %.2f + publish %.2f ;"
                  " lateral-ready -> published %.2f us" % (lo, hi, len(i), per.mean(), tot.mean(), poll_x.mean(), poll_h.mean(), poll_o.mean(), prod.mean(), epi.mean(),
                                                           pub.mean(), np.nanmean(tail)))


if __name__ == "__main__":
    main()

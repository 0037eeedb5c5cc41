#!/usr/bin/env python3
"""Timeline of the persistent LSTM kernel (run with AM_LSTM_STAMPS=1): per row group and time step, where the step's time
goes (waits, the two products, gates, publish), summarised by the number of active utterances.  Diagnostic only."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AM_LSTM_STAMPS", "1")
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline  # noqa: E402
from nnacousticmodeling_amd import synthetic as syn  # noqa: E402
from nnacousticmodeling_amd._lib import check  # noqa: E402


def main():
    rng = np.random.default_rng(20181806)
    N = 57919
    off = syn.make_offsets(rng, 192, N)
    x = rng.standard_normal((N, 40)).astype(np.float32)
    eng = Engine(0)
    layers, units = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 512)   # (4 1024: the streamed form, one stamp row per layer)
    m = Model(eng, "lstm", layers, syn.make_lstm_params(rng, 40, units, layers, 1909))
    p = Pipeline(eng)
    p.set_frontend(40, [0])
    p.set_members([m])
    p.lstm_upload(x, None, off, 0)
    for _ in range(3):
        p.lstm_run()
    eng.synchronize()
    ng, T = C.c_int(), C.c_int()
    check(p._lib.am_lstm_stamps(p._h, None, 0, C.byref(ng), C.byref(T)))
    st = np.zeros((ng.value, T.value, 8), dtype=np.uint64)
    check(p._lib.am_lstm_stamps(p._h, st.ctypes.data_as(C.c_void_p), st.size, C.byref(ng), C.byref(T)))
    st = st.astype(np.int64)
    lens = np.diff(off)
    nact = np.array([(lens > t).sum() for t in range(T.value)])
    # stamp slots: 0 step start, 1 upward wait done, 2 upward product done, 4 lateral wait + product done, 5 stores issued, 6 published
    # (layer 0 has no upward product: 0, 1, 4, 5, 6)
    print("groups", ng.value, "steps", T.value, "span of the launch: %.1f us" % ((st[st > 0].max() - st[st > 0].min()) / 100.0))
    for g in range(ng.value):
        s = st[g]
        live = s[:, 6] > 0
        if not live.any():
            continue
        dur = np.diff(s[:, 6][live])   # publish-to-publish = step period
        print("group %d: %d steps, period mean %.2f us" % (g, live.sum(), dur.mean() / 100.0))
        for lo, hi in ((161, 192), (113, 160), (65, 112), (49, 64), (33, 48), (17, 32), (1, 16)):
            sel = live & (nact >= lo) & (nact <= hi)
            sel[0] = False
            if sel.sum() < 2:
                continue
            idx = np.where(sel)[0]
            per = (s[idx, 6] - s[idx - 1, 6]) / 100.0
            used = [k for k in range(7) if (s[idx, k] > 0).any()]   # the stamp slots this kernel form fills
            seg = []
            for ka, kb in zip(used[:-1], used[1:]):
                a, b = s[idx, ka], s[idx, kb]
                ok = (a > 0) & (b > 0)
                seg.append("%d>%d %.2f" % (ka, kb, (b - a)[ok].mean() / 100.0))
            print("   rows %3d-%3d: %3d steps, period %.2f us | " % (lo, hi, len(idx), per.mean()) + ", ".join(seg))
    p.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md: the other BASELINE.json configs on one MI355X (device-resident, like bench.py).

  configs[1'] 4x1024 on N=57,919 (core-test-like, 192 utterances)     configs[2] 6x2048 + 100 i-vector columns
  configs[3]  3x512 LSTM (uni-directional), timedelay 0/5  configs[4] 10-member ensemble / dev-average
Prints one JSON object per line.  Not the headline benchmark (that is bench.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnacousticmodeling_amd._lib import AM_COMBINE_LOGITS, AM_COMBINE_LOGPOST  # noqa: E402
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline  # noqa: E402
from nnacousticmodeling_amd import synthetic as orc  # noqa: E402  (synthetic inputs; no oracle in measurement scripts)
from nnacousticmodeling_amd.kaldi_ft import loadKaldiFeatureTransform, select_centre_frame, tile_centre_frame, tdnn_splice  # noqa: E402

SEED = 20181806


def timed(eng, fn, steps, warmup):
    for _ in range(warmup):
        fn()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    eng.synchronize()
    timed.unprofiled_ms = (time.perf_counter() - t0) / steps * 1e3
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    eng.synchronize()
    wall = (time.perf_counter() - t0) / steps
    prof = eng.profile_read()
    eng.profile(False)
    return wall, {k: round(v[1] / steps, 4) for k, v in prof.items()}, {k: v[0] // steps for k, v in prof.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="all")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--precision", default="f32", choices=["f32", "split_f16"])
    ap.add_argument("--cell", default="", help="with --which cells: comma-separated subset of gru,mgrurelu,mgrurelur,peepholelstm")
    ap.add_argument("--shape", default="3x512", help="with --which cells: layers x units (4x1024: the other specom2018 recipe shape)")
    args = ap.parse_args()
    rng = np.random.default_rng(SEED)
    golden = os.path.join(ROOT, "tests", "golden")
    ft = loadKaldiFeatureTransform(os.path.join(golden, "final.feature_transform"))
    ftc = select_centre_frame(loadKaldiFeatureTransform(os.path.join(golden, "final.feature_transform")))
    eng = Engine(0)
    eng.set_precision(args.precision)
    N = 57919
    offsets = orc.make_offsets(rng, 192, N)
    x = rng.standard_normal((N, 40)).astype(np.float32)

    def report(name, frames, wall, ms, calls, flops_per_frame):
        un = timed.unprofiled_ms
        print(json.dumps({"config": name, "precision": args.precision, "frames": frames, "ms_per_pass_unprofiled": round(un, 3), "frames_per_s_unprofiled": round(frames / un * 1e3),
                          "ms_per_pass": round(wall * 1e3, 3), "frames_per_s": round(frames / wall),
                          "tflops": round(frames * flops_per_frame / wall / 1e12, 2), "kernels_ms": ms, "launches": calls}), flush=True)

    if args.which in ("all", "ff"):
        for name, layers, units, d_iv, n in (("4x1024 N=57919", 4, 1024, 0, N), ("6x2048+ivec N=57919", 6, 2048, 100, N), ("6x2048+ivec N=196608", 6, 2048, 100, 196608)):
            params = orc.make_mlp_params(rng, 440 + d_iv, units, layers, 1909)
            xx = x if n == N else rng.standard_normal((n, 40)).astype(np.float32)
            iv = rng.standard_normal((n, d_iv)).astype(np.float32) if d_iv else None
            p = Pipeline(eng)
            p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"], d_iv)
            p.set_members([Model(eng, "ff", layers, params)])
            p.ff_upload(xx, iv, n, 0, 0)
            wall, ms, calls = timed(eng, p.ff_run, args.steps, 2)
            fl = 2 * ((440 + d_iv) * units + (layers - 1) * units * units + units * 1909)
            report(name, n, wall, ms, calls, fl)
            p.close()
    if args.which in ("all", "e2e"):
        # host-to-host (PCIe-inclusive) rate of the headline model: pageable vs pinned output, one-shot call
        params = orc.make_mlp_params(rng, 440, 1024, 4, 1909)
        n = 196608
        xx = rng.standard_normal((n, 40)).astype(np.float32)
        p = Pipeline(eng)
        p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        p.set_members([Model(eng, "ff", 4, params)])
        for label, out in (("pageable", np.empty((n, 1909), np.float32)), ("pinned", eng.pinned_empty((n, 1909)))):
            p.forward_ff(xx, out=out)
            t0 = time.perf_counter()
            for _ in range(3):
                p.forward_ff(xx, out=out)
            dt = (time.perf_counter() - t0) / 3
            print(json.dumps({"config": "4x1024 N=196608 host-to-host (%s output)" % label, "ms_per_pass": round(dt * 1e3, 2), "frames_per_s": round(n / dt)}), flush=True)
        p.close()
    if args.which in ("all", "lstm"):
        params = orc.make_lstm_params(rng, 40, 512, 3, 1909)
        m = Model(eng, "lstm", 3, params)
        for td in (0, 5):
            p = Pipeline(eng)
            p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
            p.set_members([m])
            p.lstm_upload(x, None, offsets, td)
            wall, ms, calls = timed(eng, p.lstm_run, args.steps, 3)  # eager, graph capture, first replay
            fl = 2 * (40 * 2048 + 2 * 512 * 2048 + 3 * 512 * 2048 + 512 * 1909)
            report("3x512 LSTM N=57919 192 utts Tmax=%d timedelay=%d" % (int(np.diff(offsets).max()), td), N, wall, ms, calls, fl)
            p.close()
    if args.which in ("lstm256",):
        params = orc.make_lstm_params(rng, 40, 256, 3, 1909)
        m = Model(eng, "lstm", 3, params)
        p = Pipeline(eng)
        p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
        p.set_members([m])
        p.lstm_upload(x, None, offsets, 0)
        wall, ms, calls = timed(eng, p.lstm_run, args.steps, 3)
        report("3x256 LSTM N=57919 192 utts", N, wall, ms, calls, 2 * (40 * 1024 + 2 * 256 * 1024 + 3 * 256 * 1024 + 256 * 1909))
        p.close()
    if args.which in ("lstmshapes",):
        # the LSTM shapes of the reference's recipes (scripts/papers/tsd2018/run.sh:33-41: 2..6 layers x 256 / 512; specom2018: 4 x 1024), --timedelay 5
        for layers, units in ((2, 256), (6, 256), (2, 512), (3, 512), (4, 512), (5, 512), (4, 1024)):
            params = orc.make_lstm_params(rng, 40, units, layers, 1909)
            m = Model(eng, "lstm", layers, params)
            p = Pipeline(eng)
            p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
            p.set_members([m])
            p.lstm_upload(x, None, offsets, 5)
            wall, ms, calls = timed(eng, p.lstm_run, args.steps, 3)
            fl = 2 * (40 * 4 * units + layers * units * 4 * units + (layers - 1) * units * 4 * units + units * 1909)
            report("%dx%d LSTM N=57919 192 utts timedelay=5 (%d of %d members persistent)" % ((layers, units) + p.recurrent_form()), N, wall, ms, calls, fl)
            p.close()
            m.close()
    if args.which in ("lstm1024",):
        # the reference's own example network (scripts/examples: "lstm -l 4 -u 1024 --timedelay 5")
        params = orc.make_lstm_params(rng, 40, 1024, 4, 1909)
        m = Model(eng, "lstm", 4, params)
        p = Pipeline(eng)
        p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
        p.set_members([m])
        p.lstm_upload(x, None, offsets, 5)
        wall, ms, calls = timed(eng, p.lstm_run, args.steps, 3)
        fl = 2 * (40 * 4096 + 4 * 1024 * 4096 + 3 * 1024 * 4096 + 1024 * 1909)
        report("4x1024 LSTM N=57919 192 utts timedelay=5", N, wall, ms, calls, fl)
        p.close()
    if args.which in ("all", "tdnn"):
        # tsd2018's TDNN sweeps (scripts/papers/tsd2018/run.sh:22-30): 5-5-5-5 and 9-9-9-9 kernels, 1024 units
        for ksize in ([5, 5, 5, 5], [9, 9, 9, 9]):
            units = [1024] * 4
            params = orc.make_tdnn_params(rng, 40, units, ksize, 1909)
            tft = tile_centre_frame(loadKaldiFeatureTransform(os.path.join(golden, "final.feature_transform")), tdnn_splice(ksize))
            m = Model(eng, "tdnn", 4, params)
            p = Pipeline(eng)
            p.set_frontend(40, tft["shifts"], tft["addShift"], tft["rescale"])
            p.set_members([m])
            p.ff_upload(x, None, N, 0, 0)
            wall, ms, calls = timed(eng, p.ff_run, args.steps, 1)
            pos, cin, fl = sum(ksize) - 3, 40, 0
            for u, k in zip(units, ksize):
                pos -= k - 1
                fl += 2 * pos * cin * k * u
                cin = u
            fl += 2 * cin * 1909
            report("tdnn %s x1024 N=57919" % "-".join(map(str, ksize)), N, wall, ms, calls, fl)
            p.close()
    if args.which in ("all", "cells"):
        # the other recurrent cells of get_nn at the LSTM config's size
        L, U = (int(v) for v in args.shape.split("x"))
        for net in (args.cell.split(",") if args.cell else ("gru", "mgrurelu", "mgrurelur", "peepholelstm")):
            if net == "peepholelstm":
                params = orc.make_peephole_params(rng, 40, U, L, 1909)
                fl = 2 * (40 * 4 * U + (L - 1) * U * 4 * U + L * U * 4 * U + L * 3 * U * U + U * 1909)
            else:
                g = 2 if net == "mgrurelu" else 3
                params = orc.make_gru_params(rng, 40, U, L, 1909, reset_gate=net != "mgrurelu")
                fl = 2 * (g * (40 * U + (L - 1) * U * U + L * U * U) + U * 1909)
            m = Model(eng, net, L, params)
            p = Pipeline(eng)
            p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
            p.set_members([m])
            p.lstm_upload(x, None, offsets, 0)
            wall, ms, calls = timed(eng, p.lstm_run, args.steps, 3)
            report("%s %s N=57919 192 utts Tmax=%d" % (args.shape, net, int(np.diff(offsets).max())), N, wall, ms, calls, fl)
            p.close()
    if args.which in ("all", "ens"):
        plist = [orc.make_mlp_params(rng, 440, 1024, 4, 1909) for _ in range(10)]
        models = [Model(eng, "ff", 4, q) for q in plist]
        for name, comb, master in (("10-fold dev-average (log-posterior mean)", AM_COMBINE_LOGPOST, False), ("master + 9 folds logit average", AM_COMBINE_LOGITS, True)):
            p = Pipeline(eng)
            p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
            p.set_members(models, has_master=master, combine=comb)
            p.ff_upload(x, None, N, 0, 0)
            wall, ms, calls = timed(eng, p.ff_run, max(2, args.steps // 2), 1)
            report(name + " N=57919", N, wall, ms, calls, 10 * 11102208)
            p.close()


if __name__ == "__main__":
    main()

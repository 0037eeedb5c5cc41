#!/usr/bin/env python3
"""Diagnostic: run one LSTM stack through the persistent form and through the per-step kernels on the same input and report the rows that differ
(any difference is a bug: the forms are bit-equal by construction), with the oracle as referee on the utterances concerned.

  python tools/lstm_forms_diff.py LAYERS UNITS [N_UTTS [N_LONG [BASE_LEN]]]      DBG_EXTRA=k: the N_LONG utterances are k frames longer (a thin tail)

This is the tool that isolated round 4's one-tile miscompare (see stream_pass in csrc/am_lstm_stream.hip): rows 8..15 of every step with <= 16 active
utterances, only with both products in the step -- a chain of dependent v_mfma_f32_16x16x4_f32 whose destination the compiler had not kept in place."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import am_oracle as orc
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline
layers, units = int(sys.argv[1]), int(sys.argv[2])
nutt = int(sys.argv[3]) if len(sys.argv) > 3 else 150
klong = int(sys.argv[4]) if len(sys.argv) > 4 else 3
base = int(sys.argv[5]) if len(sys.argv) > 5 else 60
ft = orc.centre_frame_transform(orc.load_kaldi_feature_transform("tests/golden/final.feature_transform"))
rng = np.random.default_rng(100 * layers + units)
params = orc.make_lstm_params(rng, 40, units, layers, 1909)
lens = np.array([base + int(os.environ.get("DBG_EXTRA", "1"))] * klong + [base] * (nutt - klong))
if os.environ.get("DBG_SHORT"):
    lens[-1] = int(os.environ["DBG_SHORT"])
if not os.environ.get("DBG_NOSHUF"):
    rng.shuffle(lens)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
x = rng.standard_normal((int(off[-1]), 40)).astype(np.float32)
eng = Engine(0)
p = Pipeline(eng); p.set_frontend(40, [0], ft["addShift"], ft["rescale"]); p.set_members([Model(eng, "lstm", layers, params)])
whole = p.forward_lstm(x, off, timedelay=0); print("form", p.recurrent_form())
p.lstm_run(); whole2 = p.lstm_download()
print("second run of the same upload differs from the first in rows:", np.where(np.abs(whole2 - whole).max(axis=1) > 0)[0][:10])
if os.environ.get("DBG_SECOND"): whole = whole2
os.environ["AM_LSTM_PERSIST"] = "0"
steps = p.forward_lstm(x, off, timedelay=0); print("form", p.recurrent_form())
k = min(nutt, 8)
want = orc.predict(orc.LSTM(params, layers), x[:off[k]], off[:k + 1], True, 1, 0, ft)
print("stream vs oracle", np.abs(whole[:off[k]] - want).max(), "steps vs oracle", np.abs(steps[:off[k]] - want).max())
d = np.abs(whole - steps).max(axis=1)
bad = np.where(d > 0)[0]
print("rows differing", len(bad), "of", len(d), "max diff", d.max())
print("bad rows", bad[:20])
order = np.argsort(-lens, kind="stable")
slot_of = {int(u): i for i, u in enumerate(order)}
print("bad (utt, frame, slot):", [(int(np.searchsorted(off, r, side="right") - 1), int(r - off[np.searchsorted(off, r, side="right") - 1]), slot_of[int(np.searchsorted(off, r, side="right") - 1)]) for r in bad[:24]])
for r in bad[:3]:
    u = int(np.searchsorted(off, r, side="right") - 1)
    xo = x[off[u]:off[u + 1]]
    w = orc.predict(orc.LSTM(params, layers), xo, np.array([0, len(xo)]), True, 1, 0, ft)
    print("row", r, "utt", u, "frame", r - off[u], "of", len(xo), "stream err", np.abs(whole[off[u]:off[u+1]] - w).max(), "steps err", np.abs(steps[off[u]:off[u+1]] - w).max(),
          "cols differing", int((whole[r] != steps[r]).sum()))

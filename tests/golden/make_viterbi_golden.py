#!/usr/bin/env python3
"""Golden vectors for the Viterbi decoder (row n1) produced by the REFERENCE decoder itself.

Needs oracle/_ref/libphonerecog_ref.so (``make -C oracle``: the reference's phone_recog.cpp / data.cpp / mlf_io.cpp
compiled where they lie, behind oracle/phonerecog_ref_driver.cpp).  The real decoding graph kaldiTri1909.img is not in the
reference tree (.MISSING_LARGE_BLOBS), so the graphs are synthetic, written in the image format CPhoneRecog::Init reads.
Output: tests/golden/viterbi.npz (graphs, log-probabilities, PIP/LMW and the reference's segmentations).
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import viterbi_oracle as vo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    assert vo.reference_available(), "build oracle/_ref first: make -C oracle"
    rng = np.random.default_rng(20181806)
    out = {}
    # utterances shorter than 3 frames cannot reach a triphone exit: the reference back-trace then never terminates
    # (bestT stays 0), so the vectors use N >= 3
    cases = [(4, 30, [17, 3, 60], -20.0, 1.0), (6, 60, [45, 120, 4, 80], -5.0, 0.5), (7, 1909, [33, 200], -20.0, 1.0), (5, 40, [5, 9], -2.5, 2.0)]
    for ci, (P, S, lens, PIP, LMW) in enumerate(cases):
        g = vo.make_graph(rng, P, S)
        lps = []
        for n in lens:
            z = 3.0 * rng.standard_normal((n, S))
            lps.append((z - np.log(np.sum(np.exp(z), axis=1, keepdims=True))).astype(np.float32))
        with tempfile.TemporaryDirectory() as td:
            img = os.path.join(td, "g.img")
            vo.write_image(img, g)
            g2 = vo.read_image(img)
            assert all(np.array_equal(g[k], g2[k]) for k in g if k not in ("phones", "numStates")) and g2["phones"] == g["phones"]
            ref = vo.reference_recog(img, lps, PIP, LMW)
            raw = open(img, "rb").read()
        for u, lp in enumerate(lps):
            mine = vo.recog_one(g, lp, PIP, LMW)
            mine_named = [(b, e, g["phones"][t]) for b, e, t in mine]
            assert mine_named == ref[u], (ci, u, mine_named[:5], ref[u][:5])
            out["c%d_u%d_lp" % (ci, u)] = lp
            out["c%d_u%d_seg" % (ci, u)] = np.array([(b, e, t) for b, e, t in mine], dtype=np.int32).reshape(-1, 3)
            out["c%d_u%d_phones" % (ci, u)] = np.array([p for _, _, p in ref[u]])
        out["c%d_img" % ci] = np.frombuffer(raw, dtype=np.uint8)
        out["c%d_params" % ci] = np.array([PIP, LMW, len(lens)], dtype=np.float64)
        print("case", ci, "P", P, "T", len(g["biphoneIdxIn"]), "segments", [len(r) for r in ref])
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "viterbi.npz"), **out)
    print("written", os.path.join(HERE, "viterbi.npz"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Real-size decoder golden: the REFERENCE decoder (oracle/_ref, built from
recog_src/source/phone_recog.cpp / data.cpp / mlf_io.cpp where they lie) run on a 48-phone / 2,304-node / ~88k-triphone /
1909-state synthetic decoding graph -- the size of the absent kaldiTri1909.img -- on six utterances.

Only the seed, the utterance lengths, PIP / LMW and the reference's segmentations are stored (a few KB): the test
regenerates the graph and the log-posteriors from the seed with the same numpy Generator calls (`inputs()` below is
imported by tests/test_gpu_viterbi.py), so the 40 MB of inputs never enter the repository.
Output: tests/golden/viterbi_fullsize.npz
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

SEED = 481909
P, S = 48, 1909
LENS = [3, 57, 131, 298, 412, 76, 614, 205]
PARAMS = [(-20.0, 1.0), (-3.0, 0.6)]   # (PIP, LMW): the evaluate defaults (evaluate.py:74-75), and a setting that yields many segments


def inputs(make_graph):
    """(graph, [log-posteriors per utterance]) from SEED; `make_graph` is the graph generator (oracle's or the product's
    synthetic one: tests/test_host_io.py checks they are the same function)."""
    rng = np.random.default_rng(SEED)
    graph = make_graph(rng, P, S)
    lps = []
    for n in LENS:
        z = (2.0 * rng.standard_normal((n, S))).astype(np.float32)
        mx = np.max(z, axis=1, keepdims=True)
        lps.append((z - (mx + np.log(np.sum(np.exp(z - mx), axis=1, keepdims=True)))).astype(np.float32))
    return graph, lps


def main():
    from oracle import viterbi_oracle as vo
    assert vo.reference_available(), "build oracle/_ref first: make -C oracle"
    graph, lps = inputs(vo.make_graph)
    out = {"seed": np.array(SEED), "lens": np.array(LENS), "params": np.array(PARAMS),
           "shape": np.array([P, S, len(graph["lm"]), len(graph["biphoneIdxIn"])])}
    with tempfile.TemporaryDirectory() as td:
        img = os.path.join(td, "g.img")
        vo.write_image(img, graph)
        for pi, (PIP, LMW) in enumerate(PARAMS):
            ref = vo.reference_recog(img, lps, PIP, LMW)
            for u, segs in enumerate(ref):
                out["p%d_u%d_begin_end" % (pi, u)] = np.array([(b, e) for b, e, _ in segs], dtype=np.int32).reshape(-1, 2)
                out["p%d_u%d_phones" % (pi, u)] = np.array([p for _, _, p in segs])
            # the numpy restatement must agree on the short utterances (it needs minutes on the long ones at this graph size)
            for u in (0, 1):
                mine = [(b, e, graph["phones"][t]) for b, e, t in vo.recog_one(graph, lps[u], PIP, LMW)]
                assert mine == ref[u], u
            print("PIP", PIP, "LMW", LMW, "segments per utterance", [len(r) for r in ref])
    np.savez_compressed(os.path.join(HERE, "viterbi_fullsize.npz"), **out)
    print("graph", out["shape"])


if __name__ == "__main__":
    main()

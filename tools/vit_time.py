#!/usr/bin/env python3
"""Device time of the real-size decoder pass (192 utterances / 57,919 frames) by 64-frame segments (AM_VITERBI_TIMING=2), per kernel form
(AM_VITERBI_LDS = 3 stream (default), 2 fused, 1 staged + node kernel, 0 gathers from global memory).  Diagnostic only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AM_VITERBI_TIMING"] = "2"
import numpy as np  # noqa: E402
from nnacousticmodeling_amd.engine import Engine  # noqa: E402
from nnacousticmodeling_amd.phone_recog import PhoneRecog  # noqa: E402
from nnacousticmodeling_amd.synthetic import make_graph, make_offsets  # noqa: E402

rng = np.random.default_rng(481909)
graph = make_graph(rng, 48, 1909)
T_, B_ = len(graph["biphoneIdxIn"]), len(graph["lm"])
n57 = 57919
off = make_offsets(np.random.default_rng(20181806), 192, n57)
z = rng.standard_normal((n57, 1909), dtype=np.float32)
eng = Engine(0)
dec = PhoneRecog(eng, dict(graph, numPhones=48, numBiphones=B_, numTriphones=T_), -20.0, 1.0)
ref = None
for form in (sys.argv[1:] or ["3", "2"]):
    os.environ["AM_VITERBI_LDS"] = form
    for i in range(2):
        t0 = time.perf_counter()
        got = dec.recognise(z, off)
        print("form", form, "pass", i, "host s", round(time.perf_counter() - t0, 4), "device ms", round(eng.decode_elapsed_ms(), 2), flush=True)
    ref = ref or got
    assert got == ref

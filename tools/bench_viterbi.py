#!/usr/bin/env python3
"""Decoder measurement (row n1): GPU Viterbi on a real-size synthetic graph vs the reference decoder on the host CPU.

Graph: 48 phones -> 2,304 biphone nodes, ~88k triphones, 1909 states (the real kaldiTri1909.img is absent from the
reference tree); input: 192 utterances / 57,919 frames of random log-posteriors.  The CPU leg runs the reference's own
C++ (oracle/_ref, single thread like the shipped Linux binary) on a bounded sample of utterances.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnacousticmodeling_amd.engine import Engine  # noqa: E402
from nnacousticmodeling_amd.phone_recog import PhoneRecog  # noqa: E402
from nnacousticmodeling_amd import synthetic as syn  # noqa: E402  (synthetic inputs; the oracle is imported by the CPU-baseline leg only)


def main():
    rng = np.random.default_rng(20181806)
    P, S, N, U = 48, 1909, 57919, 192
    graph = syn.make_graph(rng, P, S)
    T, B = len(graph["biphoneIdxIn"]), len(graph["lm"])
    off = syn.make_offsets(rng, U, N)
    z = (2.0 * rng.standard_normal((N, S))).astype(np.float32)
    lp = (z - (np.max(z, axis=1, keepdims=True) + np.log(np.sum(np.exp(z - np.max(z, axis=1, keepdims=True)), axis=1, keepdims=True)))).astype(np.float32)
    eng = Engine(0)
    dec = PhoneRecog(eng, dict(graph, numPhones=P, numBiphones=B, numTriphones=T), -20.0, 1.0)
    dec.recognise(lp[:off[4]], off[:5])
    t0 = time.perf_counter()
    res = dec.recognise(lp, off)
    gpu_s = time.perf_counter() - t0
    eng.profile(True)
    eng.profile_reset()
    dec.recognise(lp, off)
    prof = eng.profile_read()
    eng.profile(False)
    out = {"graph": {"phones": P, "nodes": B, "triphones": T, "states": S}, "utterances": U, "frames": N,
           "gpu_host_to_result_s": round(gpu_s, 4), "gpu_kernel_ms": round(prof["viterbi"][1], 2),
           "gpu_frames_per_s_kernel": round(N / (prof["viterbi"][1] * 1e-3)), "segments": sum(len(r) for r in res)}
    from oracle import viterbi_oracle as vo   # CPU-baseline leg: the reference decoder built from source (oracle/_ref)
    if vo.reference_available():
        with tempfile.TemporaryDirectory() as td:
            img = os.path.join(td, "g.img")
            vo.write_image(img, graph)
            k = 4
            sample = [lp[off[u]:off[u + 1]] for u in range(k)]
            t0 = time.perf_counter()
            ref = vo.reference_recog(img, sample, -20.0, 1.0)
            cpu_s = time.perf_counter() - t0
        frames = int(off[k])
        out["cpu_reference"] = {"kind": "reference", "cores": 1, "sample": "first %d utterances (%d frames)" % (k, frames),
                                "seconds": round(cpu_s, 3), "frames_per_s": round(frames / cpu_s)}
        out["identical_to_reference_on_sample"] = all(ref[u] == res[u] for u in range(k))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

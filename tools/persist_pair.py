#!/usr/bin/env python3
"""Diagnostic: persistent recurrent launches back to back on one engine, and from two engines of one card at once (the
per-device launch chain of am_api.hip).  Prints wall time per phase and the give-up count.  AM_LSTM_PERSIST selects the form."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnacousticmodeling_amd import _lib, synthetic as syn  # noqa: E402
from nnacousticmodeling_amd.engine import Engine, Model, Pipeline  # noqa: E402


def main():
    lib = _lib.load()
    rng = np.random.default_rng(31)
    params = syn.make_lstm_params(rng, 40, 512, 3, 1909)
    jobs = []
    for k in range(2):
        eng = Engine(0)
        off = syn.make_offsets(np.random.default_rng(40 + k), 192, 30000, lo=60, hi=400)
        x = np.random.default_rng(50 + k).standard_normal((30000, 40)).astype(np.float32)
        p = Pipeline(eng)
        p.set_frontend(40, [0])
        p.set_members([Model(eng, "lstm", 3, params)])
        p.lstm_upload(x, None, off, 0)
        p.lstm_run()
        jobs.append({"eng": eng, "p": p, "alone": p.lstm_download()})
        print("engine", k, "form", p.recurrent_form(), flush=True)
    j = jobs[0]
    for n in (1, 2, 6):
        t0 = time.perf_counter()
        for _ in range(n):
            j["p"].lstm_run()
        j["p"].synchronize()
        print("one engine, %d runs back to back: %.1f ms, give-ups so far %d" % (n, (time.perf_counter() - t0) * 1e3, lib.am_lstm_fallback_count()), flush=True)
    for n in (1, 3):
        start = threading.Barrier(2)
        times = {}

        def work(i, job):
            start.wait()
            t0 = time.perf_counter()
            for _ in range(n):
                job["p"].lstm_run()
            job["p"].synchronize()
            times[i] = (time.perf_counter() - t0) * 1e3

        th = [threading.Thread(target=work, args=(i, job)) for i, job in enumerate(jobs)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        print("two engines, %d runs each: %.1f ms (per thread %s), give-ups so far %d" % (n, (time.perf_counter() - t0) * 1e3, times, lib.am_lstm_fallback_count()), flush=True)
    for job in jobs:
        assert np.array_equal(job["p"].lstm_download(), job["alone"])
    print("results equal")


if __name__ == "__main__":
    main()

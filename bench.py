#!/usr/bin/env python3
"""Headline benchmark: frames/sec of posterior compute, 4x1024 ReLU DNN on fbank40 (BASELINE.json configs[1]).

One "step" = one pass of the whole device path over one synthetic file already resident in HBM:
  fused splice(+-5) + Kaldi feature transform -> 440 -> 4 x (1024, ReLU) -> 1909 -> y - logsum(y)  (fp32).
`value` = frames all ranks processed / wall time of the K timed steps (max over ranks).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, each rank runs the same per-GPU workload on its own file (frames shard embarrassingly:
no data-path collective; torch.distributed is used only for the barrier and the max-over-ranks).  scaling = weak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20181806
FLOPS_PER_FRAME = {  # 2 * sum(in * out)
    "4x1024": 2 * (440 * 1024 + 3 * 1024 * 1024 + 1024 * 1909),
    "6x2048": 2 * (540 * 2048 + 5 * 2048 * 2048 + 2048 * 1909),
}
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=196608, help="frames per GPU per step (192 x 1024, BASELINE.json configs[1])")
    ap.add_argument("--model", default="4x1024", choices=sorted(FLOPS_PER_FRAME))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="skip the secondary split-f16 precision measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary records (end to end, N=57,919, 6x2048+ivec, 3x512 LSTM)")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="let ranks share a physical GPU when the process sees fewer devices than ranks (rehearsals only; the record says so)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="time budget of the CPU baseline samples (all of them together)")
    return ap.parse_args()


def make_workload(model, frames, rank):
    from nnacousticmodeling_amd.kaldi_ft import loadKaldiFeatureTransform
    from nnacousticmodeling_amd.synthetic import make_mlp_params
    rng = np.random.default_rng(SEED + rank)
    golden = os.path.join(ROOT, "tests", "golden")
    ft = loadKaldiFeatureTransform(os.path.join(golden, "final.feature_transform"))
    layers, units = (4, 1024) if model == "4x1024" else (6, 2048)
    d_iv = 0 if model == "4x1024" else 100
    params = make_mlp_params(np.random.default_rng(SEED), 440 + d_iv, units, layers, 1909)
    x = rng.standard_normal((frames, 40)).astype(np.float32)
    iv = rng.standard_normal((frames, d_iv)).astype(np.float32) if d_iv else None
    return ft, params, layers, x, iv, d_iv


def pmc_traffic(kernel_prefix="am::gemm_nt_f32_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (tools/prof.sh -> tools/summarise_profiles.py): (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of
    MI355X_MICROARCH.md (FETCH_SIZE counts half of a 16-B-per-lane streaming read), averaged over the kernel's launches.
    PMC counters cannot be collected from inside the process, so this is read from profiles/ (None if absent)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None
    summ = json.load(open(files[-1]))
    tot, n = 0.0, 0
    for k, v in summ.items():
        if kernel_prefix in k and "derived_hbm_bytes_per_launch" in v:
            ln = v["FETCH_SIZE"]["launches"]
            tot += v["derived_hbm_bytes_per_launch"] * ln
            n += ln
    return (tot / n if n else None), os.path.relpath(files[-1], ROOT)


def _host_cpu():
    """(model name, physical cores, logical CPUs) of this host from /proc/cpuinfo (north_star: "core count stated")."""
    model, cores, logical = None, set(), 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                logical += 1
            elif key == "model name" and model is None:
                model = val
            elif key == "physical id":
                phys = val
            elif key == "core id":
                core = val
            elif not key and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, (len(cores) or None), (logical or os.cpu_count()), usable


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(ft, params, layers, x, iv, seconds):
    """The oracle (numpy restatement of the Chainer CPU forward, kind = "port") on bounded samples of the same file.
    Main figure: predict()'s own batching -- 1024-frame batches: prepareBatch -> transform -> MLP -> y - logsum(y), all BLAS
    threads.  Beside it: the same with ONE BLAS thread, evaluate's per-utterance batching
    (evaluateModelForTest.py:98-107: one forward per utterance, ~300 rows), and the reference's Python-loop splice
    (kw_utils.py:24-36) on a 3,000-frame sample, which is where a real run of the reference spends its front-end time."""
    from oracle import am_oracle as orc
    model = orc.MLP(params, layers)

    def predict_batches(budget):
        done, t0 = 0, time.perf_counter()
        while done < x.shape[0] and time.perf_counter() - t0 < budget:
            hi = min(done + 1024, x.shape[0])
            xb = orc.prepare_batch(x, np.arange(done, hi), 11)
            xb = orc.apply_feature_transform(xb, ft)
            if iv is not None:
                xb = np.concatenate((xb, iv[done:hi]), axis=1)
            y = model(xb)
            y = y - orc.logsum(y, axis=1)
            done = hi
        return done, time.perf_counter() - t0

    threads = int(_blas_threads())
    # 1024-row batches do not scale over a whole host's BLAS threads (64 threads measured SLOWER than one in round 2), so the
    # headline figure is taken at the best of a few thread counts, each on its own bounded sample, and says which one it was
    by_threads = {}
    try:
        from threadpoolctl import threadpool_limits
        counts = sorted({1, min(4, threads), min(16, threads), threads})
        for c in counts:
            with threadpool_limits(limits=c):
                predict_batches(0.3)                                   # warm the pool at this size
                d, t = predict_batches(seconds * 0.6 / len(counts))
            by_threads[c] = (d / t, d, t)
    except Exception:   # threadpoolctl missing: one measurement with whatever BLAS does
        d, t = predict_batches(seconds * 0.5)
        by_threads[threads] = (d / t, d, t)
    best = max(by_threads, key=lambda c: by_threads[c][0])
    v, done, dt = by_threads[best]
    cpu_model, phys_cores, logical_cpus, usable_cpus = _host_cpu()
    out = {"value": v, "unit": "frames/s", "cores": best, "kind": "port",
           "host_cpu_model": cpu_model, "physical_cores": phys_cores, "logical_cpus": logical_cpus, "cpus_usable_by_this_process": usable_cpus,
           "blas_threads_best": best,
           "sample": "first %d frames of the same synthetic file, 1024-frame batches as predict_folds.py:71-90 "
                     "(numpy/BLAS restatement of the Chainer CPU forward; Chainer itself is not installable here), %.1f s; "
                     "best of the BLAS thread counts tried" % (done, dt),
           "by_blas_threads": {str(c): {"value": r[0], "frames": r[1], "seconds": round(r[2], 2)} for c, r in sorted(by_threads.items())},
           "host_blas_threads": threads}
    if 1 in by_threads:
        out["one_thread"] = {"value": by_threads[1][0], "unit": "frames/s", "cores": 1,
                             "sample": "first %d frames, same batching, %.1f s" % (by_threads[1][1], by_threads[1][2])}
    # evaluate's batching: one forward per utterance on the already spliced + transformed file
    rng = np.random.default_rng(SEED)
    n_e = min(x.shape[0], 30000)
    from nnacousticmodeling_amd.synthetic import make_offsets
    off = make_offsets(rng, max(n_e // 300, 1), n_e)
    data = orc.evaluate_frontend(x[:n_e], 5, ft, None if iv is None else iv[:n_e])
    t0, u = time.perf_counter(), 0
    while u < len(off) - 1 and time.perf_counter() - t0 < seconds * 0.2:
        y = model(data[off[u]:off[u + 1]])
        y = y - orc.logsum(y, axis=1)
        u += 1
    te = time.perf_counter() - t0
    out["per_utterance_batches"] = {"value": float(off[u] / te), "unit": "frames/s", "cores": threads,
                                    "sample": "%d utterances (%d frames), one forward per utterance as evaluateModelForTest.py:98-107, "
                                              "front end excluded, %.1f s" % (u, off[u], te)}
    ns = min(3000, x.shape[0])
    t0 = time.perf_counter()
    orc.splice_rowloop(x[:ns], ft["shifts"])
    ts = time.perf_counter() - t0
    out["python_loop_splice"] = {"value": ns / ts, "unit": "frames/s", "cores": 1,
                                 "sample": "the reference's per-row Python loop (kw_utils.py:24-36) on %d frames, %.3f s" % (ns, ts)}
    return out


def _fallbacks():
    from nnacousticmodeling_amd import _lib
    return int(_lib.load().am_lstm_fallback_count())


def timed_runs(eng, fn, steps, warmup, barrier=None, pipe=None):
    """(wall seconds of `steps` un-instrumented runs, {kernel: (launches, ms)} of `steps` further runs bracketed by HIP events).
    The headline figure and the per-kernel profile come from SEPARATE loops: event bracketing perturbs what it measures.
    With `pipe`: after the loops the pipeline is synchronised and checked (am_pipeline_synchronize) -- a persistent launch that
    gave up on a bounded wait would have made the timed passes 2-s stalls, or invalid; either way the record must not stand."""
    before = _fallbacks()
    for _ in range(warmup):
        fn()
    eng.synchronize()
    if barrier:
        barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    eng.record(0)
    for _ in range(steps):
        fn()
    eng.record(1)
    eng.synchronize()
    wall = time.perf_counter() - t0
    if barrier:
        barrier()
    dev_ms = eng.elapsed_ms(0, 1)
    eng.profile(True)
    eng.profile_reset()
    for _ in range(steps):
        fn()
    eng.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    if pipe is not None:
        pipe.synchronize()
    if _fallbacks() != before:
        raise RuntimeError("a persistent kernel gave up on an inter-workgroup wait during a timed loop: the record is invalid")
    return wall, dev_ms, prof


def gemm_roofline(prof, flops_total):
    calls = sum(prof[k][0] for k in ("gemm_hidden", "gemm_out", "gemm_stack") if k in prof)
    ms = sum(prof[k][1] for k in ("gemm_hidden", "gemm_out", "gemm_stack") if k in prof)
    achieved = flops_total / (ms * 1e-3) / 1e12 if ms > 0 else None
    return calls, ms, achieved


class _Guarded:
    """A secondary record that fails must not take the bench line (and the records behind it) down with it: the failure is recorded in its place."""

    def __init__(self, out, name):
        self.out, self.name, self.owned = out, name, []

    def __enter__(self):
        return self

    def own(self, obj):
        """Device objects of this record (models, pipelines, decoders): closed when the block ends, however it ends -- a failed record
        must not leave gigabytes of planes and member buffers allocated under the records behind it."""
        self.owned.append(obj)
        return obj

    def __exit__(self, etype, exc, tb):
        for obj in reversed(self.owned):
            try:
                obj.close()
            except Exception:
                pass
        self.owned = []
        if etype is not None and issubclass(etype, Exception):
            self.out[self.name] = {"error": "%s: %s" % (etype.__name__, exc)}
            return True
        return False


def secondary_records(eng, ft, params, layers, x, steps):
    """The other BASELINE.json configurations and SURVEY.md 8d figures, one GPU, outside the headline `value`:
    end to end (host array in, pinned host array out), the core-test-like size, 6x2048 + i-vectors, the 3x512 LSTM, the 10-member ensemble."""
    from nnacousticmodeling_amd.engine import Model, Pipeline
    from nnacousticmodeling_amd.kaldi_ft import loadKaldiFeatureTransform, select_centre_frame
    from nnacousticmodeling_amd.synthetic import make_lstm_params, make_mlp_params, make_offsets
    out = {}
    golden = os.path.join(ROOT, "tests", "golden")
    n_full = x.shape[0]
    m4 = Model(eng, "ff", layers, params)
    n57 = 57919
    ftc = select_centre_frame(loadKaldiFeatureTransform(os.path.join(golden, "final.feature_transform")))
    off = make_offsets(np.random.default_rng(SEED + 3), 192, n57)   # 192 utterances of the core-test-like file (recurrent and decoder records)
    from nnacousticmodeling_amd.synthetic import make_gru_params, make_tdnn_params
    # (i) end to end: H2D of the frames, compute, D2H of the posteriors into page-locked memory (PCIe-inclusive; never `value`)
    with _Guarded(out, "end_to_end") as g:
        p = g.own(Pipeline(eng))
        p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        p.set_members([m4])
        y = eng.pinned_empty((n_full, 1909))
        p.forward_ff(x, out=y)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            p.forward_ff(x, out=y)
        dt = (time.perf_counter() - t0) / reps
        out["end_to_end"] = {"value": n_full / dt, "unit": "frames/s", "ms_per_pass": dt * 1e3, "frames": n_full,
                             "what": "host array -> H2D -> splice+transform+4x1024+log-softmax -> D2H into page-locked host memory "
                                     "(7.6 KB of posteriors per frame over PCIe; download of chunk c overlaps compute of chunk c+1)"}
        del y
    # (ii) the same network at the TIMIT core-test size (57,919 frames = 226.2 row tiles: partial last round of tiles)
    with _Guarded(out, "n57919") as g:
        g.own(m4)
        p = g.own(Pipeline(eng))
        p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        p.set_members([m4])
        p.ff_upload(x[:n57], None, n57, 0, 0)
        wall, _, prof = timed_runs(eng, p.ff_run, max(steps, 5), 2, pipe=p)
        k = max(steps, 5)
        whole = FLOPS_PER_FRAME["4x1024"] * n57 * k / wall / 1e12
        out["n57919"] = {"value": n57 * k / wall, "unit": "frames/s", "ms_per_step": wall * 1e3 / k, "frames": n57,
                         "roofline": {"bound": "mfma", "kernel": "whole step (splice + transform, the Linear stack, log-softmax): the fraction is flop / wall time / peak",
                                      "achieved": whole, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": whole / PEAK_F32_MFMA_TFLOPS},
                         "gemm_launch_ms_sum_per_step": sum(prof[k_][1] for k_ in ("gemm_hidden", "gemm_out", "gemm_stack") if k_ in prof) / k}
        _, gms, gach = gemm_roofline(prof, FLOPS_PER_FRAME["4x1024"] * n57 * k)
        out["n57919"]["roofline_gemm"] = {"bound": "mfma", "kernel": "the Linear layers' launches alone (gemm_stack_kernel: one launch per step), as for the headline record",
                                          "achieved": gach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": gach / PEAK_F32_MFMA_TFLOPS if gach else None}
        p.close()
        m4.close()
    # (iii) BASELINE configs[2]: 6x2048 on 440 spliced + 100 i-vector columns
    with _Guarded(out, "dnn_6x2048_ivec") as g:
        rng = np.random.default_rng(SEED + 2)
        p6 = make_mlp_params(rng, 540, 2048, 6, 1909)
        iv = rng.standard_normal((n_full, 100)).astype(np.float32)
        m6 = g.own(Model(eng, "ff", 6, p6))
        p = g.own(Pipeline(eng))
        p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"], 100)
        p.set_members([m6])
        p.ff_upload(x, iv, n_full, 0, 0)
        k6 = max(steps // 4, 3)
        wall, _, prof = timed_runs(eng, p.ff_run, k6, 1, pipe=p)
        _, _, ach = gemm_roofline(prof, FLOPS_PER_FRAME["6x2048"] * n_full * k6)
        out["dnn_6x2048_ivec"] = {"value": n_full * k6 / wall, "unit": "frames/s", "ms_per_step": wall * 1e3 / k6, "frames": n_full,
                                  "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS if ach else None}}
        p.close()
        m6.close()
        del iv
    # (iv) BASELINE configs[3]: 3x512 LSTM (uni-directional: the reference has no bidirectional net), 192 utterances
    with _Guarded(out, "lstm_3x512") as g:
        rng = np.random.default_rng(SEED + 3)
        make_offsets(rng, 192, n57)   # (advances rng exactly as before: the same weights as earlier rounds)
        ml = g.own(Model(eng, "lstm", 3, make_lstm_params(rng, 40, 512, 3, 1909)))
        p = g.own(Pipeline(eng))
        p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
        p.set_members([ml])
        p.lstm_upload(x[:n57], None, off, 0)
        kl = max(steps // 2, 5)
        wall, _, prof = timed_runs(eng, p.lstm_run, kl, 3, pipe=p)
        fl = 2 * (40 * 2048 + 2 * 512 * 2048 + 3 * 512 * 2048 + 512 * 1909)
        rec_ms = prof.get("lstm_step", (0, 0.0))[1] / kl
        rec_flops = 2 * (2 * 512 * 2048 + 3 * 512 * 2048) * n57       # the recurrent kernel's share: lateral x3 + upward x2
        out["lstm_3x512"] = {"value": n57 * kl / wall, "unit": "frames/s", "ms_per_pass": wall * 1e3 / kl, "frames": n57, "utterances": 192,
                             "longest_utterance": int(np.diff(off).max()),
                             "roofline": {"bound": "mfma (dependent time steps: latency-bound below ~64 active utterances)", "kernel": "whole pass",
                                          "achieved": fl * n57 * kl / wall / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                          "frac": fl * n57 * kl / wall / 1e12 / PEAK_F32_MFMA_TFLOPS},
                             "recurrent_kernel": {"ms_per_pass": rec_ms, "launches_per_pass": prof.get("lstm_step", (0, 0.0))[0] // kl,
                                                  "achieved_tflops": rec_flops / (rec_ms * 1e-3) / 1e12 if rec_ms > 0 else None},
                             "kernels_ms_per_pass": {k_: v[1] / kl for k_, v in prof.items()}}
        p.close()
        ml.close()
    # (iv b) the reference's own recurrent recipe shape (scripts/papers/specom2018/run.sh:9: "-n lstm -l 4 -u 1024 --timedelay 5"): its weights do not fit
    # the register file; round 4: the streamed persistent form (lstm_stream_kernel, one launch, 256 workgroups), rounds 1-3: one launch per (layer, time) diagonal
    with _Guarded(out, "lstm_4x1024_td5") as g:
        rng = np.random.default_rng(SEED + 6)
        ml4 = g.own(Model(eng, "lstm", 4, make_lstm_params(rng, 40, 1024, 4, 1909)))
        p = g.own(Pipeline(eng))
        p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
        p.set_members([ml4])
        p.lstm_upload(x[:n57], None, off, 5)
        k4 = 3
        wall, _, prof = timed_runs(eng, p.lstm_run, k4, 2, pipe=p)
        fl = 2 * (40 * 4096 + 4 * 1024 * 4096 + 3 * 1024 * 4096 + 1024 * 1909)
        out["lstm_4x1024_td5"] = {"value": n57 * k4 / wall, "unit": "frames/s", "ms_per_pass": wall * 1e3 / k4, "frames": n57, "utterances": 192,
                                  "roofline": {"bound": "mfma (dependent time steps)", "kernel": "whole pass", "achieved": fl * n57 * k4 / wall / 1e12,
                                               "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": fl * n57 * k4 / wall / 1e12 / PEAK_F32_MFMA_TFLOPS},
                                  "recurrent_form": "%d of %d members as one persistent launch" % p.recurrent_form(),
                                  "launches_per_pass": {k_: v[0] // k4 for k_, v in prof.items()}}
        p.close()
        ml4.close()
    # (iv c) the other specom2018 recipe shape (run.sh:10: "-n gru -l 4 -u 1024 --timedelay 5"): per-step phase kernels, two dependent launches per tick
    with _Guarded(out, "gru_4x1024_td5") as g:
        rng = np.random.default_rng(SEED + 8)
        mg4 = g.own(Model(eng, "gru", 4, make_gru_params(rng, 40, 1024, 4, 1909, reset_gate=True)))
        p = g.own(Pipeline(eng))
        p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
        p.set_members([mg4])
        p.lstm_upload(x[:n57], None, off, 5)
        kq = 2
        wall, _, prof = timed_runs(eng, p.lstm_run, kq, 1, pipe=p)
        fl = 2 * (3 * (40 * 1024 + 3 * 1024 * 1024 + 4 * 1024 * 1024) + 1024 * 1909)
        out["gru_4x1024_td5"] = {"value": n57 * kq / wall, "unit": "frames/s", "ms_per_pass": wall * 1e3 / kq, "frames": n57, "utterances": 192,
                                 "roofline": {"bound": "mfma (dependent time steps)", "kernel": "whole pass", "achieved": fl * n57 * kq / wall / 1e12,
                                              "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": fl * n57 * kq / wall / 1e12 / PEAK_F32_MFMA_TFLOPS},
                                 "recurrent_form": "%d of %d members as one persistent launch" % p.recurrent_form(),
                                 "launches_per_pass": {k_: v[0] // kq for k_, v in prof.items()}}
        p.close()
        mg4.close()
    # (v) BASELINE configs[4]: 10 fold members (4x1024 each), logits averaged on the device (evaluate.py:35-51) -- one GPU's share;
    # on a node the folds x utterances shard over the GPUs with no collective (sharding.py)
    with _Guarded(out, "ensemble_10x4x1024") as g:
        rng = np.random.default_rng(SEED + 4)
        members = [g.own(Model(eng, "ff", layers, make_mlp_params(rng, 440, 1024, layers, 1909))) for _ in range(10)]
        p = g.own(Pipeline(eng))
        p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
        p.set_members(members)
        p.ff_upload(x[:n57], None, n57, 0, 0)
        ke = max(steps // 6, 3)
        wall, _, prof = timed_runs(eng, p.ff_run, ke, 1, pipe=p)
        whole = 10 * FLOPS_PER_FRAME["4x1024"] * n57 * ke / wall / 1e12
        _, _, gach = gemm_roofline(prof, 10 * FLOPS_PER_FRAME["4x1024"] * n57 * ke)
        out["ensemble_10x4x1024"] = {"value": n57 * ke / wall, "unit": "frames/s", "ms_per_pass": wall * 1e3 / ke, "frames": n57, "members": 10,
                                     "roofline": {"bound": "mfma", "kernel": "whole pass (10 stack launches + combine + log-softmax)", "achieved": whole,
                                                  "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": whole / PEAK_F32_MFMA_TFLOPS},
                                     "roofline_gemm": {"bound": "mfma", "achieved": gach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                                       "frac": gach / PEAK_F32_MFMA_TFLOPS if gach else None}}
        p.close()
        for m_ in members:
            m_.close()
    # (vi) row n4: the GRU stack at the recurrent config's size (chainer L.GRU: two dependent products per step), and the reference's own MGRU
    # (MGRU.py:67-85, no reset gate: one product per step).  Per-tick phase kernels, weights in stream order; the persistent form of these cells
    # (rec_stream_kernel, AM_REC_STREAM) measured slower and is not the default.
    for key, net, seed_off in (("gru_3x512", "gru", 5), ("mgrurelu_3x512", "mgrurelu", 11)):
        with _Guarded(out, key) as g:
            rng = np.random.default_rng(SEED + seed_off)
            mg = g.own(Model(eng, net, 3, make_gru_params(rng, 40, 512, 3, 1909, reset_gate=net == "gru")))
            p = g.own(Pipeline(eng))
            p.set_frontend(40, [0], ftc["addShift"], ftc["rescale"])
            p.set_members([mg])
            p.lstm_upload(x[:n57], None, off, 0)
            kg = max(steps // 4, 3)
            wall, _, prof = timed_runs(eng, p.lstm_run, kg, 2, pipe=p)
            gates = 3 if net == "gru" else 2
            fl = 2 * (gates * (40 * 512 + 2 * 512 * 512 + 3 * 512 * 512) + 512 * 1909)
            out[key] = {"value": n57 * kg / wall, "unit": "frames/s", "ms_per_pass": wall * 1e3 / kg, "frames": n57, "utterances": 192,
                        "roofline": {"bound": "mfma (dependent time steps)", "kernel": "whole pass", "achieved": fl * n57 * kg / wall / 1e12,
                                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": fl * n57 * kg / wall / 1e12 / PEAK_F32_MFMA_TFLOPS},
                        "recurrent_form": "%d of %d members as one persistent launch" % p.recurrent_form(),
                        "kernels_ms_per_pass": {k_: v[1] / kg for k_, v in prof.items()}, "launches_per_pass": {k_: v[0] // kg for k_, v in prof.items()}}
            p.close()
            mg.close()
    # (vii) row n4: TDNN 5-5-5-5 x 1024 (scripts/papers/tsd2018/run.sh:22-30): every (1, k) convolution as one MFMA GEMM with windowed rows
    with _Guarded(out, "tdnn_5555x1024") as g:
        from nnacousticmodeling_amd.kaldi_ft import tdnn_splice, tile_centre_frame
        rng = np.random.default_rng(SEED + 7)   # its own stream: the weights do not depend on which records ran (or failed) before
        ksize, units = [5, 5, 5, 5], [1024] * 4
        tft = tile_centre_frame(loadKaldiFeatureTransform(os.path.join(golden, "final.feature_transform")), tdnn_splice(ksize))
        mt = g.own(Model(eng, "tdnn", 4, make_tdnn_params(rng, 40, units, ksize, 1909)))
        p = g.own(Pipeline(eng))
        p.set_frontend(40, tft["shifts"], tft["addShift"], tft["rescale"])
        p.set_members([mt])
        p.ff_upload(x[:n57], None, n57, 0, 0)
        kt = 3
        wall, _, prof = timed_runs(eng, p.ff_run, kt, 1, pipe=p)
        pos, cin, fl = sum(ksize) - 3, 40, 0
        for u_, k_ in zip(units, ksize):
            pos -= k_ - 1
            fl += 2 * pos * cin * k_ * u_
            cin = u_
        fl += 2 * cin * 1909
        _, _, gach = gemm_roofline(prof, fl * n57 * kt)
        out["tdnn_5555x1024"] = {"value": n57 * kt / wall, "unit": "frames/s", "ms_per_pass": wall * 1e3 / kt, "frames": n57, "flop_per_frame": fl,
                                 "roofline": {"bound": "mfma", "kernel": "the convolution / output GEMM launches", "achieved": gach, "peak": PEAK_F32_MFMA_TFLOPS,
                                              "unit": "TFLOP/s", "frac": gach / PEAK_F32_MFMA_TFLOPS if gach else None}}
        p.close()
        mt.close()
    # (viii) row n1: the frame-synchronous Viterbi decoder on a synthetic graph of the real image's size (48 phones, 2,304 nodes,
    # ~88k triphones, 1909 states; the inputs tests/golden/make_viterbi_fullsize_golden.py describes), 192 utterances / 57,919 frames
    with _Guarded(out, "viterbi_realsize") as g:
        from nnacousticmodeling_amd.phone_recog import PhoneRecog
        from nnacousticmodeling_amd.synthetic import make_graph
        rng = np.random.default_rng(481909)
        graph = make_graph(rng, 48, 1909)
        T_, B_ = len(graph["biphoneIdxIn"]), len(graph["lm"])
        z = rng.standard_normal((n57, 1909), dtype=np.float32)
        z *= 2.0
        mx = z.max(axis=1, keepdims=True)
        z -= mx + np.log(np.exp(z - mx).sum(axis=1, keepdims=True))
        dec = g.own(PhoneRecog(eng, dict(graph, numPhones=48, numBiphones=B_, numTriphones=T_), -20.0, 1.0))
        dec.recognise(z[:off[4]], off[:5])
        t0 = time.perf_counter()
        res = dec.recognise(z, off)
        host_s = time.perf_counter() - t0
        pass_ms = eng.decode_elapsed_ms()   # device time of the un-instrumented pass: first launch -> last launch done (am_decode_* records both)
        eng.profile(True)
        eng.profile_reset()
        dec.recognise(z, off)
        vprof = eng.profile_read()
        eng.profile(False)
        vit_ms = pass_ms                   # the roofline figure: the whole pass as it runs in production, launch gaps included
        steps_v = int(np.diff(off).max())
        # HBM bytes the pass cannot avoid: per (utterance, frame) the state of every triphone read (18 B: 3 scores, 3 16-bit entry times) and written (18 B: the exit scores
        # stay in LDS since the node maxima are taken in the same launch), the node back-pointers (8 B) and the frame's posterior row; per
        # frame-launch the graph constants once (32 B per triphone, shared by all utterances of the launch: L2 / Infinity Cache serve the repeats)
        vit_bytes = n57 * (T_ * 36 + B_ * 8 + 1909 * 4) + steps_v * T_ * 32
        out["viterbi_realsize"] = {"value": n57 / (vit_ms * 1e-3), "unit": "frames/s (device time of the pass)", "pass_ms": vit_ms,
                                   "kernel_ms_sum_with_an_event_pair_per_launch": vprof["viterbi"][1], "launches": vprof["viterbi"][0],
                                   "host_array_to_segments_s": host_s, "frames": n57, "utterances": 192, "segments": sum(len(r) for r in res),
                                   "graph": {"phones": 48, "nodes": B_, "triphones": T_, "states": 1909},
                                   "roofline": {"bound": "hbm", "kernel": "viterbi_fs_stream_kernel (triphone update + node maxima, one launch per frame: a block keeps a triphone block and walks the active utterance slots; AM_VITERBI_LDS=2 selects round 3's viterbi_fs_fused_kernel)",
                                                "achieved": vit_bytes / (vit_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": vit_bytes / (vit_ms * 1e-3) / 8e12,
                                                "algorithmic_bytes": vit_bytes,
                                                "measured": "one HIP event pair around the pass's launch sequence on the engine's compute stream"}}
        # CPU leg of this record: the reference's own decoder built from source (oracle/_ref, checker use only), ONE thread as the
        # shipped binary, on a bounded sample of the same utterances; also checks the GPU segmentations on that sample
        from oracle import viterbi_oracle as vo
        if vo.reference_available():
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                img = os.path.join(td, "g.img")
                vo.write_image(img, graph)
                ku = 1
                while ku < 192 and off[ku] < 6000:
                    ku += 1
                sample = [z[off[u]:off[u + 1]] for u in range(ku)]
                t0 = time.perf_counter()
                ref = vo.reference_recog(img, sample, -20.0, 1.0)
                cpu_s = time.perf_counter() - t0
            cpu_model, phys_cores, _, _ = _host_cpu()
            out["viterbi_realsize"]["cpu_baseline"] = {"value": float(off[ku]) / cpu_s, "unit": "frames/s", "cores": 1, "kind": "reference",
                                                       "sample": "the first %d utterances (%d frames) of the same file through recog_src/source/phone_recog.cpp built by oracle/Makefile, %.1f s" % (ku, int(off[ku]), cpu_s),
                                                       "host_cpu_model": cpu_model, "physical_cores": phys_cores,
                                                       "segmentations_identical_on_sample": bool(all(ref[u] == res[u] for u in range(ku)))}
        else:
            out["viterbi_realsize"]["cpu_baseline"] = {"skipped": "oracle/_ref/libphonerecog_ref.so not present (built by __graft_entry__.build() where /root/reference exists)"}
        dec.close()
        del z
    return out


def product_host_to_host(n_devices, ft, params, layers, x):
    """The PRODUCT's multi-GPU path on one file, host arrays in -> host arrays out (PCIe-inclusive, never `value`): one host
    process, one thread + engine per GPU (sharding.PosteriorRunner), (a) gathered into ONE page-locked array (the reference's
    contract: predict() returns the file's posteriors), (b) every shard keeping its rows in its own page-locked block
    (map_shards: what the .lab / .npy writers and the per-shard decoder consume)."""
    from nnacousticmodeling_amd import networks, sharding
    from nnacousticmodeling_amd.engine import device_count
    have = device_count()
    if have < n_devices:
        return {"skipped": "this process sees %d GPUs, %d asked for" % (have, n_devices)}
    spec = networks.get_nn("ff", layers, [params["layer_0/W"].shape[0]], 1909).set_params(params)
    runner = sharding.PosteriorRunner(list(range(n_devices)))
    n = x.shape[0]
    out = {"gpus": n_devices, "frames": n, "what": "one %d-frame file through PosteriorRunner(range(%d)): H2D, splice+transform+4x1024+log-softmax, D2H" % (n, n_devices)}
    try:
        kw = dict(shifts=ft["shifts"], ft=ft)
        for _ in range(2):                       # engines, weights, page-locked blocks
            runner.release(runner.forward_ff(spec, x, **kw))
        reps, t0 = 4, time.perf_counter()
        for _ in range(reps):
            runner.release(runner.forward_ff(spec, x, **kw))
        dt = (time.perf_counter() - t0) / reps
        out["gathered_single_array"] = {"value": n / dt, "unit": "frames/s", "ms_per_pass": dt * 1e3}

        def keep(ctx):
            return ctx.download().shape[0]
        for _ in range(2):
            runner.map_shards(spec, x, keep, host=True, **kw)
        t0 = time.perf_counter()
        for _ in range(reps):
            rows = runner.map_shards(spec, x, keep, host=True, **kw)
        dt = (time.perf_counter() - t0) / reps
        assert sum(rows) == n
        out["per_shard_blocks"] = {"value": n / dt, "unit": "frames/s", "ms_per_pass": dt * 1e3}
    finally:
        runner.close()
    return out


def main():
    args = parse_args()
    from bench_ranks import Ranks
    from nnacousticmodeling_amd.engine import Engine, Model, Pipeline
    ranks = Ranks()  # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env; gloo for barrier + max-over-ranks only
    rank, local_rank, n_gpus = ranks.rank, ranks.local_rank, ranks.world

    ft, params, layers, x, iv, d_iv = make_workload(args.model, args.frames, rank)
    from nnacousticmodeling_amd.engine import device_count, device_pci_bus_id
    n_dev = device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no HIP device visible (there is no CPU fallback)")
    if local_rank >= n_dev and not args.allow_shared_gpu:
        # a run that silently put two ranks on one card would print "n_gpus": N for fewer than N GPUs' worth of work
        raise SystemExit("bench.py: local rank %d but only %d GPU(s) visible; one rank per GPU (--allow-shared-gpu to rehearse on fewer cards)" % (local_rank, n_dev))
    dev = local_rank % n_dev
    bus_ids = ranks.gather_objects(device_pci_bus_id(dev))   # which physical card every rank really drives
    eng = Engine(dev)
    pipe = Pipeline(eng)
    pipe.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"], d_iv)
    pipe.set_members([Model(eng, "ff", layers, params)])
    pipe.ff_upload(x, iv, args.frames, 0, 0)  # inputs resident in HBM before the timed region

    # headline loop (no per-kernel events), then the same K steps again with every launch bracketed by HIP events on the
    # engine's compute stream for the roofline figures
    wall, dev_ms, prof = timed_runs(eng, pipe.ff_run, args.steps, args.warmup, ranks.barrier, pipe=pipe)
    wall = ranks.max_over_ranks(wall)

    # secondary measurement, outside the timed region and not part of `value`: the same pass with the Linear layers in the
    # split-f16 precision mode (f32 operands as hi+lo f16 halves on the f16 matrix pipe, f32 accumulate; include/amfwd.h)
    split = None
    if not args.no_split:
        y_exact = pipe.ff_download()
        eng.set_precision("split_f16")
        wall_s, _, prof_s = timed_runs(eng, pipe.ff_run, args.steps, max(args.warmup, 1), ranks.barrier, pipe=pipe)
        wall_s = ranks.max_over_ranks(wall_s)
        overflow = pipe.overflowed()
        y_split = pipe.ff_download()
        eng.set_precision("f32")
        split = {"precision": "f32 operands as hi+lo f16 halves, 3 f16 MFMAs per product, f32 accumulate (AM_PRECISION_SPLIT_F16)",
                 "value": args.frames * args.steps * n_gpus / wall_s, "unit": "frames/s", "ms_per_step": wall_s * 1e3 / args.steps,
                 "max_abs_diff_vs_f32_path": float(np.max(np.abs(y_split - y_exact))), "f16_overflow": bool(overflow),
                 "kernels_ms_per_step": {k: v[1] / args.steps for k, v in prof_s.items()},
                 "gemm_tflops_f32_equivalent": FLOPS_PER_FRAME[args.model] * args.frames * args.steps / (prof_s["gemm_split"][1] * 1e-3) / 1e12 if "gemm_split" in prof_s else None}
        if split["gemm_tflops_f32_equivalent"]:
            f16 = 3.0 * split["gemm_tflops_f32_equivalent"]   # three f16 MFMAs per f32 product
            split["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_split_kernel", "achieved": f16, "peak": 2500.0, "unit": "TFLOP/s (f16 MFMA, dense)",
                                 "frac": f16 / 2500.0, "note": "operand stream (LDS-DMA, 64 flop/B at 256x256 tiles) is the practical bound: ~5.2 TB/s of ~6.4 TB/s"}
        del y_exact, y_split

    secondary = None
    if n_gpus == 1 and not args.no_secondary and args.model == "4x1024":
        secondary = secondary_records(eng, ft, params, layers, x, args.steps)
    # the product's own multi-GPU path (one process driving every GPU), rank 0 only, outside `value`: the other ranks have
    # finished their timed loops and wait at the barrier below with idle GPUs
    product = None
    if not args.no_secondary and args.model == "4x1024":
        ranks.barrier()
        if rank == 0:
            try:
                product = product_host_to_host(n_gpus, ft, params, layers, x)
            except Exception as exc:   # a secondary record must not take the headline line down with it
                product = {"error": "%s: %s" % (type(exc).__name__, exc)}
        ranks.barrier()

    if rank == 0:
        frames_total = args.frames * args.steps * n_gpus
        value = frames_total / wall
        flops_step = FLOPS_PER_FRAME[args.model] * args.frames
        gemm_calls, gemm_ms, achieved = gemm_roofline(prof, flops_step * args.steps)
        stack = "gemm_stack" in prof   # the whole Linear stack as one work-queue launch (the default form)
        launches_per_step = gemm_calls // max(args.steps, 1)
        traffic, traffic_src = (pmc_traffic("am::gemm_stack_kernel" if stack else "am::gemm_nt_f32_kernel")
                                if (args.model == "4x1024" and args.frames == 196608) else (None, None))
        out = {
            "metric": "frames/sec (posterior compute), %s ReLU DNN, fbank40 splice+-5, fp32" % args.model,
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (N(0,1) fbank-like frames, random-init weights, real final.feature_transform)",
            "devices": sorted(set(bus_ids)), "ranks_per_device": {b: bus_ids.count(b) for b in sorted(set(bus_ids))},
            "config": {"workload": "%s DNN 440->%s->1909, %d frames per GPU per step (BASELINE.json configs[1], 192 x 1024 frames), "
                                   "splice+transform+forward+log-softmax device-resident" % (args.model, args.model, args.frames),
                       "frames_per_gpu": args.frames, "parallelism": "frame shards, no collective"},
            "device_ms_per_step": dev_ms / args.steps,
            "roofline": {"bound": "mfma", "kernel": ("gemm_stack_kernel (all Linear layers of the network as one work-queue launch: %d launch/step)" if stack else
                                                     "gemm_nt_f32_kernel (all Linear layers: %d launches/step)") % launches_per_step,
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_F32_MFMA_TFLOPS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC, %s)" % traffic_src if traffic else None,
                         # a fused chain must move 160 B in + 7,636 B out per frame (+ the weights once); the traffic above that is
                         # hidden activations materialised between the layers (written once, read once) and A panels re-fetched per XCD.  At
                         # ~0.87 TB/s the launch is nowhere near the HBM bound: the fraction that matters is `frac` (MFMA).
                         "algorithmic_bytes_per_launch": (args.frames * (160 + 7636) + 4 * (440 * 1024 + 3 * 1024 * 1024 + 1024 * 1909)) / float(max(launches_per_step, 1)) if args.model == "4x1024" else None,
                         "materialised_activation_bytes_per_launch": (args.frames * 4 * (440 + 4 * 1024 + 4 * 1024 + 1909) + 4 * (440 * 1024 + 3 * 1024 * 1024 + 1024 * 1909)) / float(max(launches_per_step, 1)) if args.model == "4x1024" else None,
                         "avg_launch_ms": gemm_ms / gemm_calls if gemm_calls else None,
                         "measured": "HIP events around every launch on the engine's compute stream, in a second loop of the same K steps (the headline loop carries no events)"},
            "kernels_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
        }
        if split is not None:
            out["split_f16"] = split
        if secondary is not None:
            out.update(secondary)
        if product is not None:
            out["product_host_to_host"] = product
        out["persistent_fallbacks"] = _fallbacks()   # launches that gave up on a bounded wait in this process (0: every timed pass was a real one)
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ft, params, layers, x, iv, args.cpu_seconds)
        print(json.dumps(out))
    ranks.close()


if __name__ == "__main__":
    main()

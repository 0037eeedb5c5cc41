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
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the CPU baseline sample")
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


def cpu_baseline(ft, params, layers, x, iv, seconds):
    """The oracle (numpy restatement of the Chainer CPU forward, kind = "port") on a bounded sample of the same file:
    predict()'s own batching -- 1024-frame batches: prepareBatch -> transform -> MLP -> y - logsum(y)."""
    from oracle import am_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    model = orc.MLP(params, layers)
    done, t0 = 0, time.perf_counter()
    while done < x.shape[0] and time.perf_counter() - t0 < seconds:
        hi = min(done + 1024, x.shape[0])
        xb = orc.prepare_batch(x, np.arange(done, hi), 11)
        xb = orc.apply_feature_transform(xb, ft)
        if iv is not None:
            xb = np.concatenate((xb, iv[done:hi]), axis=1)
        y = model(xb)
        y = y - orc.logsum(y, axis=1)
        done = hi
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": int(threads), "kind": "port",
            "sample": "first %d frames of the same synthetic file, 1024-frame batches as predict_folds.py:71-90 "
                      "(numpy/BLAS restatement of the Chainer CPU forward; Chainer itself is not installable here), %.1f s" % (done, dt)}


def main():
    args = parse_args()
    from nnacousticmodeling_amd.dist_util import Ranks
    from nnacousticmodeling_amd.engine import Engine, Model, Pipeline
    ranks = Ranks()  # RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env; gloo for barrier + max-over-ranks only
    rank, local_rank, n_gpus = ranks.rank, ranks.local_rank, ranks.world

    ft, params, layers, x, iv, d_iv = make_workload(args.model, args.frames, rank)
    from nnacousticmodeling_amd.engine import device_count
    eng = Engine(local_rank % max(device_count(), 1))  # one rank per GPU; wraps only when ranks outnumber the visible GPUs
    pipe = Pipeline(eng)
    pipe.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"], d_iv)
    pipe.set_members([Model(eng, "ff", layers, params)])
    pipe.ff_upload(x, iv, args.frames, 0, 0)  # inputs resident in HBM before the timed region

    for _ in range(args.warmup):
        pipe.ff_run()
    eng.synchronize()

    barrier = ranks.barrier

    eng.profile(True)
    eng.profile_reset()
    barrier()
    eng.synchronize()
    t0 = time.perf_counter()
    eng.record(0)
    for _ in range(args.steps):
        pipe.ff_run()
    eng.record(1)
    eng.synchronize()
    t1 = time.perf_counter()
    barrier()
    wall = t1 - t0
    dev_ms = eng.elapsed_ms(0, 1)
    prof = eng.profile_read()
    eng.profile(False)
    wall = ranks.max_over_ranks(wall)

    # secondary measurement, outside the timed region and not part of `value`: the same pass with the Linear layers in the
    # split-f16 precision mode (f32 operands as hi+lo f16 halves on the f16 matrix pipe, f32 accumulate; include/amfwd.h)
    split = None
    if not args.no_split:
        y_exact = pipe.ff_download()
        eng.set_precision("split_f16")
        for _ in range(max(args.warmup, 1)):
            pipe.ff_run()
        eng.synchronize()
        eng.profile(True)
        eng.profile_reset()
        barrier()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.ff_run()
        eng.synchronize()
        wall_s = ranks.max_over_ranks(time.perf_counter() - ts0)
        prof_s = eng.profile_read()
        eng.profile(False)
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

    if rank == 0:
        frames_total = args.frames * args.steps * n_gpus
        value = frames_total / wall
        gemm_calls = sum(prof[k][0] for k in ("gemm_hidden", "gemm_out") if k in prof)
        gemm_ms = sum(prof[k][1] for k in ("gemm_hidden", "gemm_out") if k in prof)
        flops_step = FLOPS_PER_FRAME[args.model] * args.frames
        achieved = flops_step * args.steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None
        traffic, traffic_src = pmc_traffic() if (args.model == "4x1024" and args.frames == 196608) else (None, None)
        out = {
            "metric": "frames/sec (posterior compute), %s ReLU DNN, fbank40 splice+-5, fp32" % args.model,
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (N(0,1) fbank-like frames, random-init weights, real final.feature_transform)",
            "config": {"workload": "%s DNN 440->%s->1909, %d frames per GPU per step (BASELINE.json configs[1], 192 x 1024 frames), "
                                   "splice+transform+forward+log-softmax device-resident" % (args.model, args.model, args.frames),
                       "frames_per_gpu": args.frames, "parallelism": "frame shards, no collective"},
            "device_ms_per_step": dev_ms / args.steps,
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_f32_kernel (all Linear layers: %d launches/step)" % (gemm_calls // max(args.steps, 1)),
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_F32_MFMA_TFLOPS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (PMC, %s)" % traffic_src if traffic else None,
                         "algorithmic_bytes_per_launch": (args.frames * 4 * (440 + 4 * 1024 + 4 * 1024 + 1909) + 4 * (440 * 1024 + 3 * 1024 * 1024 + 1024 * 1909)) / 5.0 if args.model == "4x1024" else None,
                         "avg_launch_ms": gemm_ms / gemm_calls if gemm_calls else None},
            "kernels_ms_per_step": {k: v[1] / args.steps for k, v in prof.items()},
        }
        if split is not None:
            out["split_f16"] = split
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ft, params, layers, x, iv, args.cpu_seconds)
        print(json.dumps(out))
    ranks.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Drop-in for the reference's ``scripts/common/predict_folds.py`` (surface B: one ``[N, C]`` float32 ``.npy``).

Same CLI (predict_folds.py:99-125) and the same ``predict(...)`` / ``main(arg_list)`` entry points, but the whole
pipeline -- splice, Kaldi feature transform, DNN/LSTM forward, ``y - logsum(y)`` -- runs on MI355X GPUs through
libamfwd.so.  Extension: ``--gpus N`` shards frames / utterances over N GPUs with a host-side gather.

    python -m nnacousticmodeling_amd.predict_folds -n ff -l 4 -u 1024 --splice 5 --tri --ft final.feature_transform ...
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

from . import chainer_npz
from ._lib import AM_COMBINE_LOGITS, AM_COMBINE_LOGPOST
from .kaldi_ft import loadKaldiFeatureTransform, select_centre_frame, tdnn_splice, tile_centre_frame
from .networks import get_nn, is_nn_recurrent
from .sharding import get_runner, parse_devices, set_default_precision


def predict(model, x, offsets, num_classes, network, gpu, winlen, timedelay, ft, progress=True, combine=AM_COMBINE_LOGITS):
    """Same contract as the reference ``predict`` (predict_folds.py:27-95): float32 ``[N, num_classes]`` of
    ``y - logsum(y)`` in file order.

    ``gpu``: a device id or a list of ids (the reference takes one id; negative = CPU is refused here).
    FF: every frame is spliced with a window of ``winlen`` frames clamped at the ends of the file
    (prepareBatch, kw_nn_utils.py:19-43), transformed, forwarded.  Recurrent: per utterance, un-spliced frames,
    outputs shifted back by ``timedelay`` with the reference's zero tail (predict_folds.py:50,61)."""
    devices = list(gpu) if isinstance(gpu, (list, tuple)) else [gpu]
    runner = get_runner(devices)
    if model.num_classes != num_classes:
        raise ValueError("model has %d classes, expected %d" % (model.num_classes, num_classes))
    if is_nn_recurrent(network):
        return runner.forward_lstm(model, x, offsets, ft=ft, timedelay=timedelay, combine=combine)
    winhalf = int(winlen / 2)
    return runner.forward_ff(model, x, shifts=range(-winhalf, winhalf + 1), ft=ft, combine=combine)


def predict_to_npy(path, model, x, offsets, num_classes, network, gpu, winlen, timedelay, ft, progress=True, combine=AM_COMBINE_LOGITS):
    """``np.save(path, predict(...))`` without the gathered array: the ``.npy`` is created as a memory map (same bytes as
    ``np.save`` writes: predict_folds.py:220,240) and every GPU's shard thread copies its own rows into it from its own
    page-locked block, in parallel (``PosteriorRunner.map_shards``).  Returns the path written (``.npy`` appended like np.save)."""
    devices = list(gpu) if isinstance(gpu, (list, tuple)) else [gpu]
    runner = get_runner(devices)
    if model.num_classes != num_classes:
        raise ValueError("model has %d classes, expected %d" % (model.num_classes, num_classes))
    path = str(path)
    if not path.endswith(".npy"):
        path += ".npy"
    x = np.asarray(x)
    out = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(x.shape[0], num_classes))
    recurrent = is_nn_recurrent(network)
    if recurrent:
        offsets = np.asarray(offsets, dtype=np.int64)

    def per_shard(ctx):
        y = ctx.download()
        if ctx.lo is not None:                       # feed-forward: one contiguous range of the file
            out[ctx.lo:ctx.hi] = y
        else:                                        # recurrent: this shard's utterances, wherever they lie in the file
            for j, u in enumerate(ctx.utts):
                out[offsets[u]:offsets[u + 1]] = y[ctx.local_offsets[j]:ctx.local_offsets[j + 1]]

    winhalf = int(winlen / 2)
    shifts = [0] if recurrent else range(-winhalf, winhalf + 1)
    runner.map_shards(model, x, per_shard, offsets=offsets if recurrent else None, shifts=shifts, ft=ft, recurrent=recurrent,
                      timedelay=timedelay, combine=combine, host=True)
    out.flush()
    del out
    return path


def build_parser():
    parser = argparse.ArgumentParser(description="MI355X fold prediction (drop-in for predict_folds.py)")
    parser.add_argument("--network", "-n", default="ff", help='Neural network type, "ff" or "lstm"')
    parser.add_argument("--gpu", "-g", type=int, default=0, help="first GPU ID")
    parser.add_argument("--gpus", type=int, default=1, help="number of GPUs to shard over (extension)")
    parser.add_argument("--units", "-u", type=int, nargs="+", default=[1024], help="Number of units")
    parser.add_argument("--layers", "-l", type=int, default=2, help="Number of hidden layers")
    parser.add_argument("--activation", "-a", default="relu", help="FF activation function (sigmoid, tanh or relu)")
    parser.add_argument("--tdnn-ksize", type=int, nargs="+", default=[5], help="TDNN kernel size")
    parser.add_argument("--timedelay", type=int, default=0, help="Delay target values by this many time steps")
    parser.add_argument("--splice", type=int, default=0, help="Splicing size")
    parser.add_argument("--dropout", "-d", type=float, nargs="+", default=[0], help="accepted for compatibility; identity at inference")
    parser.add_argument("--ft", help="Kaldi feature transform file")
    parser.add_argument("--tri", action="store_true", help="Use triphones")
    parser.add_argument("--data-dir", default="data/fmllr")
    parser.add_argument("--offset-dir", default="data")
    parser.add_argument("--ivector-dir", help="Directory with i-vector files")
    parser.add_argument("--data", default="data_{}.npy", help="Input data")
    parser.add_argument("--offsets", default="offsets_{}.npy", help="Input offsets")
    parser.add_argument("--ivectors", default="ivectors_{}.npy")
    parser.add_argument("--fold-data-dir", help="Directory with fold input data")
    parser.add_argument("--fold-output-dir", help="Directory with predicted fold output")
    parser.add_argument("--fold-model-dir", help="Directory with output fold model")
    parser.add_argument("--fold-output-dev", help="Output file with predicted development data values")
    parser.add_argument("--fold-data-pattern", default="data_{}.npy")
    parser.add_argument("--fold-offset-pattern", default="offsets_{}.npy")
    parser.add_argument("--fold-ivector-pattern", default="ivectors_{}.npy")
    parser.add_argument("--fold-output-pattern", default="data_{}.npy")
    parser.add_argument("--fold-network-pattern", default="fold_{}.npz")
    parser.add_argument("--precision", default="f32", choices=["f32", "split_f16"],
                        help="f32: exact f32 MFMA (default); split_f16: f32-accurate Linear layers on the f16 matrix pipe, ~2.3x faster (extension)")
    parser.add_argument("--no-progress", action="store_true", help="Disable progress bar")
    return parser


def load_feature_transform(args, recurrent):
    """predict_folds.py:170-188: recurrent nets keep the centre frame's transform, tdnn tiles it over its own window."""
    if args.ft is None or args.ft == "-":
        return None
    ft = loadKaldiFeatureTransform(str(Path(args.data_dir, args.ft)))
    if recurrent:
        select_centre_frame(ft)
    elif args.network == "tdnn":
        tile_centre_frame(ft, tdnn_splice(args.tdnn_ksize))
    return ft


def concat_ivectors(x, ivectors):
    """predict_folds.py:196-198 / 235-237: i-vectors are concatenated BEFORE splice and transform in this script
    (SURVEY.md M5); with a feature transform the reference then fails on a numpy broadcast -- reproduced as an error by
    ``Pipeline.set_frontend``."""
    return np.concatenate((x, ivectors), axis=1)


def main(arg_list=None):
    parser = build_parser()
    args = parser.parse_args(list(map(str, arg_list))) if arg_list is not None else parser.parse_args()

    if args.fold_output_dir is None or args.fold_model_dir is None:
        parser.error("--fold-output-dir and --fold-model-dir are required")
    out_file = Path(args.fold_output_dir, args.fold_output_dev if args.fold_output_dev is not None else args.fold_output_pattern)
    Path(out_file.parent).mkdir(exist_ok=True, parents=True)

    num_classes = 1909 if args.tri else 39
    if args.activation not in ("sigmoid", "tanh", "relu"):
        print("Wrong activation function specified")
        sys.exit(1)
    devices = parse_devices(args.gpu, args.gpus)
    set_default_precision(args.precision)
    recurrent = is_nn_recurrent(args.network)
    splice = tdnn_splice(args.tdnn_ksize) if args.network == "tdnn" else args.splice   # predict_folds.py:163-167
    winlen = 2 * splice + 1
    ft = load_feature_transform(args, recurrent)

    def fold_models():
        fold = 0
        while True:
            model_file = Path(args.fold_model_dir, args.fold_network_pattern.format(fold))
            if not model_file.is_file():
                break
            model = get_nn(args.network, args.layers, args.units, num_classes, args.activation, args.tdnn_ksize, args.dropout)
            chainer_npz.load_npz(str(model_file), model)
            yield fold, model
            fold += 1

    if args.fold_output_dev is not None:
        x = np.load(str(Path(args.data_dir, args.data.format("dev"))))
        offsets = np.load(str(Path(args.offset_dir, args.offsets.format("dev")))) if recurrent else None
        if args.ivector_dir:
            x = concat_ivectors(x, np.load(str(Path(args.ivector_dir, args.ivectors.format("dev")))))
        models = []
        for fold, model in fold_models():
            print("Predicting fold {} data".format(fold))
            models.append(model)
        if not models:
            print("Error: No fold networks found")
            sys.exit(2)
        # mean of the per-fold log-posteriors, renormalised (predict_folds.py:199-220), all members resident on the GPU
        from .networks import NNWithRPL
        predict_to_npy(Path(args.fold_output_dir, args.fold_output_dev), NNWithRPL(None, models), x, offsets, num_classes, args.network,
                       devices, winlen, args.timedelay, ft, not args.no_progress, combine=AM_COMBINE_LOGPOST)
    else:
        n_folds = 0
        for fold, model in fold_models():
            print("Predicting fold {} data".format(fold))
            x = np.load(str(Path(args.fold_data_dir, args.fold_data_pattern.format(fold))))
            offsets = np.load(str(Path(args.fold_data_dir, args.fold_offset_pattern.format(fold)))) if recurrent else None
            if args.ivector_dir:
                x = concat_ivectors(x, np.load(str(Path(args.fold_data_dir, args.fold_ivector_pattern.format(fold)))))
            predict_to_npy(Path(args.fold_output_dir, args.fold_output_pattern.format(fold)), model, x, offsets, num_classes, args.network,
                           devices, winlen, args.timedelay, ft, not args.no_progress)
            n_folds += 1
        if n_folds == 0:
            print("Error: No fold networks found")
            sys.exit(2)


if __name__ == "__main__":
    main()

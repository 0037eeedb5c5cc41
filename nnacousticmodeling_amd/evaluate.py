#!/usr/bin/env python3
"""Drop-in for the reference's ``scripts/common/evaluate.py`` (surface A: ``lab/<utt>.lab`` + ``.scp`` -> PhoneRecog -> PER).

Same CLI (evaluate.py:55-86) and ``main(arg_list)`` entry point.  The reference splices the whole file in a Python
loop, transforms it in numpy and forwards utterance by utterance through Chainer; here splice + transform + forward +
prior subtraction + log-normalisation run fused on MI355X GPUs (libamfwd.so) over the whole file, sharded over
``--gpus N`` devices (extension).

    python -m nnacousticmodeling_amd.evaluate -n ff -l 4 -u 1024 --splice 5 --tri --recog-dir recog --rpl --master-network result/model ...
"""
from __future__ import annotations

import argparse
from pathlib import Path

import numpy as np

from . import chainer_npz
from .evaluate_model_for_test import evaluateModelTestTri
from .kaldi_ft import loadKaldiFeatureTransform, select_centre_frame, tdnn_splice, tile_centre_frame
from .networks import NNWithRPL, RPL4Spec, get_nn, is_nn_recurrent  # noqa: F401  (NNWithRPL re-exported like the reference module)
from .sharding import parse_devices, set_default_precision


def build_parser():
    parser = argparse.ArgumentParser(description="MI355X evaluation (drop-in for evaluate.py)")
    parser.add_argument("--network", "-n", default="ff", help='Neural network type, either "ff" or "lstm"')
    parser.add_argument("--model", "-m", default="", help="Path to the model")
    parser.add_argument("--units", "-u", type=int, nargs="+", default=[1024], help="Number of units")
    parser.add_argument("--layers", "-l", type=int, default=2, help="Number of hidden layers")
    parser.add_argument("--activation", "-a", default="relu", help="FF activation function (sigmoid, tanh or relu)")
    parser.add_argument("--tdnn-ksize", type=int, nargs="+", default=[5], help="TDNN kernel size")
    parser.add_argument("--timedelay", type=int, default=0, help="parsed and unused, exactly as in the reference (SURVEY.md M6)")
    parser.add_argument("--splice", type=int, default=0, help="Splicing size")
    parser.add_argument("--dropout", "-d", type=float, nargs="+", default=[0], help="accepted for compatibility; identity at inference")
    parser.add_argument("--tri", action="store_true", help="Use triphones")
    parser.add_argument("--ft", default="final.feature_transform", help="Kaldi feature transform file")
    parser.add_argument("--data-dir", default="data/fmllr")
    parser.add_argument("--offset-dir", default="data")
    parser.add_argument("--ivector-dir")
    parser.add_argument("--recog-dir", required=True, help="Directory with recognizer files")
    parser.add_argument("--utt-list-dir", default="data", help="Directory with utterance lists")
    parser.add_argument("--data", default="data_{}.npy")
    parser.add_argument("--offsets", default="offsets_{}.npy")
    parser.add_argument("--ivectors", default="ivectors_{}.npy")
    parser.add_argument("--PIP", type=float, default=20)
    parser.add_argument("--LMW", type=float, default=1)
    parser.add_argument("--ap-coef", type=float, default=1)
    parser.add_argument("--ap-file", default="log_ap_Kaldi1909.npy", help="Path relative to recogdir")
    parser.add_argument("--gpu", "-g", type=int, default=0, help="first GPU ID")
    parser.add_argument("--gpus", type=int, default=1, help="number of GPUs to shard over (extension)")
    parser.add_argument("--test-or-dev", default="test")
    parser.add_argument("--rpl", action="store_true", help="Use RPL layer with folds")
    parser.add_argument("--no-rpl-layer", action="store_true", help="Disable RPL layer")
    parser.add_argument("--rpl-model", default="result_rpl/model", help="RPL layer model")
    parser.add_argument("--fold-model-dir", default="fold_models")
    parser.add_argument("--fold-network-pattern", default="fold_{0}.npz")
    parser.add_argument("--master-network", default="-", help="Master network")
    parser.add_argument("--tmp-dir", default="lab", help="where the .lab/.scp files go (the reference hard-codes 'lab')")
    parser.add_argument("--decoder", default="auto", choices=["auto", "gpu", "subprocess", "none"],
                        help="auto/gpu: in-process GPU Viterbi when <recog-dir>/kaldiTri1909.img exists; subprocess: the reference's PhoneRecog executable")
    parser.add_argument("--no-lab", action="store_true", help="do not write the .lab/.scp posterior files (only the external executable needs them)")
    parser.add_argument("--precision", default="f32", choices=["f32", "split_f16"],
                        help="f32: exact f32 MFMA (default); split_f16: f32-accurate Linear layers on the f16 matrix pipe, ~2.3x faster (extension)")
    parser.add_argument("--no-progress", action="store_true", help="Disable progress bar")
    return parser


def load_models(args, num_classes):
    """evaluate.py:105-136."""
    def new_net():
        return get_nn(args.network, args.layers, args.units, num_classes, args.activation, args.tdnn_ksize, args.dropout)

    if not args.rpl:
        return chainer_npz.load_npz(args.model, new_net())
    master = None
    if args.master_network != "-":
        print("Loading master network")
        master = chainer_npz.load_npz(args.master_network, new_net())
    folds = []
    if args.fold_network_pattern != "-":
        fold = 0
        while True:
            model_file = Path(args.fold_model_dir, args.fold_network_pattern.format(fold))
            if not model_file.is_file():
                break
            print("Loading fold {} network".format(fold))
            folds.append(chainer_npz.load_npz(str(model_file), new_net()))
            fold += 1
    rpl = None
    if args.rpl_model != "-":
        rpl = chainer_npz.load_npz(args.rpl_model, RPL4Spec(num_classes))
    return NNWithRPL(master, folds, rpl)


def main(arg_list=None):
    parser = build_parser()
    args = parser.parse_args(list(map(str, arg_list))) if arg_list is not None else parser.parse_args()
    num_classes = 1909 if args.tri else 39
    if args.activation not in ("sigmoid", "tanh", "relu"):
        print("Wrong activation function specified")
        return None
    devices = parse_devices(args.gpu, args.gpus)
    set_default_precision(args.precision)
    model = load_models(args, num_classes)
    recurrent = is_nn_recurrent(args.network)

    ft = None
    if args.ft is not None and args.ft != "-":
        ft = loadKaldiFeatureTransform(str(Path(args.data_dir, args.ft)))
        if recurrent:
            select_centre_frame(ft)
        elif args.network == "tdnn":
            tile_centre_frame(ft, tdnn_splice(args.tdnn_ksize))   # evaluate.py:152-159

    data = np.load(str(Path(args.data_dir, args.data.format(args.test_or_dev))))
    offsets = np.load(str(Path(args.offset_dir, args.offsets.format(args.test_or_dev))))
    ivectors = None
    if args.ivector_dir is not None:
        ivectors = np.load(str(Path(args.ivector_dir, args.ivectors.format(args.test_or_dev))))
    # evaluate.py:164-171 order: splice over the whole file -> transform -> THEN append i-vectors; done on the GPU
    splice = tdnn_splice(args.tdnn_ksize) if args.network == "tdnn" else args.splice   # evaluate.py:138-141
    shifts = list(range(-splice, splice + 1)) if splice > 0 else [0]

    if not args.tri:
        print("Monophones not implemented")
        return None
    ap = args.ap_coef * np.load(str(Path(args.recog_dir, args.ap_file)))
    per = evaluateModelTestTri(model, data, offsets, args.PIP, args.LMW, ap=ap, testOrDev=args.test_or_dev, tmpDir=args.tmp_dir,
                               uttlistdir=args.utt_list_dir, recogdir=args.recog_dir, GPUID=devices, progress=not args.no_progress,
                               rnn=recurrent, frontend=(shifts, ft), ivectors=ivectors,
                               run_decoder=(False if args.decoder == "none" else args.decoder), write_lab=not args.no_lab)
    print("PER: {0:.2f} %".format(per))
    return per


if __name__ == "__main__":
    main()

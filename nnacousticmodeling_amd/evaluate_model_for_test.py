"""Posterior writer + scorer: drop-in for the reference's ``scripts/util/evaluateModelForTest.py`` (surface A).

``evaluateModelTestTri`` keeps the reference's signature and side effects (evaluateModelForTest.py:36-133): per
utterance ``lab/<utt>.lab`` in ``saveBin`` format holding ``y - ap`` re-normalised with ``y - logsum(y)``, one
``lab/<test|dev>.scp`` line per utterance in list order, then the PhoneRecog subprocess and the PER computation.
The network forward, prior subtraction and log-normalisation run on the GPU over the whole file at once (rows are
independent, so batching per utterance as the reference does gives the same rows).
"""
from __future__ import annotations

import os
import subprocess
from pathlib import Path

import numpy as np

from .kw_io import saveBin
from .networks import as_ensemble
from .sharding import get_runner


def compute_posteriors(model, data, offsets, ap=None, devices=(0,), rnn=False, frontend=None, ivectors=None):
    """float32 ``[N, C]`` log-posteriors (minus priors, renormalised) for the whole file.

    ``frontend=None``: ``data`` is what the reference passes -- already spliced / transformed / i-vector-extended
    (evaluate.py:163-171) -- and is fed to the network as is.  ``frontend=(shifts, ft)``: ``data`` holds the raw frames
    and the GPU applies splice + transform (+ ``ivectors`` appended after the transform) fused in one pass."""
    runner = get_runner(devices)
    ens = as_ensemble(model)
    shifts, ft = ([0], None) if frontend is None else frontend
    if rnn:
        if len(list(shifts)) != 1:
            raise ValueError("operands could not be broadcast together: recurrent networks take un-spliced frames")
        return runner.forward_lstm(ens, data, offsets, ft=ft, ivectors=ivectors, timedelay=0, prior=ap)
    return runner.forward_ff(ens, data, shifts=shifts, ft=ft, ivectors=ivectors, prior=ap)


def write_posteriors(y, offsets, testList, labDirOut, testOrDev):
    """``.lab`` per utterance + the ``.scp`` list, byte-compatible with the reference (evaluateModelForTest.py:43-45,
    88-94, 113-117).  Returns the scp path."""
    labDirOut = Path(labDirOut)
    labDirOut.mkdir(exist_ok=True, parents=True)
    scp = Path(labDirOut, testOrDev + ".scp")
    with open(str(scp), "wt") as fscp:
        for i, f in enumerate(testList):
            f = f.strip()
            labout = Path(labDirOut, f + ".lab")
            saveBin(str(labout), y[offsets[i]:offsets[i + 1], :])
            fscp.write(str(labout) + "\n")
    return scp


def evaluateModelTestTri(model, data, offsets, PIP, LMW, ap=None, GPUID=0, testOrDev="test", tmpDir="lab", uttlistdir=".",
                         recogdir=".", progress=True, rnn=False, frontend=None, ivectors=None, run_decoder="auto", write_lab=True):
    """Returns the PER (float); -1 when the utterance list and the offsets disagree, as the reference does.

    Differences from the reference, all on the host side: ``GPUID`` may be a list of device ids (negative = CPU is
    refused: no CPU fallback); ``frontend`` / ``ivectors`` select the fused on-GPU frontend (see
    ``compute_posteriors``).  ``run_decoder``: "auto" / "gpu" decode in-process on the GPU (``phone_recog.PhoneRecog``,
    same segmentations as the reference binary) when ``kaldiTri1909.img`` exists under ``recogdir``; "subprocess" runs the
    reference's ``PhoneRecog`` executable as the reference does; without the graph image "auto" stops after writing the
    posteriors and returns NaN.  ``write_lab=False`` skips the ``.lab``/``.scp`` files (442 MB for the core test set),
    which only the external executable needs.

    With several GPUs every shard owns whole utterances (frame ranges cut at utterance boundaries, or utterance bins for
    recurrent nets), writes their ``.lab`` files from its own page-locked block and decodes them on its own GPU straight from
    device memory (``PosteriorRunner.map_shards``): nothing is gathered into one array, and with ``write_lab=False`` the
    posteriors never leave the devices."""
    testList = open(uttlistdir + "/" + testOrDev + ".list").readlines()
    if len(testList) != len(offsets) - 1:
        print("Error: wrong number of utterances")
        return -1
    devices = list(GPUID) if isinstance(GPUID, (list, tuple)) else [GPUID]
    if progress:
        print("Calculating network outputs")
    img = str(Path(recogdir, "kaldiTri1909.img"))
    have_img = os.path.isfile(img)
    in_process = bool(run_decoder) and run_decoder in ("auto", "gpu") and have_img
    need_lab = write_lab or run_decoder == "subprocess"
    labDirOut = Path(tmpDir)
    labDirOut.mkdir(exist_ok=True, parents=True)
    offsets = np.asarray(offsets, dtype=np.int64)
    names = [f.strip() for f in testList]
    shifts, ft = ([0], None) if frontend is None else frontend
    if rnn and len(list(shifts)) != 1:
        raise ValueError("operands could not be broadcast together: recurrent networks take un-spliced frames")

    def make_decoder(engine):
        from .phone_recog import PhoneRecog
        return PhoneRecog(engine, img, -abs(PIP), LMW)

    def per_shard(ctx):
        """This shard's utterances: .lab files from its own host block, segmentations from its own GPU."""
        segs = None
        if in_process:   # first: the decode launches run on the device while the host part below writes files
            segs = ctx.decoder(("viterbi", img, float(PIP), float(LMW)), make_decoder).recognise_pipeline(ctx.pipeline, ctx.local_offsets)
        if need_lab:
            y = ctx.download()
            for j, u in enumerate(ctx.utts):
                saveBin(str(Path(labDirOut, names[u] + ".lab")), y[ctx.local_offsets[j]:ctx.local_offsets[j + 1], :])
        return ctx.utts, segs

    runner = get_runner(devices)
    shard_results = runner.map_shards(as_ensemble(model), data, per_shard, offsets=offsets, shifts=shifts, ft=ft, ivectors=ivectors,
                                      prior=ap, recurrent=rnn, timedelay=0, host=need_lab)
    scp = Path(labDirOut, testOrDev + ".scp")
    if need_lab:
        if progress:
            print("Writing output files")
        with open(str(scp), "wt") as fscp:   # list order, as the reference writes it (evaluateModelForTest.py:43-45,91-93,114-116)
            for f in names:
                fscp.write(str(Path(labDirOut, f + ".lab")) + "\n")

    recog = recogdir + "/PhoneRecog" + (".exe" if os.name == "nt" else "")
    result_txt = str(Path(labDirOut, "vysledek_" + testOrDev + ".txt"))
    if not run_decoder:
        return float("nan")
    if not have_img:
        if run_decoder == "auto":
            print("kaldiTri1909.img not found under %s: posteriors written to %s, decoding skipped" % (recogdir, scp))
            return float("nan")
        raise FileNotFoundError(img)
    from .scoring import computeWER, convert_results, loadMlf
    if run_decoder == "subprocess":
        cmd = [recog, str(scp), img, result_txt, str(-abs(PIP)), str(LMW)]
        subprocess.run(cmd, cwd=os.getcwd())
    else:
        from .phone_recog import base_name_copy, write_mlf
        results = [None] * len(names)
        for utts, segs in shard_results:
            for u, sg in zip(utts, segs):
                results[int(u)] = sg
        write_mlf(result_txt, [base_name_copy(str(Path(labDirOut, f + ".lab"))) for f in names], results)
    p39 = str(Path(labDirOut, "vysledek_" + testOrDev + "_p39.txt"))
    convert_results(str(Path(recogdir, "phones.60-48-39.map")), result_txt, p39)
    ref = loadMlf(str(Path(recogdir, testOrDev + "_ref.mlf")))
    result = loadMlf(p39)
    return computeWER(result, ref, True)

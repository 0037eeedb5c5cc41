"""Kaldi nnet1 ``final.feature_transform`` reader (host side of the fused splice+transform kernel).

Mirrors ``loadKaldiFeatureTransform`` (reference scripts/util/kw_nn_utils.py:4-11) and the centre-frame slicing the
reference applies for recurrent nets (scripts/common/evaluate.py:145-151, predict_folds.py:172-178).  The transform
itself -- ``(x + addShift) * rescale`` (kw_nn_utils.py:13-17) -- runs on the GPU, fused with the splice.
"""
from __future__ import annotations

import numpy as np


def loadKaldiFeatureTransform(filename):
    """Same line-indexed parse and the same dict keys as the reference: shape, shifts, addShift, rescale."""
    with open(filename) as fid:
        lines = fid.readlines()
    if len(lines) < 7:
        raise ValueError("%s: not a <Splice>/<AddShift>/<Rescale> feature transform (only %d lines)" % (filename, len(lines)))
    ft = {}
    ft["shape"] = [int(tok) for tok in lines[1].split()[1:]]
    ft["shifts"] = [int(tok) for tok in lines[2].split()[1:-1]]
    ft["addShift"] = np.asarray([float(tok) for tok in lines[4].split()[3:-1]], dtype=np.float32)
    ft["rescale"] = np.asarray([float(tok) for tok in lines[6].split()[3:-1]], dtype=np.float32)
    if ft["addShift"].shape != (ft["shape"][0],) or ft["rescale"].shape != (ft["shape"][0],):
        raise ValueError("%s: <AddShift>/<Rescale> length does not match <Splice> %d" % (filename, ft["shape"][0]))
    return ft


def select_centre_frame(ft):
    """In place, as the reference does for recurrent networks: keep the transform of the un-shifted frame only."""
    dim = ft["shape"][1]
    zi = ft["shifts"].index(0)
    ft["rescale"] = ft["rescale"][zi * dim:(zi + 1) * dim]
    ft["addShift"] = ft["addShift"][zi * dim:(zi + 1) * dim]
    ft["shape"][0] = dim
    ft["shifts"] = [0]
    return ft


def tile_centre_frame(ft, splice):
    """In place, as the reference does for ``tdnn`` (evaluate.py:152-159, predict_folds.py:179-186): the un-shifted
    frame's transform repeated over the 2*splice+1 frames of the network's own window."""
    dim = ft["shape"][1]
    zi = ft["shifts"].index(0)
    winlen = 2 * splice + 1
    ft["rescale"] = np.tile(ft["rescale"][zi * dim:(zi + 1) * dim], winlen)
    ft["addShift"] = np.tile(ft["addShift"][zi * dim:(zi + 1) * dim], winlen)
    ft["shape"][0] = dim * winlen
    ft["shifts"] = list(range(-splice, splice + 1))
    return ft


def tdnn_splice(ksize):
    """Frames of context a TDNN needs on each side: (sum(ksize) - len(ksize)) // 2 (evaluate.py:138-139)."""
    return (sum(ksize) - len(ksize)) // 2

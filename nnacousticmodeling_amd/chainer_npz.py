"""Chainer ``save_npz`` / ``load_npz`` model files without Chainer.

``serializers.save_npz(path, L.Classifier(model))`` (reference scripts/common/train.py:393-395) writes a numpy ``.npz``
whose keys are link paths joined by ``/`` under the ``predictor/`` prefix:
  ff   : predictor/layer_{l}/W [units,in], predictor/layer_{l}/b [units], predictor/out/W [C,units], predictor/out/b [C]
  lstm : predictor/layer_{l}/upward/W [4U,in], predictor/layer_{l}/upward/b [4U], predictor/layer_{l}/lateral/W [4U,U]
  tdnn : predictor/layer_{l}/W [units_l, in_channels_l, 1, k_l] (L.Convolution2D), predictor/layer_{l}/b, out/W [C, units_last]
  gru / mgrurelu[r] : predictor/layer_{l}/{W_r,U_r,W_z,U_z,W,U}/{W,b}  (MGRU.py:10-32; no W_r/U_r without reset gate)
  peepholelstm : the lstm keys + predictor/layer_{l}/{peep_i,peep_f,peep_o}/W [U,U]
  RPL4 : predictor/W, predictor/b, predictor/lb  each [1,C]
The final model is saved as ``{out}/model`` with no ``.npz`` suffix; fold models as ``fold_{k}.npz``.
Only numpy's non-pickling loader is used.
"""
from __future__ import annotations

import numpy as np

PREFIX = "predictor/"


def read_npz(path, prefix=PREFIX):
    """-> {key without prefix: float32 array}."""
    with np.load(str(path), allow_pickle=False) as z:
        return {k[len(prefix):]: np.asarray(z[k]) for k in z.files if k.startswith(prefix)}


def write_npz(path, params, prefix=PREFIX):
    """Write to the exact filename (Chainer opens the file itself, so no suffix is appended)."""
    with open(str(path), "wb") as fid:
        np.savez_compressed(fid, **{prefix + k: np.asarray(v) for k, v in params.items()})


def load_npz(path, target):
    """Counterpart of ``chainer.serializers.load_npz(path, L.Classifier(model))``: fill ``target`` (a
    ``networks.NetworkSpec`` or ``networks.RPL4Spec``) from the file and check the shapes the way Chainer would."""
    params = read_npz(path)
    if not params:
        raise KeyError("%s holds no 'predictor/...' entries: not a model saved through L.Classifier" % path)
    target.set_params(params, source=str(path))
    return target

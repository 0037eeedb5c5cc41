"""Host-side network descriptions mirroring the reference's model zoo (every get_nn network).

``get_nn`` / ``is_nn_recurrent`` follow scripts/common/chainer_networks.py:163-187; ``NNWithRPL`` follows
scripts/common/evaluate.py:19-51; ``RPL4`` follows scripts/common/RPL.py:58-74.  The objects here hold host copies of
the parameters only; all arithmetic runs on the GPU (``sharding.PosteriorRunner`` uploads them through the C ABI).
"""
from __future__ import annotations

import numpy as np

from ._lib import AM_CELL_GRU, AM_CELL_MGRU, AM_CELL_PEEPHOLE

SUPPORTED = ("ff", "tdnn", "lstm", "zoneoutlstm", "zoneoutdropoutlstm", "gru", "mgrurelu", "mgrurelur", "peepholelstm")
# At inference (chainer.config.train = False) zoneout returns the new state, so L.StatefulZoneoutLSTM computes exactly
# what L.LSTM computes (upward Linear + bias-free lateral Linear, interleaved gates, h = None -> zeros on the first step)
# and its npz keys are the same; both zoneout variants of the reference (chainer_networks.py:64-101) therefore run on the
# LSTM device path.  Like every network row, this rests on Chainer 3.5 semantics that cannot be pinned here.
LSTM_EQUIVALENT = ("lstm", "zoneoutlstm", "zoneoutdropoutlstm")
# The other recurrent cells of the reference's get_nn (chainer_networks.py:174-181).  Per network: the device cell, the
# candidate activation get_nn hard-wires (L.GRU: tanh; NetMGRU: F.relu whatever --activation says), and the per-layer npz
# keys in the order am_model_create_rnn takes them.  MGRU.py:10-32 names the Linears W_r/U_r (reset), W_z/U_z (update), W/U.
_GRU_KEYS = tuple("%s/%s" % (n, k) for n in ("W_r", "U_r", "W_z", "U_z", "W", "U") for k in ("W", "b"))
RNN_CELLS = {
    "gru": (AM_CELL_GRU, "tanh", _GRU_KEYS),
    "mgrurelur": (AM_CELL_GRU, "relu", _GRU_KEYS),
    "mgrurelu": (AM_CELL_MGRU, "relu", _GRU_KEYS[4:]),
    "peepholelstm": (AM_CELL_PEEPHOLE, "tanh", ("upward/W", "upward/b", "lateral/W", "peep_i/W", "peep_f/W", "peep_o/W")),
}
ALL_REFERENCE_NETWORKS = ("ff", "tdnn", "lstm", "zoneoutlstm", "zoneoutdropoutlstm", "peepholelstm", "gru", "mgrurelu", "mgrurelur")


def is_nn_recurrent(n):
    return n.endswith("lstm") or n.startswith("gru") or n.startswith("mgru")


class NetworkSpec:
    """An un-initialised network, as ``get_nn`` returns it; ``chainer_npz.load_npz`` fills the parameters."""

    def __init__(self, network, layers, units, num_classes, activation="relu", tdnn_ksize=None):
        self.reference_network = network                                   # the name given on the command line
        network = "lstm" if network in LSTM_EQUIVALENT else network        # the device path that runs it
        self.network = network
        if network == "tdnn":   # TDNN(units, num_classes, tdnn_ksize, ...): one conv layer per entry, --layers is ignored (chainer_networks.py:24-36)
            self.units_list = [int(u) for u in units]
            self.ksize = [int(k) for k in tdnn_ksize]
            if len(self.units_list) != len(self.ksize):
                raise ValueError("TDNN n_units argument must have the same length as ksize")
            layers, units = len(self.units_list), self.units_list[-1]
            self.input_win_size = sum(self.ksize) - len(self.ksize) + 1
        self.layers = int(layers)
        self.units = int(units)
        self.num_classes = int(num_classes)
        self.activation = activation
        self.params = None
        self.source = None

    @property
    def recurrent(self):
        return is_nn_recurrent(self.network)

    def reset_state(self):  # the device path keeps no state between calls
        pass

    def expected_keys(self):
        keys = []
        for l in range(self.layers):
            if self.network in ("ff", "tdnn"):
                keys += ["layer_%d/W" % l, "layer_%d/b" % l]
            elif self.network in RNN_CELLS:
                keys += ["layer_%d/%s" % (l, k) for k in RNN_CELLS[self.network][2]]
            else:
                keys += ["layer_%d/upward/W" % l, "layer_%d/upward/b" % l, "layer_%d/lateral/W" % l]
        return keys + ["out/W", "out/b"]

    def set_params(self, params, source=None):
        """Shape checks equivalent to what ``load_npz`` enforces: biases and lateral weights are created with the
        declared sizes (``L.Linear(None, n_units)``, ``L.LSTM(None, n_units)``), only input widths come from the file."""
        missing = [k for k in self.expected_keys() if k not in params]
        if missing:
            raise KeyError("%s: missing parameter(s) %s for a %d-layer %s network" % (source, missing, self.layers, self.network))
        p = {k: np.ascontiguousarray(params[k], dtype=np.float32) for k in self.expected_keys()}
        for l in range(self.layers):
            if self.network == "tdnn":   # L.Convolution2D(None, units, ksize=(1, k)): W [units, in_channels, 1, k]
                w, b = p["layer_%d/W" % l], p["layer_%d/b" % l]
                cin = self.units_list[l - 1] if l > 0 else (w.shape[1] if w.ndim == 4 else -1)
                if w.shape != (self.units_list[l], cin, 1, self.ksize[l]) or b.shape != (self.units_list[l],):
                    raise ValueError("%s: layer_%d/W has shape %s, expected %s" % (source, l, w.shape, (self.units_list[l], cin, 1, self.ksize[l])))
                continue
            if self.network in RNN_CELLS and self.network != "peepholelstm":
                # every Linear of a GRU / MGRU link has `units` outputs; the state Linears are units x units
                for k in RNN_CELLS[self.network][2]:
                    v, name = p["layer_%d/%s" % (l, k)], k.split("/")[0]
                    ok = v.shape == (self.units,) if k.endswith("/b") else (v.shape[0] == self.units and (name.startswith("W") or v.shape[1] == self.units))
                    if not ok:
                        raise ValueError("%s: layer_%d/%s has shape %s but --units says %d" % (source, l, k, v.shape, self.units))
                continue
            if self.network == "ff":
                w, b = p["layer_%d/W" % l], p["layer_%d/b" % l]
                want = self.units
            else:
                w, b = p["layer_%d/upward/W" % l], p["layer_%d/upward/b" % l]
                want = 4 * self.units
                lat = p["layer_%d/lateral/W" % l]
                if lat.shape != (4 * self.units, self.units):
                    raise ValueError("%s: layer_%d/lateral/W has shape %s, expected %s" % (source, l, lat.shape, (4 * self.units, self.units)))
                if self.network == "peepholelstm":
                    for k in ("peep_i/W", "peep_f/W", "peep_o/W"):
                        if p["layer_%d/%s" % (l, k)].shape != (self.units, self.units):
                            raise ValueError("%s: layer_%d/%s has shape %s, expected %s" % (source, l, k, p["layer_%d/%s" % (l, k)].shape, (self.units, self.units)))
            if b.shape != (want,) or w.shape[0] != want:
                raise ValueError("%s: layer %d holds %d outputs but --units says %d" % (source, l, b.shape[0], want))
        if p["out/b"].shape != (self.num_classes,) or p["out/W"].shape[0] != self.num_classes:
            raise ValueError("%s: output layer has %d classes, expected %d" % (source, p["out/b"].shape[0], self.num_classes))
        self.params = p
        self.source = source
        return self

    @property
    def in_dim(self):
        key = "layer_0/W" if self.network in ("ff", "tdnn") else "layer_0/W_z/W" if self.network in RNN_CELLS and self.network != "peepholelstm" else "layer_0/upward/W"
        if self.layers == 0:
            key = "out/W"
        if self.network == "tdnn":
            return int(self.params[key].shape[1]) * self.input_win_size
        return int(self.params[key].shape[1])


class RPL4Spec:
    """RPL4 parameters W, b, lb, each [1, C] (RPL.py:58-66)."""

    def __init__(self, num_classes):
        self.num_classes = int(num_classes)
        self.W = np.zeros((1, num_classes), dtype=np.float32)
        self.b = np.zeros((1, num_classes), dtype=np.float32)
        self.lb = np.full((1, num_classes), -20.0, dtype=np.float32)

    def set_params(self, params, source=None):
        for k in ("W", "b", "lb"):
            if k not in params:
                raise KeyError("%s: missing RPL parameter %r" % (source, k))
            v = np.ascontiguousarray(params[k], dtype=np.float32).reshape(1, -1)
            if v.shape[1] != self.num_classes:
                raise ValueError("%s: RPL parameter %s has %d entries, expected %d" % (source, k, v.shape[1], self.num_classes))
            setattr(self, k, v)
        return self


def get_nn(network, layers, units, num_classes, activation="relu", tdnn_ksize=None, dropout=(0,)):
    """Same argument order as the reference ``get_nn``.  ``activation`` is a name (relu/sigmoid/tanh); dropout is
    accepted and ignored (identity at inference, evaluate.py:94).  Every network of the reference's get_nn is built."""
    if network in SUPPORTED:
        if activation not in ("relu", "sigmoid", "tanh"):
            raise ValueError("Wrong activation function specified")
        if network == "tdnn":
            return NetworkSpec(network, layers, list(units), num_classes, activation, list(tdnn_ksize))
        return NetworkSpec(network, layers, units[0] if isinstance(units, (list, tuple)) else units, num_classes, activation)
    if network in ALL_REFERENCE_NETWORKS:
        raise NotImplementedError("network %r exists in the reference but is not built here yet (supported: %s)" % (network, ", ".join(SUPPORTED)))
    raise ValueError("Wrong network type specified")


class NNWithRPL:
    """Ensemble wrapper with the semantics of the reference class of the same name (evaluate.py:19-51):
    master only -> master(x); master + K folds -> (K*master + sum folds)/(2K) on raw logits; folds only -> mean of
    logits; then the optional RPL4 layer."""

    def __init__(self, master=None, folds=(), rpl=None):
        self.master = master
        self.folds = list(folds)
        self.num_folds = len(self.folds)
        self.rpl = rpl
        if master is None and not self.folds:
            raise ValueError("NNWithRPL needs a master network or at least one fold network")

    def reset_state(self):
        pass

    def members(self):
        return ([self.master] if self.master is not None else []) + self.folds

    @property
    def has_master(self):
        return self.master is not None

    @property
    def recurrent(self):
        return self.members()[0].recurrent

    @property
    def num_classes(self):
        return self.members()[0].num_classes


def as_ensemble(model):
    """A bare NetworkSpec behaves like NNWithRPL(master=spec)."""
    return model if isinstance(model, NNWithRPL) else NNWithRPL(master=model)

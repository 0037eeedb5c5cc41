"""CPU oracle for the posterior-compute path of OrcusCZ/NNAcousticModeling.

TEST INFRASTRUCTURE ONLY.  Nothing under ``nnacousticmodeling_amd/`` may import
this module: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

It is a plain-numpy restatement of the reference algorithm.  Every function
cites the reference file:line (relative to the reference repo root) it follows.

Parity pinning
--------------
* ``splice`` / ``prepare_batch`` / ``apply_feature_transform`` / ``logsum`` /
  ``load_kaldi_feature_transform`` / ``save_bin`` / ``load_bin`` are pinned against
  the reference's own numpy utilities (``scripts/util/kw_utils.py``,
  ``scripts/util/kw_nn_utils.py``), which import cleanly in the build container:
  ``tests/golden/make_golden.py`` ran them and froze inputs + outputs in
  ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this module
  against those vectors.
* The network arithmetic (``MLP``, ``LSTM``, ``RPL4``, ``NNWithRPL`` and the two
  time loops) lives in Chainer 3.5, which is absent from the reference tree and
  from this image, and the reference ships no test, model or golden vector for it:
  **parity unpinned** for those rows.  They restate Chainer's published CPU
  semantics and are anchored on the reference's call sites.

``dtype`` arguments let the tests run a float64 shadow for error budgeting.
"""
from __future__ import annotations

import io
import zipfile

import numpy as np

NUM_CLASSES_TRI = 1909  # scripts/common/evaluate.py:92


# --------------------------------------------------------------------------
# Kaldi feature transform (scripts/util/kw_nn_utils.py:4-17)
# --------------------------------------------------------------------------
def load_kaldi_feature_transform(filename):
    """Line-indexed parse of a Kaldi nnet1 ``final.feature_transform`` text file.

    Follows ``loadKaldiFeatureTransform`` (kw_nn_utils.py:4-11): line 1 ``<Splice> out in``,
    line 2 ``[ shifts ]``, line 4 tokens ``[3:-1]`` = addShift, line 6 tokens ``[3:-1]`` = rescale.
    """
    with open(filename) as fid:
        s = fid.readlines()
    shape = [int(t) for t in s[1].split()[1:]]
    shifts = [int(t) for t in s[2].split()[1:-1]]
    add_shift = np.asarray([float(t) for t in s[4].split()[3:-1]], dtype=np.float32)
    rescale = np.asarray([float(t) for t in s[6].split()[3:-1]], dtype=np.float32)
    assert add_shift.shape == (shape[0],) and rescale.shape == (shape[0],)
    return {"shape": shape, "shifts": shifts, "addShift": add_shift, "rescale": rescale}


def centre_frame_transform(ft):
    """Recurrent nets keep only the centre frame of the transform (evaluate.py:145-151)."""
    dim = ft["shape"][1]
    zi = ft["shifts"].index(0)
    return {
        "shape": [dim, dim],
        "shifts": [0],
        "addShift": ft["addShift"][zi * dim:(zi + 1) * dim].copy(),
        "rescale": ft["rescale"][zi * dim:(zi + 1) * dim].copy(),
    }


def tdnn_frame_transform(ft, ksize):
    """TDNN: the centre frame's transform tiled over the network's own window of 2*splice+1 frames,
    splice = (sum(ksize) - len(ksize)) // 2 (evaluate.py:138-139,152-159)."""
    dim = ft["shape"][1]
    zi = ft["shifts"].index(0)
    splice_ctx = (sum(ksize) - len(ksize)) // 2
    winlen = 2 * splice_ctx + 1
    return {
        "shape": [dim * winlen, dim],
        "shifts": list(range(-splice_ctx, splice_ctx + 1)),
        "addShift": np.tile(ft["addShift"][zi * dim:(zi + 1) * dim], winlen),
        "rescale": np.tile(ft["rescale"][zi * dim:(zi + 1) * dim], winlen),
    }


def apply_feature_transform(x, ft):
    """``(x + addShift) * rescale`` -- add first, then multiply (kw_nn_utils.py:13-17)."""
    x1 = x + ft["addShift"]
    return x1 * ft["rescale"]


# --------------------------------------------------------------------------
# Splicing (scripts/util/kw_utils.py:24-36, kw_nn_utils.py:19-43)
# --------------------------------------------------------------------------
def splice(data, shifts):
    """``out[i, w*M:(w+1)*M] = data[clamp(i+shift_w, 0, N-1)]``.

    The clamp is global over the whole file, i.e. it crosses utterance boundaries
    (kw_utils.py:24-36 is called on the full array at evaluate.py:163-165).  Output is float32
    as in the reference (``np.zeros(..., dtype=np.float32)``).
    """
    data = np.asarray(data)
    n, m = data.shape
    shifts = list(shifts)
    out = np.zeros((n, m * len(shifts)), dtype=np.float32)
    rows = np.arange(n)
    for wi, w in enumerate(shifts):
        out[:, wi * m:(wi + 1) * m] = data[np.clip(rows + w, 0, n - 1)]
    return out


def splice_rowloop(data, shifts):
    """The same result computed the way the reference computes it: a Python loop over rows and shifts with a per-row clamp
    (kw_utils.py:24-36).  Kept only so the CPU baseline can show where the reference's time goes (bench.py times it on a
    3,000-frame sample); `splice` above is what the parity tests use."""
    data = np.asarray(data)
    n, m = data.shape
    shifts = list(shifts)
    out = np.zeros((n, m * len(shifts)), dtype=np.float32)
    for i in range(n):
        for wi, w in enumerate(shifts):
            j = min(max(i + w, 0), n - 1)
            out[i][wi * m:(wi + 1) * m] = data[j]
    return out


def prepare_batch(data, idxs, winlen):
    """Splice a batch of row indices with window ``winlen`` (kw_nn_utils.py:19-43).

    ``winhalf = int(winlen/2)``; rows near the file ends are clamped, all others are a flat
    slice -- numerically identical to ``splice(data, range(-winhalf, winhalf+1))[idxs]``.
    """
    winhalf = int(winlen / 2)
    idxs = np.sort(np.asarray(idxs))
    num, dim = data.shape
    out = np.zeros((len(idxs), winlen * dim), dtype=np.float32)
    for wi, w in enumerate(range(-winhalf, winhalf + 1)):
        out[:, wi * dim:(wi + 1) * dim] = data[np.clip(idxs + w, 0, num - 1)]
    return out


# --------------------------------------------------------------------------
# logsum / file formats (scripts/util/kw_utils.py:4-22, 38-43)
# --------------------------------------------------------------------------
def logsum(lp, axis=1):
    """Row log-sum-exp with the reference's NaN rule (kw_utils.py:38-43).

    ``mx + log(sum(exp(lp - mx)))``; NaN results (rows of all -inf, rows holding +inf or NaN)
    become ``-1e20``.
    """
    inf = 1e20
    with np.errstate(all="ignore"):
        mx = np.max(lp, axis=axis).reshape([lp.shape[0], 1])
        lps = mx + np.log(np.sum(np.exp(lp - mx), axis=axis)).reshape([lp.shape[0], 1])
    lps[np.isnan(lps)] = -inf
    return lps


def log_normalise(y):
    """``y - logsum(y, axis=1)`` as used at predict_folds.py:57,88,219 and evaluateModelForTest.py:77,112."""
    return y - logsum(y, axis=1)


def save_bin(filename, x):
    """``uint32 dims[2]`` + raw array bytes; 1-D input is written as ``[n, 1]`` (kw_utils.py:4-12)."""
    dims = np.asarray(x.shape, dtype=np.uint32)
    if len(dims) == 1:
        dims = np.resize(dims, 2)
        dims[1] = 1
    with open(filename, "wb") as fid:
        dims.tofile(fid, sep="")
        x.tofile(fid, sep="")


def load_bin(filename, dtype=np.float32):
    """Inverse of ``save_bin`` (kw_utils.py:14-22)."""
    with open(filename, "rb") as fid:
        dims = np.fromfile(fid, dtype=np.uint32, sep="", count=2)
        x = np.fromfile(fid, dtype=dtype, sep="")
    if dims[1] > 1:
        x = x.reshape(dims)
    return x


# --------------------------------------------------------------------------
# Chainer 3.5 CPU element-wise semantics (parity unpinned)
# --------------------------------------------------------------------------
def _sigmoid(x):
    """Chainer's CPU sigmoid: ``tanh(x * 0.5) * 0.5 + 0.5`` in the input dtype."""
    half = x.dtype.type(0.5)
    return np.tanh(x * half) * half + half


def _activation(name):
    if name == "relu":
        return lambda x: np.maximum(x, x.dtype.type(0))
    if name == "sigmoid":
        return _sigmoid
    if name == "tanh":
        return np.tanh
    raise ValueError("Wrong activation function specified")  # evaluate.py:102-103


def _linear(x, w, b):
    """``L.Linear``: ``y = x @ W.T + b`` with ``W`` stored ``[out, in]``."""
    y = x.dot(w.T)
    if b is not None:
        y += b
    return y


class MLP:
    """``MLP.__call__`` (chainer_networks.py:8-22): L x act(Linear) then ``out`` Linear.

    Dropout is the identity because ``chainer.config.train = False`` (evaluate.py:94).
    ``params`` uses the Chainer npz keys without the ``predictor/`` prefix.
    """

    recurrent = False

    def __init__(self, params, layers, activation="relu", dtype=np.float32):
        self.layers = layers
        self.act = _activation(activation)
        self.dtype = dtype
        self.w = [np.asarray(params["layer_%d/W" % l], dtype=dtype) for l in range(layers)]
        self.b = [np.asarray(params["layer_%d/b" % l], dtype=dtype) for l in range(layers)]
        self.w_out = np.asarray(params["out/W"], dtype=dtype)
        self.b_out = np.asarray(params["out/b"], dtype=dtype)

    def reset_state(self):
        pass

    def __call__(self, x):
        x = np.asarray(x, dtype=self.dtype)
        for w, b in zip(self.w, self.b):
            x = self.act(_linear(x, w, b))
        return _linear(x, self.w_out, self.b_out)


class LSTM:
    """Stack of stateful ``L.LSTM`` links + ``out`` (chainer_networks.py:44-62).

    One call = one time step for the whole batch.  Gate pre-activations are interleaved:
    ``g.reshape(B, U, 4)`` -> a, i, f, o; ``c' = tanh(a) sig(i) + sig(f) c``; ``h' = sig(o) tanh(c')``.
    First call after ``reset_state``: lateral term skipped (h is None) and c = 0.
    """

    recurrent = True

    def __init__(self, params, layers, dtype=np.float32):
        self.layers = layers
        self.dtype = dtype
        self.w_up = [np.asarray(params["layer_%d/upward/W" % l], dtype=dtype) for l in range(layers)]
        self.b_up = [np.asarray(params["layer_%d/upward/b" % l], dtype=dtype) for l in range(layers)]
        self.w_lat = [np.asarray(params["layer_%d/lateral/W" % l], dtype=dtype) for l in range(layers)]
        self.w_out = np.asarray(params["out/W"], dtype=dtype)
        self.b_out = np.asarray(params["out/b"], dtype=dtype)
        self.reset_state()

    def reset_state(self):
        self.h = [None] * self.layers
        self.c = [None] * self.layers

    def __call__(self, x):
        x = np.asarray(x, dtype=self.dtype)
        for l in range(self.layers):
            g = _linear(x, self.w_up[l], self.b_up[l])
            if self.h[l] is not None:
                g += self.h[l].dot(self.w_lat[l].T)
            if self.c[l] is None:
                self.c[l] = np.zeros((x.shape[0], g.shape[1] // 4), dtype=self.dtype)
            r = g.reshape(g.shape[0], g.shape[1] // 4, 4)
            a = np.tanh(r[:, :, 0])
            i = _sigmoid(r[:, :, 1])
            f = _sigmoid(r[:, :, 2])
            o = _sigmoid(r[:, :, 3])
            self.c[l] = a * i + f * self.c[l]
            self.h[l] = o * np.tanh(self.c[l])
            x = self.h[l]
        return _linear(x, self.w_out, self.b_out)


class TDNN:
    """``TDNN`` (chainer_networks.py:24-42): ``x.reshape(B, -1, 1, W)`` with ``W = sum(ksize) - len(ksize) + 1``, then
    ``L.Convolution2D(None, units, ksize=(1, k), stride=1)`` + activation per layer (cross-correlation, no padding), then
    ``out`` on the flattened map.  The reshape makes channel c, position p the element ``c*W + p`` of the spliced row.
    Chainer semantics, parity unpinned."""

    recurrent = False

    def __init__(self, params, layers, activation="relu", dtype=np.float32):
        self.dtype = dtype
        self.w = [np.asarray(params["layer_%d/W" % l], dtype=dtype) for l in range(layers)]
        self.b = [np.asarray(params["layer_%d/b" % l], dtype=dtype) for l in range(layers)]
        self.w_out = np.asarray(params["out/W"], dtype=dtype)
        self.b_out = np.asarray(params["out/b"], dtype=dtype)
        self.act = _activation(activation)
        self.input_win_size = sum(w.shape[3] for w in self.w) - layers + 1

    def reset_state(self):
        pass

    def __call__(self, x):
        x = np.asarray(x, dtype=self.dtype)
        h = x.reshape(x.shape[0], -1, self.input_win_size)                       # [B, C, W] (the height-1 axis dropped)
        for w, b in zip(self.w, self.b):
            k = w.shape[3]
            win = np.lib.stride_tricks.sliding_window_view(h, k, axis=2)          # [B, C, P, k]
            h = self.act(np.einsum("bcpj,ucj->bup", win, w[:, :, 0, :], optimize=True).astype(self.dtype) + b[None, :, None])
        return _linear(h.reshape(h.shape[0], -1), self.w_out, self.b_out)


class MGRUNet:
    """Stack of stateful MGRU links + ``out`` (``NetMGRU``, chainer_networks.py:143-161; cell: MGRU.py:35-85), and --
    with ``use_reset_gate=True, activation="tanh"`` -- chainer's ``L.GRU`` stack (``GRU``, chainer_networks.py:123-141):
    MGRU.py is chainer's StatefulGRU with the reset gate made optional and the candidate activation a parameter.

    ``z = W_z x``; ``h~ = W x``; with a previous state: ``z += U_z h``, ``h~ += U (r*h)`` where
    ``r = sig(W_r x + U_r h)`` (or ``h~ += U h`` without reset gate); ``z = sig(z)``; ``h~ = act(h~)``;
    ``h' = z*h~ + (1-z)*h`` (``linear_interpolate``), or ``z*h~`` on the first step.  Every Linear has a bias; the
    state Linears -- biases included -- are skipped on the first step (MGRU.py:70-76).  Parity unpinned (Chainer absent)."""

    recurrent = True
    GATES = ("W_r", "U_r", "W_z", "U_z", "W", "U")

    def __init__(self, params, layers, use_reset_gate=True, activation="tanh", dtype=np.float32):
        self.layers, self.dtype, self.use_reset_gate = layers, dtype, use_reset_gate
        self.act = {"relu": lambda v: np.maximum(v, 0), "tanh": np.tanh, "sigmoid": _sigmoid}[activation]
        names = self.GATES if use_reset_gate else self.GATES[2:]
        self.lin = [{n: (np.asarray(params["layer_%d/%s/W" % (l, n)], dtype=dtype), np.asarray(params["layer_%d/%s/b" % (l, n)], dtype=dtype))
                     for n in names} for l in range(layers)]
        self.w_out = np.asarray(params["out/W"], dtype=dtype)
        self.b_out = np.asarray(params["out/b"], dtype=dtype)
        self.reset_state()

    def reset_state(self):
        self.h = [None] * self.layers

    def __call__(self, x):
        x = np.asarray(x, dtype=self.dtype)
        one = self.dtype(1)
        for l in range(self.layers):
            L, h = self.lin[l], self.h[l]
            z = _linear(x, *L["W_z"])
            h_bar = _linear(x, *L["W"])
            if h is not None:
                z += _linear(h, *L["U_z"])
                if self.use_reset_gate:
                    r = _sigmoid(_linear(x, *L["W_r"]) + _linear(h, *L["U_r"]))
                    h_bar += _linear(r * h, *L["U"])
                else:
                    h_bar += _linear(h, *L["U"])
            z = _sigmoid(z)
            h_bar = self.act(h_bar)
            self.h[l] = z * h_bar + (one - z) * h if h is not None else z * h_bar
            x = self.h[l]
        return _linear(x, self.w_out, self.b_out)


class PeepholeLSTM:
    """Stack of chainer ``L.StatefulPeepholeLSTM`` links + ``out`` (chainer_networks.py:103-121).  Chainer 3.5 semantics
    (source not in the container, parity unpinned): ``g = upward(x) [+ lateral(h)]``, gates interleaved a,i,f,o as in
    ``L.LSTM``; ``i = sig(i + peep_i(c))``, ``f = sig(f + peep_f(c))``, ``c' = tanh(a)*i + f*c``,
    ``o = sig(o + peep_o(c'))``, ``h' = o*tanh(c')``; the three peepholes are bias-free U x U Linears; c starts at 0."""

    recurrent = True

    def __init__(self, params, layers, dtype=np.float32):
        self.layers, self.dtype = layers, dtype
        g = lambda l, k: np.asarray(params["layer_%d/%s" % (l, k)], dtype=dtype)  # noqa: E731
        self.p = [{k: g(l, k) for k in ("upward/W", "upward/b", "lateral/W", "peep_i/W", "peep_f/W", "peep_o/W")} for l in range(layers)]
        self.w_out = np.asarray(params["out/W"], dtype=dtype)
        self.b_out = np.asarray(params["out/b"], dtype=dtype)
        self.reset_state()

    def reset_state(self):
        self.h = [None] * self.layers
        self.c = [None] * self.layers

    def __call__(self, x):
        x = np.asarray(x, dtype=self.dtype)
        for l in range(self.layers):
            P = self.p[l]
            g = _linear(x, P["upward/W"], P["upward/b"])
            if self.h[l] is not None:
                g += self.h[l].dot(P["lateral/W"].T)
            if self.c[l] is None:
                self.c[l] = np.zeros((x.shape[0], g.shape[1] // 4), dtype=self.dtype)
            r = g.reshape(g.shape[0], g.shape[1] // 4, 4)
            c = self.c[l]
            a = np.tanh(r[:, :, 0])
            i = _sigmoid(r[:, :, 1] + c.dot(P["peep_i/W"].T))
            f = _sigmoid(r[:, :, 2] + c.dot(P["peep_f/W"].T))
            c_next = a * i + f * c
            o = _sigmoid(r[:, :, 3] + c_next.dot(P["peep_o/W"].T))
            self.c[l] = c_next
            self.h[l] = o * np.tanh(c_next)
            x = self.h[l]
        return _linear(x, self.w_out, self.b_out)


class RPL4:
    """``RPL4.__call__`` (RPL.py:58-74): log_softmax -> x + x*W + b -> log-add-exp with floor ``lb``."""

    def __init__(self, params, dtype=np.float32):
        self.dtype = dtype
        self.w = np.asarray(params["W"], dtype=dtype).reshape(1, -1)
        self.b = np.asarray(params["b"], dtype=dtype).reshape(1, -1)
        self.lb = np.asarray(params["lb"], dtype=dtype).reshape(1, -1)

    def __call__(self, h):
        h = np.asarray(h, dtype=self.dtype)
        m = np.max(h, axis=1, keepdims=True)
        x = h - (m + np.log(np.sum(np.exp(h - m), axis=1, keepdims=True)))  # F.log_softmax
        h2 = x + x * self.w + self.b
        mx = np.maximum(h2, self.lb)
        mn = np.minimum(h2, self.lb)
        return mx + np.log(1.0 + np.exp(mn - mx)).astype(self.dtype)


class NNWithRPL:
    """Logit-level ensemble wrapper ``NNWithRPL`` (evaluate.py:19-51).

    master only -> master(x); master + K folds -> (K*master + sum folds) / (2K);
    folds only -> mean of folds; then the optional RPL layer.
    """

    def __init__(self, master=None, folds=(), rpl=None):
        self.master = master
        self.folds = list(folds)
        self.rpl = rpl
        members = ([master] if master is not None else []) + self.folds
        self.recurrent = bool(members and members[0].recurrent)

    def reset_state(self):
        if self.master is not None:
            self.master.reset_state()
        for f in self.folds:
            f.reset_state()

    def __call__(self, x):
        k = len(self.folds)
        if self.master is not None and k == 0:
            h = self.master(x)
        elif self.master is not None:
            h = self.master(x) * self.master.dtype(k)
            for f in self.folds:
                h += f(x)
            h /= 2 * k
        else:
            h = 0
            for f in self.folds:
                h = h + f(x)
            h /= k
        if self.rpl is not None:
            h = self.rpl(h)
        return h


def get_nn(network, layers, params, activation="relu", dtype=np.float32):
    """``get_nn`` restricted to the in-scope networks (chainer_networks.py:163-184)."""
    if network == "ff":
        return MLP(params, layers, activation, dtype)
    if network == "tdnn":
        return TDNN(params, layers, activation, dtype)
    if network in ("lstm", "zoneoutlstm", "zoneoutdropoutlstm"):   # zoneout is the identity at inference
        return LSTM(params, layers, dtype)
    if network == "gru":
        return MGRUNet(params, layers, True, "tanh", dtype)
    if network == "mgrurelu":
        return MGRUNet(params, layers, False, "relu", dtype)
    if network == "mgrurelur":
        return MGRUNet(params, layers, True, "relu", dtype)
    if network == "peepholelstm":
        return PeepholeLSTM(params, layers, dtype)
    raise ValueError("network %r is outside the hot-path scope" % network)


def is_nn_recurrent(n):
    """chainer_networks.py:186-187."""
    return n.endswith("lstm") or n.startswith("gru") or n.startswith("mgru")


# --------------------------------------------------------------------------
# The two drivers (predict_folds.py:27-95, evaluateModelForTest.py:36-122)
# --------------------------------------------------------------------------
def predict(model, x, offsets, recurrent, winlen, timedelay, ft, batch_size=1024):
    """Surface B.  Restates ``predict`` (predict_folds.py:27-95); returns float32 ``[N, C]``.

    FF: batches of 1024 rows: prepareBatch -> ft -> model -> ``y - logsum``.
    RNN: utterances sorted by length descending, each edge-padded by ``timedelay`` rows before the
    transform, zeros beyond; time loop over ``Tmax + timedelay``; outputs at ``t >= timedelay`` are
    stored at ``t - timedelay``; un-sorted and concatenated in file order.
    """
    if recurrent:
        offsets = np.asarray(offsets)
        utt_len = offsets[1:] - offsets[:-1]
        utt_idx = np.flip(utt_len.argsort(), axis=0)
        utt_idx_rev = np.zeros(len(utt_len), dtype=np.int64)
        utt_idx_rev[utt_idx] = np.arange(len(utt_len))
        xb = np.zeros((len(utt_len), utt_len[utt_idx[0]] + timedelay, x.shape[1]), dtype=np.float32)
        for i, idx in enumerate(utt_idx):
            x_ = x[offsets[idx]:offsets[idx + 1], :]
            x_ = np.pad(x_, ((0, timedelay), (0, 0)), mode="edge")
            xb[i, :x_.shape[0], :] = apply_feature_transform(x_, ft) if ft is not None else x_
        yb = None
        model.reset_state()
        for t in range(xb.shape[1]):
            n_active = int(np.sum(utt_len > t))
            y = np.asarray(model(xb[:, t, :]), dtype=np.float32)
            y = y - logsum(y, axis=1)
            if yb is None:
                yb = np.zeros((xb.shape[0], xb.shape[1] - timedelay, y.shape[1]), dtype=np.float32)
            if t >= timedelay:
                yb[:n_active, t - timedelay, :] = y[:n_active]
        out = [yb[idx, :utt_len[i], :].reshape((utt_len[i], -1)) for i, idx in enumerate(utt_idx_rev)]
        return np.concatenate(out, axis=0)
    out = []
    offset = 0
    while offset < x.shape[0]:
        offset_end = min(offset + batch_size, x.shape[0])
        x_ = prepare_batch(x, np.arange(offset, offset_end), winlen)
        if ft is not None:
            x_ = apply_feature_transform(x_, ft)
        y = np.asarray(model(x_), dtype=np.float32)
        out.append(y - logsum(y, axis=1))
        offset += batch_size
    return np.concatenate(out, axis=0)


def evaluate_frontend(data, splice_ctx, ft, ivectors=None):
    """evaluate.py:163-171: splice over the whole file -> ft -> THEN concatenate i-vectors."""
    if splice_ctx > 0:
        data = splice(data, range(-splice_ctx, splice_ctx + 1))
    if ft is not None:
        data = apply_feature_transform(data, ft)
    if ivectors is not None:
        data = np.concatenate((data, ivectors), axis=1)
    return data


def evaluate_posteriors(model, data, offsets, ap=None, rnn=False):
    """Surface A up to the file writer.  Restates evaluateModelForTest.py:52-116.

    ``data`` is the already spliced/transformed matrix.  Returns a list of per-utterance float32
    ``[len_u, C]`` arrays in utterance-list order: ``y - ap`` then ``y - logsum(y)``.
    FF: one forward per utterance.  RNN: sort by length descending, zero-pad to Tmax, step the
    whole batch, keep ``yb[:n_active, t]`` (no time-delay shift).
    """
    offsets = np.asarray(offsets)
    n_utts = len(offsets) - 1
    if rnn:
        lens = np.diff(offsets)
        idx_sorted = np.flip(lens.argsort(), axis=0)
        idx_rev = np.zeros(n_utts, dtype=np.int64)
        idx_rev[idx_sorted] = np.arange(n_utts)
        xb = np.zeros((n_utts, lens[idx_sorted[0]], data.shape[1]), dtype=np.float32)
        for i, idx in enumerate(idx_sorted):
            xb[i, :lens[idx], :] = data[offsets[idx]:offsets[idx + 1], :]
        yb = None
        model.reset_state()
        for t in range(xb.shape[1]):
            n_active = int(np.sum(lens > t))
            y = np.asarray(model(xb[:, t, :]), dtype=np.float32)
            if ap is not None:
                y = y - ap
            y = y - logsum(y, axis=1)
            if yb is None:
                yb = np.zeros((xb.shape[0], xb.shape[1], y.shape[1]), dtype=np.float32)
            yb[:n_active, t, :] = y[:n_active]
        return [yb[idx_rev[i], :lens[i], :] for i in range(n_utts)]
    out = []
    for i in range(n_utts):
        y = np.asarray(model(data[offsets[i]:offsets[i + 1], :]), dtype=np.float32)
        if ap is not None:
            y = y - ap
        out.append(y - logsum(y, axis=1))
    return out


def fold_average_dev(models, x, offsets, recurrent, winlen, timedelay, ft):
    """Dev-average mode (predict_folds.py:199-220): mean of per-fold log-posteriors, renormalised."""
    y_out = 0
    for m in models:
        y_out = y_out + predict(m, x, offsets, recurrent, winlen, timedelay, ft)
    y_out /= len(models)
    return y_out - logsum(y_out, axis=1)


# --------------------------------------------------------------------------
# Chainer npz model files (train.py:393-395)
# --------------------------------------------------------------------------
def save_chainer_npz(path, params, prefix="predictor/"):
    """Write ``params`` the way ``serializers.save_npz(path, L.Classifier(model))`` does.

    Keys are ``predictor/<link path>``; written with ``savez_compressed`` to the exact filename
    (no ``.npz`` suffix is appended, cf. ``{out}/model`` at train.py:393).
    """
    with open(path, "wb") as fid:
        np.savez_compressed(fid, **{prefix + k: v for k, v in params.items()})


def load_chainer_npz(path, prefix="predictor/"):
    with np.load(path, allow_pickle=False) as z:
        return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


# --------------------------------------------------------------------------
# Synthetic models / inputs -- shared by tests and bench
# --------------------------------------------------------------------------
def make_mlp_params(rng, in_dim, units, layers, n_out, bias_scale=0.1):
    p = {}
    d = in_dim
    for l in range(layers):
        p["layer_%d/W" % l] = (rng.standard_normal((units, d)) / np.sqrt(d)).astype(np.float32)
        p["layer_%d/b" % l] = (bias_scale * rng.standard_normal(units)).astype(np.float32)
        d = units
    p["out/W"] = (rng.standard_normal((n_out, d)) / np.sqrt(d)).astype(np.float32)
    p["out/b"] = (bias_scale * rng.standard_normal(n_out)).astype(np.float32)
    return p


def make_lstm_params(rng, in_dim, units, layers, n_out, bias_scale=0.1):
    p = {}
    d = in_dim
    for l in range(layers):
        p["layer_%d/upward/W" % l] = (rng.standard_normal((4 * units, d)) / np.sqrt(d)).astype(np.float32)
        b = (bias_scale * rng.standard_normal(4 * units)).astype(np.float32)
        b.reshape(units, 4)[:, 2] += 1.0  # Chainer's default forget-gate bias, interleaved column 2
        p["layer_%d/upward/b" % l] = b
        p["layer_%d/lateral/W" % l] = (rng.standard_normal((4 * units, units)) / np.sqrt(units)).astype(np.float32)
        d = units
    p["out/W"] = (rng.standard_normal((n_out, d)) / np.sqrt(d)).astype(np.float32)
    p["out/b"] = (bias_scale * rng.standard_normal(n_out)).astype(np.float32)
    return p


def make_tdnn_params(rng, in_channels, units, ksize, n_out, bias_scale=0.1):
    """Synthetic parameters of a TDNN in Chainer's layout: layer_{l}/W [units_l, in_l, 1, k_l], layer_{l}/b."""
    p = {}
    c = in_channels
    for l, (u, k) in enumerate(zip(units, ksize)):
        p["layer_%d/W" % l] = (rng.standard_normal((u, c, 1, k)) / np.sqrt(c * k)).astype(np.float32)
        p["layer_%d/b" % l] = (bias_scale * rng.standard_normal(u)).astype(np.float32)
        c = u
    p["out/W"] = (rng.standard_normal((n_out, c)) / np.sqrt(c)).astype(np.float32)
    p["out/b"] = (bias_scale * rng.standard_normal(n_out)).astype(np.float32)
    return p


def make_gru_params(rng, in_dim, units, layers, n_out, reset_gate=True, bias_scale=0.1):
    """Synthetic parameters of a GRU / MGRU stack in Chainer's key layout (layer_{l}/{W_r,U_r,W_z,U_z,W,U}/{W,b})."""
    p = {}
    d = in_dim
    for l in range(layers):
        for name in (("W_r", "U_r") if reset_gate else ()) + ("W_z", "U_z", "W", "U"):
            k = d if name.startswith("W") else units
            p["layer_%d/%s/W" % (l, name)] = (rng.standard_normal((units, k)) / np.sqrt(k)).astype(np.float32)
            p["layer_%d/%s/b" % (l, name)] = (bias_scale * rng.standard_normal(units)).astype(np.float32)
        d = units
    p["out/W"] = (rng.standard_normal((n_out, d)) / np.sqrt(d)).astype(np.float32)
    p["out/b"] = (bias_scale * rng.standard_normal(n_out)).astype(np.float32)
    return p


def make_peephole_params(rng, in_dim, units, layers, n_out, bias_scale=0.1):
    """Synthetic parameters of a StatefulPeepholeLSTM stack (upward, lateral, peep_i, peep_f, peep_o)."""
    p = make_lstm_params(rng, in_dim, units, layers, n_out, bias_scale)
    for l in range(layers):
        for name in ("peep_i", "peep_f", "peep_o"):
            p["layer_%d/%s/W" % (l, name)] = (0.5 * rng.standard_normal((units, units)) / np.sqrt(units)).astype(np.float32)
    return p


def make_offsets(rng, n_utts, n_frames, lo=90, hi=780):
    """Utterance lengths ~ lognormal(ln 290, 0.32) clipped to [lo, hi], rescaled to sum to n_frames."""
    lens = np.clip(rng.lognormal(np.log(290.0), 0.32, n_utts), lo, hi)
    lens = np.maximum(1, np.floor(lens * (n_frames / lens.sum()))).astype(np.int64)
    diff = int(n_frames - lens.sum())
    order = rng.permutation(n_utts)
    k = 0
    while diff != 0:
        j = order[k % n_utts]
        step = 1 if diff > 0 else -1
        if lens[j] + step >= 1:
            lens[j] += step
            diff -= step
        k += 1
    offsets = np.zeros(n_utts + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(lens)
    assert offsets[-1] == n_frames
    return offsets

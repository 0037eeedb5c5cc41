"""Object wrappers over the C ABI: Engine (one GPU), Model (weights in HBM), Pipeline (frontend + members + epilogue).

All arithmetic happens in libamfwd.so's HIP kernels; this module only marshals numpy arrays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .networks import RNN_CELLS
from ._lib import (AM_ACT_RELU, AM_ACT_SIGMOID, AM_ACT_TANH, AM_COMBINE_LOGITS, AM_COMBINE_LOGPOST, AM_NET_FF, AM_NET_LSTM,
                   AmError, c_float_p, c_int64_p, c_int_p, check)

ACTIVATIONS = {"relu": AM_ACT_RELU, "sigmoid": AM_ACT_SIGMOID, "tanh": AM_ACT_TANH}


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return None if a is None else a.ctypes.data_as(c_float_p)


def device_count():
    return int(_lib.load().am_device_count())


def device_pci_bus_id(device):
    """PCI bus id of a visible device ("0000:c1:00.0"): identifies the physical card behind a device index."""
    buf = C.create_string_buffer(64)
    check(_lib.load().am_device_pci_bus_id(int(device), buf, 64))
    return buf.value.decode()


class _Pinned:
    """Owner of one am_host_alloc block, exported through the array interface (numpy keeps it alive as .base)."""

    def __init__(self, shape, dtype):
        self._lib = _lib.load()
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        ptr = C.c_void_p()
        check(self._lib.am_host_alloc(nbytes, C.byref(ptr)))
        self._ptr = ptr
        self.__array_interface__ = {"data": (ptr.value, False), "shape": tuple(int(s) for s in shape), "typestr": dtype.str, "version": 3}

    def __del__(self):
        try:
            if self._ptr:
                self._lib.am_host_free(self._ptr)
                self._ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    return np.asarray(_Pinned(shape, dtype))


class Engine:
    """One MI355X: compute stream, events, workspace.  Raises AmError when the device is not a gfx950 GPU."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        check(self._lib.am_engine_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.am_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        check(self._lib.am_engine_device_name(self._h, buf, 256))
        return buf.value.decode()

    def synchronize(self):
        check(self._lib.am_engine_synchronize(self._h))

    def record(self, slot):
        check(self._lib.am_event_record(self._h, int(slot)))

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        check(self._lib.am_event_elapsed_ms(self._h, int(a), int(b), C.byref(ms)))
        return float(ms.value)

    def decode_elapsed_ms(self):
        """Device time of the launch sequence of this engine's last decode (PhoneRecog.recognise / recognise_pipeline)."""
        ms = C.c_float()
        check(self._lib.am_decode_elapsed_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def set_precision(self, mode):
        """"f32" (exact f32 MFMA, default) or "split_f16" (f32-accurate products on the f16 pipe, see include/amfwd.h)."""
        modes = {"f32": _lib.AM_PRECISION_F32, "exact": _lib.AM_PRECISION_F32, "split_f16": _lib.AM_PRECISION_SPLIT_F16}
        check(self._lib.am_engine_set_precision(self._h, modes[mode]))

    def profile(self, on=True):
        check(self._lib.am_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        check(self._lib.am_profile_reset(self._h))

    def profile_read(self):
        """-> {kernel name: (launches, total ms)} since the last reset."""
        calls = (C.c_int64 * _lib.AM_KERNEL_ID_MAX)()
        ms = (C.c_double * _lib.AM_KERNEL_ID_MAX)()
        check(self._lib.am_profile_read(self._h, calls, ms))
        out = {}
        for i in range(_lib.AM_KERNEL_ID_MAX):
            nm = self._lib.am_kernel_name(i)
            if nm is not None and calls[i]:
                out[nm.decode()] = (int(calls[i]), float(ms[i]))
        return out

    def pinned_empty(self, shape, dtype=np.float32):
        """numpy array on page-locked host memory (am_host_alloc): downloads into it run at PCIe line rate."""
        return pinned_empty(shape, dtype)

    # stand-alone stage entry points --------------------------------------------------------------
    def lognormalise(self, lp, ap=None):
        lp = np.array(lp, dtype=np.float32, order="C", copy=True)
        ap_ = None if ap is None else _f32(np.asarray(ap).reshape(-1))
        check(self._lib.am_lognormalise(self._h, _fp(lp), lp.shape[0], lp.shape[1], _fp(ap_)))
        return lp

    def linear(self, x, w, b=None, act=None):
        x, w = _f32(x), _f32(w)
        b_ = None if b is None else _f32(b)
        y = np.empty((x.shape[0], w.shape[0]), dtype=np.float32)
        a = -1 if act is None else ACTIVATIONS[act]
        check(self._lib.am_linear(self._h, _fp(x), x.shape[0], x.shape[1], _fp(w), _fp(b_), w.shape[0], a, _fp(y)))
        return y


class Model:
    """A network's weights resident on the engine's GPU.

    ``params``: Chainer npz keys without the ``predictor/`` prefix:
    ff   -> layer_{l}/W [units,in], layer_{l}/b, out/W [C,units], out/b
    lstm -> layer_{l}/upward/W [4U,in], layer_{l}/upward/b [4U], layer_{l}/lateral/W [4U,U], out/W, out/b
    tdnn -> layer_{l}/W [units_l, in_l, 1, k_l], layer_{l}/b, out/W, out/b
    gru / mgrurelu / mgrurelur / peepholelstm -> the keys listed in ``networks.RNN_CELLS`` + out/W, out/b
    Layer input sizes are taken from the stored shapes, as ``L.Linear(None, ...)`` does.
    """

    def __init__(self, engine, network, layers, params, activation="relu"):
        self.engine = engine
        self.network = network
        self.layers = int(layers)
        self._lib = engine._lib
        if activation not in ACTIVATIONS:
            raise ValueError("Wrong activation function specified")
        ws, bs, lats = [], [], []
        if network == "ff":
            kind = AM_NET_FF
            for l in range(self.layers):
                ws.append(_f32(params["layer_%d/W" % l]))
                bs.append(_f32(params["layer_%d/b" % l]))
        elif network == "lstm":
            kind = AM_NET_LSTM
            for l in range(self.layers):
                ws.append(_f32(params["layer_%d/upward/W" % l]))
                bs.append(_f32(params["layer_%d/upward/b" % l]))
                lats.append(_f32(params["layer_%d/lateral/W" % l]))
            _check_gate_layout(bs)
        elif network in RNN_CELLS:
            self._create_rnn(engine, network, params)
            return
        elif network == "tdnn":
            self._create_tdnn(engine, params, ACTIVATIONS[activation])
            return
        else:
            raise ValueError("network %r is not supported by the MI355X engine (supported: ff, tdnn, lstm, %s)" % (network, ", ".join(RNN_CELLS)))
        ws.append(_f32(params["out/W"]))
        bs.append(_f32(params["out/b"]))
        n = self.layers + 1
        in_dims = (C.c_int * n)(*[w.shape[1] for w in ws])
        out_dims = (C.c_int * n)(*[w.shape[0] for w in ws])
        w_arr = (c_float_p * n)(*[_fp(w) for w in ws])
        b_arr = (c_float_p * n)(*[_fp(b) for b in bs])
        lat_arr = (c_float_p * max(len(lats), 1))(*[_fp(w) for w in lats]) if lats else None
        h = C.c_void_p()
        check(self._lib.am_model_create(engine._h, kind, self.layers, ACTIVATIONS[activation], in_dims, out_dims,
                                        w_arr, b_arr, lat_arr, C.byref(h)))
        self._h = h
        self.in_dim = int(ws[0].shape[1])
        self.num_classes = int(ws[-1].shape[0])
        self.recurrent = network == "lstm"

    def _create_tdnn(self, engine, params, act):
        """tdnn: layer_{l}/W [units_l, in_channels_l, 1, k_l] (L.Convolution2D), layer_{l}/b, out/W, out/b (am_model_create_tdnn)."""
        ws = [_f32(params["layer_%d/W" % l]) for l in range(self.layers)]
        bs = [_f32(params["layer_%d/b" % l]) for l in range(self.layers)]
        out_w, out_b = _f32(params["out/W"]), _f32(params["out/b"])
        if self.layers < 1:
            raise ValueError("a TDNN needs at least one convolution layer")
        cin = ws[0].shape[1] if ws[0].ndim == 4 else -1
        for l, (w, b) in enumerate(zip(ws, bs)):   # the C side trusts these shapes
            if w.ndim != 4 or w.shape[1] != cin or w.shape[2] != 1 or b.shape != (w.shape[0],):
                raise ValueError("layer_%d/W has shape %s / b %s: not a (1,k) convolution over %d channels" % (l, w.shape, b.shape, cin))
            cin = w.shape[0]
        if out_w.shape[1:] != (cin,) or out_b.shape != (out_w.shape[0],):
            raise ValueError("out/W has shape %s, expected (C, %d): the last convolution must leave one position" % (out_w.shape, cin))
        n = self.layers
        units = (C.c_int * n)(*[w.shape[0] for w in ws])
        ksize = (C.c_int * n)(*[w.shape[3] for w in ws])
        w_arr = (c_float_p * n)(*[_fp(w) for w in ws])
        b_arr = (c_float_p * n)(*[_fp(b) for b in bs])
        h = C.c_void_p()
        check(self._lib.am_model_create_tdnn(engine._h, n, act, int(ws[0].shape[1]), units, ksize, w_arr, b_arr, _fp(out_w), _fp(out_b),
                                             int(out_w.shape[0]), C.byref(h)))
        self._h = h
        self.win = sum(int(w.shape[3]) for w in ws) - n + 1
        self.in_dim = int(ws[0].shape[1]) * self.win
        self.num_classes = int(out_w.shape[0])
        self.recurrent = False

    def _create_rnn(self, engine, network, params):
        """gru / mgrurelu / mgrurelur / peepholelstm (am_model_create_rnn; parameter order: networks.RNN_CELLS)."""
        cell, act, keys = RNN_CELLS[network]
        arrs = [_f32(params["layer_%d/%s" % (l, k)]) for l in range(self.layers) for k in keys]
        out_w, out_b = _f32(params["out/W"]), _f32(params["out/b"])
        first = arrs[keys.index("W_z/W") if "W_z/W" in keys else 0]
        units = int(out_w.shape[1])
        in_dim = int(first.shape[1]) if first.ndim == 2 else -1
        for l in range(self.layers):   # the C side trusts these shapes
            d = in_dim if l == 0 else units
            for k, a in zip(keys, arrs[l * len(keys):(l + 1) * len(keys)]):
                name = k.split("/")[0]
                rows = 4 * units if name in ("upward", "lateral") else units
                want = (rows,) if k.endswith("/b") else (rows, d if name in ("upward", "W_r", "W_z", "W") else units)
                if a.shape != want:
                    raise ValueError("layer_%d/%s has shape %s, expected %s" % (l, k, a.shape, want))
        if out_b.shape != (out_w.shape[0],):
            raise ValueError("out/b has shape %s, expected (%d,)" % (out_b.shape, out_w.shape[0]))
        p_arr = (c_float_p * len(arrs))(*[_fp(a) for a in arrs])
        h = C.c_void_p()
        check(self._lib.am_model_create_rnn(engine._h, cell, self.layers, ACTIVATIONS[act], int(first.shape[1]), units,
                                            int(out_w.shape[0]), p_arr, _fp(out_w), _fp(out_b), C.byref(h)))
        self._h = h
        self.in_dim = int(first.shape[1])
        self.num_classes = int(out_w.shape[0])
        self.recurrent = True

    def close(self):
        if getattr(self, "_h", None):
            self._lib.am_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _check_gate_layout(upward_biases):
    """Loader self-check from SURVEY.md Appendix A.3: Chainer interleaves gates (column 4u+k) and initialises the forget
    bias (k = 2) to 1; warn loudly when a contiguous-block reading of the bias looks more plausible."""
    import warnings
    for l, b in enumerate(upward_biases):
        u = b.shape[0] // 4
        if u < 4:
            continue
        inter = b.reshape(u, 4).mean(axis=0)
        block = b.reshape(4, u).mean(axis=1)
        if np.argmax(block) == 2 and block[2] - np.delete(block, 2).max() > 0.5 and np.argmax(inter) != 2:
            warnings.warn("LSTM layer %d: upward/b looks gate-blocked, not gate-interleaved as Chainer stores it; "
                          "posteriors will be wrong if the file was not written by Chainer" % l)


class Pipeline:
    """Frontend + member networks + epilogue, plus the resident I/O buffers of one shard."""

    def __init__(self, engine):
        self.engine = engine
        self._lib = engine._lib
        h = C.c_void_p()
        check(self._lib.am_pipeline_create(engine._h, C.byref(h)))
        self._h = h
        self._members = []
        self.num_classes = None
        self.d_iv = 0
        self.D = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.am_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # configuration ---------------------------------------------------------------------------------
    def set_frontend(self, D, shifts, add_shift=None, rescale=None, d_iv=0):
        shifts = [int(s) for s in shifts]
        arr = (C.c_int * len(shifts))(*shifts)
        a = None if add_shift is None else _f32(add_shift)
        r = None if rescale is None else _f32(rescale)
        if a is not None and (a.size != len(shifts) * D or r is None or r.size != a.size):
            raise ValueError("operands could not be broadcast together: transform has %d entries, spliced frame has %d"
                             % (a.size, len(shifts) * D))
        check(self._lib.am_pipeline_set_frontend(self._h, int(D), len(shifts), arr, _fp(a), _fp(r), int(d_iv)))
        self.D, self.shifts, self.d_iv = int(D), shifts, int(d_iv)
        self.max_shift = max(abs(s) for s in shifts)

    def set_members(self, members, has_master=False, combine=AM_COMBINE_LOGITS):
        members = list(members)
        arr = (C.c_void_p * len(members))(*[m._h for m in members])
        check(self._lib.am_pipeline_set_members(self._h, len(members), arr, 1 if has_master else 0, int(combine)))
        self._members = members
        self.num_classes = members[0].num_classes
        self.recurrent = members[0].recurrent

    def set_rpl4(self, w, b, lb):
        if w is None:
            check(self._lib.am_pipeline_set_rpl4(self._h, None, None, None))
            return
        w, b, lb = (_f32(np.asarray(v).reshape(-1)) for v in (w, b, lb))
        check(self._lib.am_pipeline_set_rpl4(self._h, _fp(w), _fp(b), _fp(lb)))

    def set_prior(self, ap):
        ap_ = None if ap is None else _f32(np.asarray(ap).reshape(-1))
        check(self._lib.am_pipeline_set_prior(self._h, _fp(ap_)))

    def set_normalise(self, on):
        check(self._lib.am_pipeline_set_normalise(self._h, 1 if on else 0))

    def synchronize(self):
        """Wait for the enqueued passes and make sure the resident posteriors are valid (am_pipeline_synchronize)."""
        check(self._lib.am_pipeline_synchronize(self._h))

    # feed-forward ----------------------------------------------------------------------------------
    def ff_upload(self, x_ext, ivec, n, halo_lo, halo_hi):
        x_ext = _f32(x_ext)
        iv = None if ivec is None else _f32(ivec)
        if x_ext.shape != (halo_lo + n + halo_hi, self.D):
            raise ValueError("x must be [halo_lo + n + halo_hi, D]")
        check(self._lib.am_ff_upload(self._h, _fp(x_ext), _fp(iv), int(n), int(halo_lo), int(halo_hi)))
        self._n = int(n)

    def ff_run(self):
        check(self._lib.am_ff_run(self._h))

    def ff_download(self, out=None):
        y = out if out is not None else np.empty((self._n, self.num_classes), dtype=np.float32)
        assert y.dtype == np.float32 and y.flags.c_contiguous and y.shape == (self._n, self.num_classes)
        check(self._lib.am_ff_download(self._h, _fp(y)))
        return y

    def forward_ff(self, x_ext, ivec=None, n=None, halo_lo=0, halo_hi=0, out=None):
        """Host array in, host array out (am_forward_ff): the download of chunk c overlaps the compute of chunk c+1."""
        x_ext = _f32(x_ext)
        iv = None if ivec is None else _f32(ivec)
        n = x_ext.shape[0] - halo_lo - halo_hi if n is None else n
        if x_ext.shape != (halo_lo + n + halo_hi, self.D):
            raise ValueError("x must be [halo_lo + n + halo_hi, D]")
        y = out if out is not None else np.empty((n, self.num_classes), dtype=np.float32)
        assert y.dtype == np.float32 and y.flags.c_contiguous and y.shape == (n, self.num_classes)
        check(self._lib.am_forward_ff(self._h, _fp(x_ext), _fp(iv), int(n), int(halo_lo), int(halo_hi), _fp(y)))
        self._n = int(n)
        return y

    def overflowed(self):
        """True if the last split-f16 run met an activation that does not fit f16 (its results must be recomputed)."""
        flag = C.c_int(0)
        check(self._lib.am_pipeline_overflowed(self._h, C.byref(flag)))
        return bool(flag.value)

    # recurrent -------------------------------------------------------------------------------------
    def lstm_upload(self, x, ivec, offsets, timedelay=0):
        x = _f32(x)
        iv = None if ivec is None else _f32(ivec)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        if off[-1] != x.shape[0]:
            raise ValueError("offsets[-1] must equal the number of frames")
        check(self._lib.am_lstm_upload(self._h, _fp(x), _fp(iv), off.ctypes.data_as(c_int64_p), len(off) - 1, int(timedelay)))
        self._n = int(x.shape[0])

    def lstm_run(self):
        check(self._lib.am_lstm_run(self._h))

    def lstm_download(self, out=None):
        y = out if out is not None else np.empty((self._n, self.num_classes), dtype=np.float32)
        assert y.dtype == np.float32 and y.flags.c_contiguous and y.shape == (self._n, self.num_classes)
        check(self._lib.am_lstm_download(self._h, _fp(y)))
        return y

    def forward_lstm(self, x, offsets, ivec=None, timedelay=0, out=None):
        self.lstm_upload(x, ivec, offsets, timedelay)
        self.lstm_run()
        return self.lstm_download(out)

    def recurrent_form(self):
        """(members that run as ONE persistent launch per chunk of utterance slots, members) of the current recurrent upload."""
        a, b = C.c_int(0), C.c_int(0)
        check(self._lib.am_pipeline_recurrent_form(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # stage entry point -------------------------------------------------------------------------------
    def frontend_apply(self, x_ext, ivec=None, n=None, halo_lo=0, halo_hi=0):
        x_ext = _f32(x_ext)
        iv = None if ivec is None else _f32(ivec)
        n = x_ext.shape[0] - halo_lo - halo_hi if n is None else n
        out = np.empty((n, len(self.shifts) * self.D + self.d_iv), dtype=np.float32)
        check(self._lib.am_frontend_apply(self._h, _fp(x_ext), _fp(iv), int(n), int(halo_lo), int(halo_hi), _fp(out)))
        return out

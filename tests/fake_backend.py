"""TEST INFRASTRUCTURE: a stand-in for ``nnacousticmodeling_amd.engine`` built on the CPU oracle.

It lets the ``-m "not gpu"`` suite exercise the host logic (sharding, gather, CLIs, file writers) in a container without
a GPU.  It is never imported by the product; ``PosteriorRunner(backend=...)`` is the only injection point.
It implements the same Engine / Model / Pipeline surface with the semantics documented in include/amfwd.h.
"""
import numpy as np

from oracle import am_oracle as orc

AM_COMBINE_LOGITS, AM_COMBINE_LOGPOST = 0, 1
created_engines = []


class Engine:
    def __init__(self, device=0):
        self.device = device
        created_engines.append(device)

    def close(self):
        pass

    def set_precision(self, mode):   # the stand-in computes in numpy f32 either way
        self.precision = mode


class Model:
    def __init__(self, engine, network, layers, params, activation="relu"):
        self.net = orc.get_nn(network, layers, params, activation)
        self.recurrent = orc.is_nn_recurrent(network)
        self.num_classes = params["out/W"].shape[0]

    def close(self):
        pass


class Pipeline:
    def __init__(self, engine):
        self.engine = engine
        self.rpl = None
        self.prior = None
        self.normalise = True

    def close(self):
        pass

    def set_frontend(self, D, shifts, add_shift=None, rescale=None, d_iv=0):
        self.D, self.shifts, self.d_iv = D, list(shifts), d_iv
        if add_shift is not None and len(add_shift) != len(self.shifts) * D:
            raise ValueError("operands could not be broadcast together")
        self.ft = None if add_shift is None else {"addShift": np.asarray(add_shift, np.float32), "rescale": np.asarray(rescale, np.float32)}

    def set_members(self, members, has_master=False, combine=AM_COMBINE_LOGITS):
        self.members, self.has_master, self.combine = list(members), has_master, combine
        self.num_classes = members[0].num_classes

    def set_rpl4(self, w, b, lb):
        self.rpl = None if w is None else orc.RPL4({"W": w, "b": b, "lb": lb})

    def set_prior(self, ap):
        self.prior = None if ap is None else np.asarray(ap, np.float32).reshape(1, -1)

    def set_normalise(self, on):
        self.normalise = bool(on)

    def _ensemble(self):
        nets = [m.net for m in self.members]
        if self.has_master:
            return orc.NNWithRPL(nets[0], nets[1:], self.rpl)
        return orc.NNWithRPL(None, nets, self.rpl)

    def _finish(self, logits_fn, rows):
        if self.combine == AM_COMBINE_LOGPOST:
            y = 0
            for m in self.members:
                yy = logits_fn(m.net)
                y = y + (yy - orc.logsum(yy, axis=1))
            y /= len(self.members)
        else:
            y = logits_fn(self._ensemble())
        y = np.asarray(y, np.float32)
        if self.prior is not None:
            y = y - self.prior
        if self.normalise:
            y = y - orc.logsum(y, axis=1)
        return y

    def forward_ff(self, x_ext, ivec=None, n=None, halo_lo=0, halo_hi=0, out=None):
        n = x_ext.shape[0] - halo_lo - halo_hi if n is None else n
        feats = orc.splice(x_ext, self.shifts)[halo_lo:halo_lo + n]  # clamps at the ends of THIS array (amfwd.h)
        if self.ft is not None:
            feats = orc.apply_feature_transform(feats, self.ft)
        if ivec is not None:
            feats = np.concatenate((feats, ivec), axis=1)
        y = self._finish(lambda net: net(feats), n)
        self._res = y     # (resident, as after am_forward_ff)
        if out is not None:
            out[...] = y
            return out
        return y

    # resident form (upload / run / download), as the product's Pipeline offers it
    def ff_upload(self, x_ext, ivec, n, halo_lo, halo_hi):
        self._job = ("ff", np.array(x_ext, np.float32), None if ivec is None else np.array(ivec, np.float32), n, halo_lo, halo_hi)
        self._res = None

    def ff_run(self):
        _, x_ext, ivec, n, h_lo, h_hi = self._job
        self._res = self.forward_ff(x_ext, ivec, n, h_lo, h_hi)

    def ff_download(self, out=None):
        if out is not None:
            out[...] = self._res
            return out
        return self._res.copy()

    def lstm_upload(self, x, ivec, offsets, timedelay=0):
        self._job = ("lstm", np.array(x, np.float32), None if ivec is None else np.array(ivec, np.float32), np.array(offsets), timedelay)
        self._res = None

    def lstm_run(self):
        _, x, ivec, offsets, timedelay = self._job
        self._res = self.forward_lstm(x, offsets, ivec, timedelay)

    lstm_download = ff_download

    def forward_lstm(self, x, offsets, ivec=None, timedelay=0, out=None):
        feats = x if self.ft is None else orc.apply_feature_transform(x, self.ft)
        if ivec is not None:
            feats = np.concatenate((feats, ivec), axis=1)
        offsets = np.asarray(offsets)

        def logits(net):
            res = np.zeros((x.shape[0], self.num_classes), dtype=np.float32)
            for u in range(len(offsets) - 1):
                net.reset_state()
                for t in range(offsets[u], offsets[u + 1]):
                    res[t] = net(feats[t:t + 1])[0]
            return res

        y = self._finish(logits, x.shape[0])
        if timedelay > 0:  # zero tail quirk of predict_folds.py:50,61 (dev mode renormalises the zero rows: -log C)
            fill = -np.log(np.float32(self.num_classes)) if (self.combine == AM_COMBINE_LOGPOST and self.normalise) else 0.0
            shifted = np.full_like(y, fill)
            for u in range(len(offsets) - 1):
                lo, hi = offsets[u], offsets[u + 1]
                if hi - lo > timedelay:
                    shifted[lo:hi - timedelay] = y[lo + timedelay:hi]
            y = shifted
        if out is not None:
            out[...] = y
            return out
        return y

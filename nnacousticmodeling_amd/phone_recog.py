"""In-process GPU phone recogniser: drop-in for the reference's ``recog/PhoneRecog`` executable.

``PhoneRecog scp img out.mlf PIP LMW`` (reference recog_src/source/main.cpp:18-85) reads the per-utterance ``.lab`` files,
runs a triphone Viterbi (``CPhoneRecog::RecogOne``, phone_recog.cpp:138-254) and writes an HTK MLF (``CMlfWriter``,
mlf_io.cpp:20-41).  Here the decoding-graph image is parsed on the host (format: ``CPhoneRecog::Init``,
phone_recog.cpp:84-136), the Viterbi runs on the GPU with one workgroup per utterance (``am_decode_*``), and the MLF is
written in the same text format -- without the ``.lab`` round trip when the posteriors are already on the device.
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

from ._lib import c_float_p, c_int64_p, c_int_p, check


def read_image(path):
    """``kaldiTri1909.img`` -> dict of numpy arrays + ``phones`` (one output label per triphone)."""
    with open(path, "rb") as f:
        raw = f.read()
    pos = 0

    def take(dt, n):
        nonlocal pos
        nbytes = np.dtype(dt).itemsize * n
        if n < 0 or pos + nbytes > len(raw):
            raise ValueError("Error reading image file %s (truncated at byte %d)" % (path, pos))
        a = np.frombuffer(raw, dtype=dt, count=n, offset=pos).copy()
        pos += nbytes
        return a

    P, B, T, S = (int(v) for v in take("<i4", 4))
    g = {"numPhones": P, "numBiphones": B, "numTriphones": T, "numStates": S}
    g["startIdxs"], g["endIdxs"] = take("<i4", P), take("<i4", P)
    g["biphoneIdxIn"], g["biphoneIdxOut"] = take("<i4", T), take("<i4", T)
    g["stateIdxs"] = take("<i4", 3 * T)
    g["tpSelfloop"], g["tpTransition"] = take("<f4", 3 * T), take("<f4", 3 * T)
    g["outOffsets"] = take("<i4", B + 1)
    g["outValues"] = take("<i4", int(g["outOffsets"][B]))
    g["lm"] = take("<f4", B)
    nchars = int(take("<i4", 1)[0])
    table = raw[pos:pos + nchars]
    if len(table) != nchars:
        raise ValueError("Error reading image file %s (phone table truncated)" % path)
    # Init splits on every ' ' after the first character and keeps T pointers (phone_recog.cpp:123-129)
    g["phones"] = table.decode("latin-1").split(" ")[:T]
    return g


def base_name_copy(path):
    """``baseNameCopy`` (main.cpp:87-97): text after the last '/' or '\\\\' up to (not including) the LAST '.'."""
    b, e = 0, len(path)
    for i, ch in enumerate(path):
        if ch in "/\\":
            b = i + 1
        if ch == ".":
            e = i - 1
    return path[b:e + 1]


def write_mlf(path, utt_names, results):
    """``CMlfWriter`` text format (mlf_io.cpp:20-41): times are frame indices followed by the literal ``00000``."""
    with open(path, "w") as f:
        f.write("#!MLF!#\n")
        for name, segs in zip(utt_names, results):
            f.write('"*/%s.rec"\n' % name)
            for begin, end, phone in segs:
                f.write("%d00000 %d00000 %s\n" % (begin, end, phone))
            f.write(".\n")


class PhoneRecog:
    """GPU decoder bound to one engine.  ``PIP`` is used as given (the reference caller passes ``-abs(PIP)``)."""

    def __init__(self, engine, img_file, PIP, LMW):
        self.engine = engine
        self._lib = engine._lib
        g = self.graph = read_image(img_file) if isinstance(img_file, str) else img_file
        P, B, T, S = (int(g[k]) if k in g else None for k in ("numPhones", "numBiphones", "numTriphones", "numStates"))
        P = len(g["startIdxs"]) if P is None else P
        B = len(g["lm"]) if B is None else B
        T = len(g["biphoneIdxIn"]) if T is None else T
        self.num_states = S
        self.phones = list(g["phones"])
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
        arrs = [i32(g["startIdxs"]), i32(g["endIdxs"]), i32(g["biphoneIdxIn"]), i32(g["stateIdxs"]), f32(g["tpSelfloop"]),
                f32(g["tpTransition"]), i32(g["outOffsets"]), i32(g["outValues"]), f32(g["lm"])]
        ptr = lambda a: a.ctypes.data_as(c_int_p if a.dtype == np.int32 else c_float_p)  # noqa: E731
        h = C.c_void_p()
        check(self._lib.am_decoder_create(engine._h, P, B, T, S, *[ptr(a) for a in arrs], float(PIP), float(LMW), C.byref(h)))
        self._h = h

    FORMS = {"auto": 0, "frame_sync": 1, "per_utt_compact": 2, "per_utt_wide": 3}

    def set_form(self, form):
        """Kernel form (include/amfwd.h AM_VITERBI_*): "auto", "frame_sync", "per_utt_compact", "per_utt_wide"."""
        check(self._lib.am_decoder_set_form(self._h, self.FORMS[form]))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.am_decoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _collect(self, offsets, sb, se, st, sc):
        out = []
        for u in range(len(offsets) - 1):
            o, c = int(offsets[u]), int(sc[u])
            out.append([(int(sb[o + j]), int(se[o + j]), self.phones[int(st[o + j])]) for j in range(c)])
        return out

    def _buffers(self, offsets):
        n = max(int(offsets[-1]), 1)
        mk = lambda k: np.zeros(k, dtype=np.int32)  # noqa: E731
        return mk(n), mk(n), mk(n), mk(max(len(offsets) - 1, 1))

    def recognise(self, lp, offsets):
        """lp: float32 ``[N, numStates]`` (what the ``.lab`` files hold); -> per utterance [(begin, end, phone), ...]."""
        lp = np.ascontiguousarray(lp, dtype=np.float32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if lp.shape[1] != self.num_states:
            raise ValueError("Num states in log-probs file and model image differ!")
        sb, se, st, sc = self._buffers(offsets)
        ip = lambda a: a.ctypes.data_as(c_int_p)  # noqa: E731
        check(self._lib.am_decode_host(self._h, lp.ctypes.data_as(c_float_p), offsets.ctypes.data_as(c_int64_p), len(offsets) - 1,
                                       ip(sb), ip(se), ip(st), ip(sc)))
        return self._collect(offsets, sb, se, st, sc)

    def recognise_pipeline(self, pipeline, offsets):
        """Decode the posteriors ``pipeline`` has just produced, straight from device memory."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        sb, se, st, sc = self._buffers(offsets)
        ip = lambda a: a.ctypes.data_as(c_int_p)  # noqa: E731
        check(self._lib.am_decode_pipeline(self._h, pipeline._h, offsets.ctypes.data_as(c_int64_p), len(offsets) - 1, ip(sb), ip(se), ip(st), ip(sc)))
        return self._collect(offsets, sb, se, st, sc)

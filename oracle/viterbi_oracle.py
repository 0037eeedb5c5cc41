"""CPU oracle for the triphone Viterbi decoder.  TEST INFRASTRUCTURE ONLY.

Restates ``CPhoneRecog::Init`` / ``RecogOne`` (reference recog_src/source/phone_recog.cpp:84-136, 138-254) in numpy,
float32 operation for float32 operation, so results are bit-identical to the reference binary built from source
(``oracle/_ref/libphonerecog_ref.so``, see oracle/Makefile): tests/golden/make_viterbi_golden.py checks that on synthetic
decoding graphs and freezes the vectors (``parity pinned by the reference itself``).

Decoding-graph image (``kaldiTri1909.img`` format, phone_recog.cpp:84-136), all little-endian:
  int32 numPhones, numBiphones, numTriphones, numStates
  int32 startIdxs[numPhones], endIdxs[numPhones]          biphone nodes where a path may start / end
  int32 biphoneIdxIn[T], biphoneIdxOut[T]                 node a triphone leaves from / arrives at
  int32 stateIdxs[3T]                                     emitting-state (posterior column) of state s of triphone n at [3n+s]
  float tpSelfloop[3T], tpTransition[3T]                  log transition probabilities, [3n+s]
  int32 outOffsets[B+1], outValues[outOffsets[B]]         CSR list: triphones arriving at each node
  float lm[B]                                             node (bigram) log-probabilities, scaled by LMW at load
  int32 numChars, char phoneTable[numChars]               T space-separated output labels
"""
from __future__ import annotations

import ctypes as C
import os
import struct

import numpy as np

INF = np.float32(1e20)


def write_image(path, g):
    """g: dict with the fields above (numpy arrays) and ``phones`` (list of T strings)."""
    T, B, P = len(g["biphoneIdxIn"]), len(g["lm"]), len(g["startIdxs"])
    table = " ".join(g["phones"]).encode("ascii")
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", P, B, T, int(g["numStates"])))
        for key, dt in (("startIdxs", np.int32), ("endIdxs", np.int32), ("biphoneIdxIn", np.int32), ("biphoneIdxOut", np.int32),
                        ("stateIdxs", np.int32), ("tpSelfloop", np.float32), ("tpTransition", np.float32),
                        ("outOffsets", np.int32), ("outValues", np.int32), ("lm", np.float32)):
            f.write(np.ascontiguousarray(g[key], dtype=dt).tobytes())
        f.write(struct.pack("<i", len(table)))
        f.write(table)


def read_image(path):
    with open(path, "rb") as f:
        raw = f.read()
    pos = 0

    def take(dt, n):
        nonlocal pos
        a = np.frombuffer(raw, dtype=dt, count=n, offset=pos).copy()
        pos += a.nbytes
        return a

    P, B, T, S = (int(v) for v in take(np.int32, 4))
    g = {"numStates": S}
    g["startIdxs"], g["endIdxs"] = take(np.int32, P), take(np.int32, P)
    g["biphoneIdxIn"], g["biphoneIdxOut"] = take(np.int32, T), take(np.int32, T)
    g["stateIdxs"] = take(np.int32, 3 * T)
    g["tpSelfloop"], g["tpTransition"] = take(np.float32, 3 * T), take(np.float32, 3 * T)
    g["outOffsets"] = take(np.int32, B + 1)
    g["outValues"] = take(np.int32, int(g["outOffsets"][B]))
    g["lm"] = take(np.float32, B)
    nchars = int(take(np.int32, 1)[0])
    g["phones"] = raw[pos:pos + nchars].decode("ascii").split(" ")
    return g


def make_graph(rng, n_phones=5, n_states=60):
    """Synthetic graph with the structure of a triphone loop: nodes = phone pairs (a,b), triphone (a,b,c): (a,b) -> (b,c)."""
    P = n_phones
    B = P * P
    tri = [(a, b, c) for a in range(P) for b in range(P) for c in range(P) if rng.random() < 0.8]
    T = len(tri)
    node = lambda a, b: a * P + b  # noqa: E731
    g = {"numStates": n_states}
    g["startIdxs"] = np.array([node(0, i) for i in range(P)], dtype=np.int32)
    g["endIdxs"] = np.array([node(i, 0) for i in range(P)], dtype=np.int32)
    g["biphoneIdxIn"] = np.array([node(a, b) for a, b, c in tri], dtype=np.int32)
    g["biphoneIdxOut"] = np.array([node(b, c) for a, b, c in tri], dtype=np.int32)
    g["stateIdxs"] = rng.integers(0, n_states, 3 * T).astype(np.int32)
    g["tpSelfloop"] = np.log(rng.uniform(0.3, 0.8, 3 * T)).astype(np.float32)
    g["tpTransition"] = np.log(rng.uniform(0.2, 0.7, 3 * T)).astype(np.float32)
    lists = [[] for _ in range(B)]
    for n, (a, b, c) in enumerate(tri):
        lists[node(b, c)].append(n)
    g["outOffsets"] = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
    g["outValues"] = np.array([n for l in lists for n in l], dtype=np.int32)
    g["lm"] = (-3.0 * rng.random(B)).astype(np.float32)
    g["phones"] = ["p%d" % b for a, b, c in tri]
    return g


def recog_one(g, lp, PIP, LMW):
    """-> list of (begin, end, triphone id); ``phones[id]`` is the label the reference prints.

    float32 arithmetic in the reference's order: every score is a single float add of two floats; ties keep the
    *stay* branch (``pGo > pStay``) and the first maximum (``mx < pOut``)."""
    f32 = np.float32
    lp = np.ascontiguousarray(lp, dtype=f32)
    N = lp.shape[0]
    P, T, B = len(g["startIdxs"]), len(g["biphoneIdxIn"]), len(g["lm"])
    assert lp.shape[1] == g["numStates"], "Num states in log-probs file and model image differ!"
    lm = (g["lm"] * f32(LMW)).astype(f32)                      # phone_recog.cpp:112
    PIP = f32(PIP)
    bin_, st = g["biphoneIdxIn"], g["stateIdxs"].reshape(T, 3)
    tps, tpt = g["tpSelfloop"].reshape(T, 3), g["tpTransition"].reshape(T, 3)
    off, vals = g["outOffsets"], g["outValues"]
    p = np.full(B, -INF, dtype=f32)
    # lm - log((float)numPhones): double arithmetic, rounded once to float (phone_recog.cpp:148)
    p[g["startIdxs"]] = (lm[g["startIdxs"]].astype(np.float64) - np.log(np.float64(f32(P)))).astype(f32)
    pp = np.full((3, T), -INF, dtype=f32)
    tt = np.zeros((3, T), dtype=np.int32)
    bestI = np.zeros((N, B), dtype=np.int32)
    bestT = np.zeros((N, B), dtype=np.int32)
    with np.errstate(over="ignore"):
        for k in range(N):
            row = lp[k]
            pIn = p[bin_] + PIP
            pStay = [pp[s] + tps[:, s] for s in range(3)]
            pGo = [pIn, pp[0] + tpt[:, 0], pp[1] + tpt[:, 1]]
            tGo = [np.full(T, k - 1, dtype=np.int32), tt[0].copy(), tt[1].copy()]
            for s in (2, 1, 0):
                go = pGo[s] > pStay[s]
                pp[s] = np.where(go, pGo[s], pStay[s]) + row[st[:, s]]
                tt[s] = np.where(go, tGo[s], tt[s])
            pOut = pp[2] + tpt[:, 2]
            for i in range(B):
                mx, mxi = -INF, 0
                seg = vals[off[i]:off[i + 1]]
                if len(seg):
                    cand = pOut[seg]
                    j = int(np.argmax(cand))               # first maximum, like the strict '<' scan
                    if mx < cand[j]:
                        mx, mxi = cand[j], int(seg[j])
                p[i] = mx + lm[i]
                bestI[k, i] = mxi
                bestT[k, i] = tt[2, mxi]
    ends = g["endIdxs"]
    mx, mxi = -INF, 0
    for i in range(P):
        if mx < p[ends[i]]:
            mx, mxi = p[ends[i]], int(ends[i])
    res = []
    t = N - 1
    while t >= 0:
        tri = int(bestI[t, mxi])
        end = t + 1
        t = int(bestT[t, mxi])
        res.insert(0, (t + 1, end, tri))
        mxi = int(bin_[tri])
    return res


# ---- the reference itself (only where oracle/_ref has been built) ---------------------------------------
_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libphonerecog_ref.so")


def reference_available():
    return os.path.isfile(_REF)


def reference_recog(img_path, lps, PIP, LMW):
    """Run the reference decoder (built from the reference sources) on a list of [N_u, S] arrays.
    -> list of [(begin, end, phone string)]."""
    lib = C.CDLL(_REF)
    lib.ref_recog_create.restype = C.c_void_p
    lib.ref_recog_create.argtypes = [C.c_char_p, C.c_float, C.c_float]
    lib.ref_recog_one.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int, C.c_int]
    lib.ref_recog_destroy.argtypes = [C.c_void_p]
    h = lib.ref_recog_create(img_path.encode(), PIP, LMW)
    assert h, "reference Init failed"
    out = []
    for lp in lps:
        lp = np.ascontiguousarray(lp, dtype=np.float32)
        n = lp.shape[0]
        b = (C.c_int * (n + 1))()
        e = (C.c_int * (n + 1))()
        buf = C.create_string_buffer(64 * (n + 1))
        k = lib.ref_recog_one(h, lp.ctypes.data_as(C.POINTER(C.c_float)), n, lp.shape[1], b, e, buf, len(buf), n + 1)
        assert k >= 0
        names = buf.raw.split(b"\0")[:k]
        out.append([(int(b[i]), int(e[i]), names[i].decode()) for i in range(k)])
    lib.ref_recog_destroy(h)
    return out

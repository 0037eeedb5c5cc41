"""Synthetic inputs for benchmarks and sweeps: random-init parameters in Chainer's key layout, utterance
lengths, and a decoding graph with the structure of a triphone loop in the image format CPhoneRecog::Init reads.

There is no TIMIT data, no trained model and no ``kaldiTri1909.img`` in the reference tree (missing blobs), so every
measurement runs on these.  Nothing here computes network outputs: these are input generators only, kept inside the
package so that the measurement scripts (bench.py, tools/) depend on nothing but the product.  The test-suite's CPU
restatement carries the same generators; the random streams are identical (checked by tests/test_host_io.py).
"""
import struct

import numpy as np


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

"""GPU Viterbi decoder (row n1) through the C ABI: bit-identical segmentations to the reference decoder's golden vectors,
to the numpy oracle on larger graphs, and end to end (posteriors decoded straight from device memory, PER scoring)."""
import os

import numpy as np
import pytest

from oracle import am_oracle as orc
from oracle import viterbi_oracle as vo

pytestmark = pytest.mark.gpu


def _logsoftmax(z):
    return (z - np.log(np.sum(np.exp(z), axis=1, keepdims=True))).astype(np.float32)


def test_segmentations_match_reference_golden(engine, golden_dir, tmp_path):
    from nnacousticmodeling_amd.phone_recog import PhoneRecog
    g = np.load(os.path.join(golden_dir, "viterbi.npz"))
    for ci in range(int(g["n_cases"])):
        PIP, LMW, n_utts = g["c%d_params" % ci]
        path = tmp_path / ("g%d.img" % ci)
        path.write_bytes(bytes(g["c%d_img" % ci]))
        dec = PhoneRecog(engine, str(path), float(PIP), float(LMW))
        lps = [g["c%d_u%d_lp" % (ci, u)] for u in range(int(n_utts))]
        offsets = np.concatenate([[0], np.cumsum([lp.shape[0] for lp in lps])])
        got = dec.recognise(np.concatenate(lps), offsets)
        for u in range(int(n_utts)):
            seg = g["c%d_u%d_seg" % (ci, u)]
            phones = list(g["c%d_u%d_phones" % (ci, u)])
            assert [(b, e) for b, e, _ in got[u]] == [(int(r[0]), int(r[1])) for r in seg]
            assert [p for _, _, p in got[u]] == phones
        dec.close()


def test_real_size_graph_matches_reference_golden_in_all_three_forms(engine, golden_dir):
    """The size of the absent kaldiTri1909.img: 48 phones, 2,304 nodes, 88,294 triphones, 1909 states, eight utterances of
    3..614 frames, two (PIP, LMW) settings.  tests/golden/viterbi_fullsize.npz holds only the seed and the segmentations the
    REFERENCE decoder (recog_src/source/phone_recog.cpp:138-254, built from source in the build container) produced; the
    graph and the log-posteriors are regenerated here from the seed.  All three kernel forms must reproduce them exactly."""
    import importlib.util
    from nnacousticmodeling_amd.phone_recog import PhoneRecog
    spec = importlib.util.spec_from_file_location("mkvf", os.path.join(golden_dir, "make_viterbi_fullsize_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    g = np.load(os.path.join(golden_dir, "viterbi_fullsize.npz"))
    graph, lps = mk.inputs(vo.make_graph)
    assert list(g["shape"]) == [mk.P, mk.S, len(graph["lm"]), len(graph["biphoneIdxIn"])] and list(g["lens"]) == mk.LENS
    offsets = np.concatenate([[0], np.cumsum(mk.LENS)])
    lp = np.concatenate(lps)
    gdict = dict(graph, numPhones=mk.P, numBiphones=len(graph["lm"]), numTriphones=len(graph["biphoneIdxIn"]))
    for pi, (PIP, LMW) in enumerate(g["params"]):
        dec = PhoneRecog(engine, gdict, float(PIP), float(LMW))
        for form in ("frame_sync", "per_utt_compact", "per_utt_wide", "auto"):
            dec.set_form(form)
            got = dec.recognise(lp, offsets)
            for u in range(len(mk.LENS)):
                be = g["p%d_u%d_begin_end" % (pi, u)]
                assert [(b, e) for b, e, _ in got[u]] == [(int(r[0]), int(r[1])) for r in be], (form, pi, u)
                assert [p for _, _, p in got[u]] == list(g["p%d_u%d_phones" % (pi, u)]), (form, pi, u)
        dec.close()


@pytest.mark.parametrize("P,S,PIP", [(12, 300, -4.0), (9, 1909, -20.0), (10, 128, 0.0)])
def test_larger_graphs_match_oracle(engine, P, S, PIP):
    from nnacousticmodeling_amd.phone_recog import PhoneRecog
    rng = np.random.default_rng(P * 1000 + S)
    graph = vo.make_graph(rng, P, S)
    lens = [3, 40, 97, 5, 211]
    lps = [_logsoftmax(4.0 * rng.standard_normal((n, S))) for n in lens]
    offsets = np.concatenate([[0], np.cumsum(lens)])
    dec = PhoneRecog(engine, dict(graph, numPhones=P, numBiphones=len(graph["lm"]), numTriphones=len(graph["biphoneIdxIn"])), PIP, 0.7)
    got = dec.recognise(np.concatenate(lps), offsets)
    for lp, segs in zip(lps, got):
        want = [(b, e, graph["phones"][t]) for b, e, t in vo.recog_one(graph, lp, PIP, 0.7)]
        assert segs == want
    if vo.reference_available():  # the reference itself, when its build travelled to this box
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            vo.write_image(os.path.join(td, "g.img"), graph)
            assert vo.reference_recog(os.path.join(td, "g.img"), lps, PIP, 0.7) == got
    # utterances shorter than 3 frames (the reference never terminates on them): zero segments, the rest unaffected
    short = dec.recognise(np.concatenate([lps[1][:2], lps[1]]), np.array([0, 2, 42]))
    assert short[0] == [] and short[1] == got[1]
    dec.close()


@pytest.mark.parametrize("P,S", [(12, 300), (48, 1909)])
def test_frame_sync_lds_form_matches_the_other_forms(engine, P, S, monkeypatch):
    """The frame-synchronous form stages the frame's log-posterior row and the node scores in LDS and takes the node maxima in
    the same launch (AM_VITERBI_LDS=1: in a launch of their own, from 12 active utterances on -- 24 or more: 2,048 triphones
    per staging, else 1,024 --, below that gathering from global memory like AM_VITERBI_LDS=0).  40 utterances of 3..90 frames walk through all three; every form must give the oracle's segmentations
    (small graph) / the per-utterance kernel's (real-size graph: the oracle on a sample)."""
    from nnacousticmodeling_amd.phone_recog import PhoneRecog
    rng = np.random.default_rng(P * 7 + S)
    graph = vo.make_graph(rng, P, S)
    lens = [int(v) for v in rng.integers(3, 91, size=40)]
    lps = [_logsoftmax(4.0 * rng.standard_normal((n, S))).astype(np.float32) for n in lens]
    offsets = np.concatenate([[0], np.cumsum(lens)])
    lp = np.concatenate(lps)
    dec = PhoneRecog(engine, dict(graph, numPhones=P, numBiphones=len(graph["lm"]), numTriphones=len(graph["biphoneIdxIn"])), -6.0, 0.8)
    dec.set_form("frame_sync")
    monkeypatch.delenv("AM_VITERBI_LDS", raising=False)
    stream = dec.recognise(lp, offsets)         # the default (round 4): the fused launch as a stream, a block walks several utterance slots
    monkeypatch.setenv("AM_VITERBI_STREAM", "3")   # three slot groups: every block walks up to 14 of the 40 slots (odd and even trip counts)
    stream3 = dec.recognise(lp, offsets)
    monkeypatch.setenv("AM_VITERBI_STREAM", "40")  # one slot per block
    stream40 = dec.recognise(lp, offsets)
    monkeypatch.delenv("AM_VITERBI_STREAM")
    monkeypatch.setenv("AM_VITERBI_LDS", "2")   # round 3's form: triphone update and node maxima in one launch, a block per (slot, triphone block)
    fused = dec.recognise(lp, offsets)
    assert stream == stream3 == stream40 == fused
    monkeypatch.setenv("AM_VITERBI_LDS", "1")   # staged gathers, node maxima in their own launch
    staged = dec.recognise(lp, offsets)
    monkeypatch.setenv("AM_VITERBI_LDS", "0")
    gathered = dec.recognise(lp, offsets)
    dec.set_form("per_utt_compact")
    per_utt = dec.recognise(lp, offsets)
    assert fused == staged == gathered == per_utt
    assert sum(len(r) for r in staged) > 40
    for u in (range(40) if P == 12 else (0, 17, 39)):
        want = [(b, e, graph["phones"][t]) for b, e, t in vo.recog_one(graph, lps[u], -6.0, 0.8)]
        assert staged[u] == want, u
    dec.close()


def test_arrival_lists_that_are_not_a_permutation(engine):
    """The decoder stores its state in arrival-list order when every triphone arrives at exactly one node; a graph where one
    triphone arrives nowhere and another at two nodes must take the general (gather) form and still match the oracle."""
    from nnacousticmodeling_amd.phone_recog import PhoneRecog
    rng = np.random.default_rng(5)
    P, S = 7, 90
    graph = vo.make_graph(rng, P, S)
    off, val = np.array(graph["outOffsets"]), np.array(graph["outValues"])
    node = int(np.argmax(np.diff(off) >= 2))
    val[off[node]] = val[off[node + 1]]          # first arrival of `node` replaced by one that also arrives at the next node
    # and one node with a long list (> 64 arrivals: the node kernel's unrolled trips plus its tail loop)
    lists = [list(val[off[i]:off[i + 1]]) for i in range(len(off) - 1)]
    lists[node + 1] += [int(t) for t in rng.integers(0, len(graph["biphoneIdxIn"]), 150)]
    off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
    val = np.array([t for l in lists for t in l], dtype=np.int32)
    graph = dict(graph, outOffsets=off, outValues=val)
    lens = [30, 75, 12]
    lps = [_logsoftmax(3.0 * rng.standard_normal((n, S))) for n in lens]
    dec = PhoneRecog(engine, dict(graph, numPhones=P, numBiphones=len(graph["lm"]), numTriphones=len(graph["biphoneIdxIn"])), -1.5, 0.8)
    got = dec.recognise(np.concatenate(lps), np.concatenate([[0], np.cumsum(lens)]))
    for lp, segs in zip(lps, got):
        assert segs == [(b, e, graph["phones"][t]) for b, e, t in vo.recog_one(graph, lp, -1.5, 0.8)]
    dec.close()


def test_decoder_forms_agree_in_fresh_processes():
    """Three kernels implement the decoder: frame-synchronous (default), one workgroup per utterance with compact records
    (AM_VITERBI_FS=0) and with 32-bit records (AM_VITERBI_COMPACT=0; what graphs with >= 65,536 nodes or columns get).  The
    choice is read once per process, so each runs in its own; all must give the default form's segmentations, incl. ties."""
    import subprocess
    import sys
    import tempfile
    code = r"""
import os, sys, json, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import viterbi_oracle as vo
from nnacousticmodeling_amd.engine import Engine
from nnacousticmodeling_amd.phone_recog import PhoneRecog
rng = np.random.default_rng(77)
graph = vo.make_graph(rng, 11, 400)
lens = [3, 57, 120, 4, 33, 250, 2, 19]
z = np.round(3.0 * rng.standard_normal((sum(lens), 400)), 1)          # coarse values: plenty of exact ties
lp = (z - np.log(np.sum(np.exp(z), axis=1, keepdims=True))).astype(np.float32)
off = np.concatenate([[0], np.cumsum(lens)])
dec = PhoneRecog(Engine(0), dict(graph, numPhones=11, numBiphones=len(graph["lm"]), numTriphones=len(graph["biphoneIdxIn"])), -2.0, 0.9)
json.dump(dec.recognise(lp, off), open(sys.argv[1], "w"))
"""
    outs = []
    for env in ({}, {"AM_VITERBI_FS": "0"}, {"AM_VITERBI_COMPACT": "0"}):
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "segs.json")
            res = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300,
                                 cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            assert res.returncode == 0, res.stderr[-2000:]
            outs.append(open(out).read())
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 100
    # and the oracle's (first strict maximum wins, phone_recog.cpp:190-200) on the same tie-heavy input
    import json
    rng = np.random.default_rng(77)
    graph = vo.make_graph(rng, 11, 400)
    lens = [3, 57, 120, 4, 33, 250, 2, 19]
    z = np.round(3.0 * rng.standard_normal((sum(lens), 400)), 1)
    lp = (z - np.log(np.sum(np.exp(z), axis=1, keepdims=True))).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)])
    got = json.loads(outs[0])
    for u, n in enumerate(lens):
        if n >= 3:
            want = [[b, e, graph["phones"][t]] for b, e, t in vo.recog_one(graph, lp[off[u]:off[u + 1]], -2.0, 0.9)]
            assert got[u] == want, u
        else:
            assert got[u] == []


def test_evaluate_end_to_end_with_gpu_decoder(engine, tmp_path, real_ft_path, real_priors):
    """evaluate CLI on a synthetic 1909-state graph: posteriors -> in-process GPU Viterbi -> 60->39 map -> NIST PER."""
    from nnacousticmodeling_amd import chainer_npz, evaluate, scoring
    from nnacousticmodeling_amd.phone_recog import PhoneRecog
    rng = np.random.default_rng(2024)
    data_dir, recog_dir, lab_dir = tmp_path / "data", tmp_path / "recog", tmp_path / "lab"
    data_dir.mkdir(), recog_dir.mkdir()
    open(data_dir / "final.feature_transform", "w").write(open(real_ft_path).read())
    np.save(recog_dir / "log_ap_Kaldi1909.npy", real_priors)
    golden = os.path.dirname(real_ft_path)
    open(recog_dir / "phones.60-48-39.map", "w").write(open(os.path.join(golden, "phones.60-48-39.map")).read())
    phones60 = [l.split()[1] for l in open(os.path.join(golden, "phones.60-48-39.map")) if len(l.split()) == 3]
    graph = vo.make_graph(rng, 8, 1909)
    graph["phones"] = [phones60[int(p[1:]) * 7 % len(phones60)] for p in graph["phones"]]  # real 60-set labels
    vo.write_image(str(recog_dir / "kaldiTri1909.img"), graph)
    n_utts, n = 6, 700
    off = orc.make_offsets(rng, n_utts, n, lo=40, hi=200)
    x = rng.standard_normal((n, 40)).astype(np.float32)
    np.save(data_dir / "data_test.npy", x)
    np.save(data_dir / "offsets_test.npy", off)
    names = ["SPK%d_SX%d" % (u, u) for u in range(n_utts)]
    (data_dir / "test.list").write_text("".join(s + "\n" for s in names))
    # reference transcriptions: perturb nothing -> use the decoder's own output of a first run as "truth" for utterance 0..2
    params = orc.make_mlp_params(rng, 440, 128, 2, 1909)
    chainer_npz.write_npz(tmp_path / "model", params)
    base = ["-n", "ff", "-l", 2, "-u", 128, "--splice", 5, "--tri", "--data-dir", data_dir, "--offset-dir", data_dir, "--utt-list-dir", data_dir,
            "--recog-dir", recog_dir, "-m", tmp_path / "model", "--tmp-dir", lab_dir, "--no-progress", "--PIP", 3, "--LMW", 0.5]
    # first pass without a reference MLF present -> write a reference from the oracle decode of the oracle posteriors
    ft = orc.load_kaldi_feature_transform(real_ft_path)
    post = orc.evaluate_posteriors(orc.MLP(params, 2), orc.evaluate_frontend(x, 5, ft), off, ap=real_priors)
    pmap = {l.split()[1]: l.split()[2] for l in open(os.path.join(golden, "phones.60-48-39.map")) if len(l.split()) == 3}
    with open(recog_dir / "test_ref.mlf", "w") as f:
        f.write("#!MLF!#\n")
        for u in range(n_utts):
            segs = vo.recog_one(graph, post[u], -3.0, 0.5)
            words = [pmap[graph["phones"][t]] for _, _, t in segs]
            if u % 2:
                words = words[:-1] + ["sil"]  # make some utterances imperfect
            f.write('"*/%s.lab"\n%s\n.\n' % (names[u], "\n".join(words)))
    per = evaluate.main(base)
    assert per is not None and 0.0 <= per < 100.0
    # the MLF the GPU decoder wrote == oracle Viterbi on the posteriors the GPU wrote to the .lab files (bit-identical paths)
    from nnacousticmodeling_amd.kw_io import loadBin
    got = scoring.loadMlf(str(lab_dir / "vysledek_test.txt"))
    for u in range(n_utts):
        lp = loadBin(str(lab_dir / (names[u] + ".lab")))
        want = vo.recog_one(graph, lp, -3.0, 0.5)
        assert got[names[u]][0] == [graph["phones"][t] for _, _, t in want]
        assert got[names[u]][1] == [b * 100000 for b, _, _ in want] and got[names[u]][2] == [e * 100000 for _, e, _ in want]
    want_per = scoring.computeWER(scoring.loadMlf(str(lab_dir / "vysledek_test_p39.txt")), scoring.loadMlf(str(recog_dir / "test_ref.mlf")), True)
    assert per == want_per
    # --no-lab on one GPU: the posteriors never leave the device (resident pipeline -> recognise_pipeline); same MLF, same PER
    mlf_with_lab = (lab_dir / "vysledek_test.txt").read_text()
    lab2 = tmp_path / "lab_resident"
    per2 = evaluate.main([a if a is not lab_dir else lab2 for a in base] + ["--no-lab"])
    assert per2 == per
    assert (lab2 / "vysledek_test.txt").read_text().replace(str(lab2), str(lab_dir)) == mlf_with_lab
    assert not list(lab2.glob("*.lab"))
    # ... and on several shards (three engines on this one card stand in for three GPUs): every shard decodes its own utterances
    # from its own device memory and writes its own .lab files; same MLF, same PER, same .lab bytes (rows do not depend on the shard)
    from nnacousticmodeling_amd import networks
    from nnacousticmodeling_amd.evaluate_model_for_test import evaluateModelTestTri
    spec = networks.get_nn("ff", 2, [128], 1909).set_params(params)
    shifts = list(range(-5, 6))
    for tag, write_lab in (("lab_sharded", True), ("lab_sharded_nolab", False)):
        lab3 = tmp_path / tag
        per3 = evaluateModelTestTri(spec, x, off, 3, 0.5, ap=real_priors, GPUID=[(0, 0), (0, 1), (0, 2)], testOrDev="test", tmpDir=str(lab3),
                                    uttlistdir=str(data_dir), recogdir=str(recog_dir), progress=False, frontend=(shifts, ft), run_decoder="gpu",
                                    write_lab=write_lab)
        assert per3 == per
        assert (lab3 / "vysledek_test.txt").read_text().replace(str(lab3), str(lab_dir)) == mlf_with_lab
        if write_lab:
            for nm in names:
                assert (lab3 / (nm + ".lab")).read_bytes() == (lab_dir / (nm + ".lab")).read_bytes()
            assert (lab3 / "test.scp").read_text().replace(str(lab3), str(lab_dir)) == (lab_dir / "test.scp").read_text()
        else:
            assert not list(lab3.glob("*.lab"))
    # decoding straight from device memory gives the same segmentation as decoding the downloaded posteriors
    from nnacousticmodeling_amd.engine import Model, Pipeline
    p = Pipeline(engine)
    p.set_frontend(40, ft["shifts"], ft["addShift"], ft["rescale"])
    p.set_members([Model(engine, "ff", 2, params)])
    p.set_prior(real_priors)
    y = p.forward_ff(x)
    dec = PhoneRecog(engine, str(recog_dir / "kaldiTri1909.img"), -3.0, 0.5)
    assert dec.recognise_pipeline(p, off) == dec.recognise(y, off)
    dec.close()

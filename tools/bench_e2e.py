#!/usr/bin/env python3
"""Where the wall time of the two surfaces goes END TO END: files in -> files out, one MI355X.

Surface A (evaluate.py): load model / data -> posteriors (H2D, compute, D2H) -> 192 .lab files + .scp (what only the
external PhoneRecog needs) -> in-process GPU decoder -> MLF -> 60->39 mapping -> PER.  Surface B (predict_folds.py): one
[N, 1909] .npy.  Synthetic core-test-like set (192 utterances / 57,919 frames), synthetic real-size decoding graph, real
phone map; the reference MLF is the decoder's own output, so the PER printed is 0 by construction -- this is a timing tool.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnacousticmodeling_amd import chainer_npz, scoring  # noqa: E402
from nnacousticmodeling_amd.evaluate_model_for_test import compute_posteriors, write_posteriors  # noqa: E402
from nnacousticmodeling_amd.kaldi_ft import loadKaldiFeatureTransform  # noqa: E402
from nnacousticmodeling_amd.networks import get_nn  # noqa: E402
from nnacousticmodeling_amd.phone_recog import PhoneRecog, base_name_copy, write_mlf  # noqa: E402
from nnacousticmodeling_amd.sharding import get_runner  # noqa: E402
from nnacousticmodeling_amd import synthetic as orc  # noqa: E402  (synthetic inputs; no oracle in measurement scripts)
from nnacousticmodeling_amd import synthetic as vo  # noqa: E402


def main():
    rng = np.random.default_rng(20181806)
    golden = os.path.join(ROOT, "tests", "golden")
    N, U = 57919, 192
    t = {}
    with tempfile.TemporaryDirectory() as td:
        # ---- the files a user would have ----------------------------------------------------------------------
        off = orc.make_offsets(rng, U, N)
        np.save(os.path.join(td, "data_test.npy"), rng.standard_normal((N, 40)).astype(np.float32))
        np.save(os.path.join(td, "offsets_test.npy"), off)
        names = ["DR%d_SPK%03d_SX%d" % (u % 8, u, 100 + u) for u in range(U)]
        open(os.path.join(td, "test.list"), "w").write("".join(n + "\n" for n in names))
        chainer_npz.write_npz(os.path.join(td, "model"), orc.make_mlp_params(rng, 440, 1024, 4, 1909))
        phones = sorted({l.split()[1] for l in open(os.path.join(golden, "phones.60-48-39.map")) if len(l.split()) == 3})   # the 48-phone set
        graph = vo.make_graph(rng, 48, 1909)
        graph["phones"] = [phones[int(n[1:])] for n in graph["phones"]]   # make_graph labels a triphone "p<centre phone>": use real names
        vo.write_image(os.path.join(td, "kaldiTri1909.img"), graph)

        # ---- surface A ------------------------------------------------------------------------------------------
        c = time.perf_counter
        t0 = c()
        model = chainer_npz.load_npz(os.path.join(td, "model"), get_nn("ff", 4, [1024], 1909, "relu"))
        ft = loadKaldiFeatureTransform(os.path.join(golden, "final.feature_transform"))
        ap = np.load(os.path.join(golden, "log_ap_Kaldi1909.npy"))
        data = np.load(os.path.join(td, "data_test.npy"))
        offsets = np.load(os.path.join(td, "offsets_test.npy"))
        t["load model + data + transform + priors"] = c() - t0
        fe = (list(range(-5, 6)), ft)
        t0 = c()
        y = compute_posteriors(model, data, offsets, ap=ap, devices=[0], frontend=fe)
        t["posteriors, first call (engine + weight upload + H2D + compute + D2H)"] = c() - t0
        t0 = c()
        y = compute_posteriors(model, data, offsets, ap=ap, devices=[0], frontend=fe)
        t["posteriors, second call (as above without engine / weights; page-locks the 442 MB output pool, ~0.1 s)"] = c() - t0
        get_runner([0]).release(y)   # the lease goes back to the pool: the next call downloads into the same page-locked block
        del y
        t0 = c()
        y = compute_posteriors(model, data, offsets, ap=ap, devices=[0], frontend=fe)
        t["posteriors, steady state with release() (H2D + compute + D2H into the pooled page-locked output)"] = c() - t0
        del y                        # dropped, not released (the reference's contract): the pool answers with pageable memory from now on
        t0 = c()
        y = compute_posteriors(model, data, offsets, ap=ap, devices=[0], frontend=fe)
        t["posteriors, steady state without release() (pageable output)"] = c() - t0
        lab = os.path.join(td, "lab")
        os.makedirs(lab)
        t0 = c()
        write_posteriors(y, offsets, [n + "\n" for n in names], lab, "test")
        t["192 .lab files + .scp (442 MB; only the external PhoneRecog needs them)"] = c() - t0
        t0 = c()
        dec = PhoneRecog(get_runner([0])._dev(0).engine, os.path.join(td, "kaldiTri1909.img"), -20.0, 1.0)
        t["decoder: read image + upload graph"] = c() - t0
        t0 = c()
        res = dec.recognise(y, offsets)
        t["decoder: H2D of the posteriors + Viterbi + segments back"] = c() - t0
        t0 = c()
        mlf = os.path.join(lab, "vysledek_test.txt")
        write_mlf(mlf, [base_name_copy(os.path.join(lab, n + ".lab")) for n in names], res)
        p39 = os.path.join(lab, "vysledek_test_p39.txt")
        scoring.convert_results(os.path.join(golden, "phones.60-48-39.map"), mlf, p39)
        per = scoring.computeWER(scoring.loadMlf(p39), scoring.loadMlf(p39), True)
        t["MLF + 60->39 mapping + NIST alignment (PER)"] = c() - t0
        dec.close()
        # ---- surface B ------------------------------------------------------------------------------------------
        from nnacousticmodeling_amd.predict_folds import predict
        del y
        t0 = c()
        yb = predict(model, data, None, 1909, "ff", [0], 11, 0, ft, progress=False)
        t["surface B: predict() steady state (H2D + compute + D2H)"] = c() - t0
        t0 = c()
        np.save(os.path.join(td, "data_0.npy"), yb)
        t["surface B: np.save of the [N,1909] posteriors (442 MB)"] = c() - t0
    print(json.dumps({"frames": N, "utterances": U, "per_self_reference": per, "seconds": {k: round(v, 4) for k, v in t.items()}}, indent=1))


if __name__ == "__main__":
    main()

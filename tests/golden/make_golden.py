#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own numpy utilities.

Run ONCE in the build container (where /root/reference exists); the resulting ``*.npz`` /
``*.bin`` fixtures are committed and are the only thing the tests read -- the reference itself
never travels.  Only ``scripts/util/kw_utils.py`` and ``scripts/util/kw_nn_utils.py`` are imported
(numpy-only modules); everything that needs Chainer cannot be imported here.

Fixtures (SURVEY.md 8c list):
  1. ft_parse.npz        loadKaldiFeatureTransform(kaldi/final.feature_transform)
  2. splice.npz          splicing() on seeded [257,40] + prepareBatch() on index batches
  3. ft_apply.npz        applyKaldiFeatureTransform() on (2)
  4. logsum.npz          logsum()/normalise on seeded [16,1909] incl. -inf row and +80 outlier
  5. savebin_2d.bin / savebin_1d.bin + savebin.npz   saveBin() bytes
  6. prior_sub.npz       y - ap_coef*ap with the real priors file
Data files copied verbatim (data, not source): final.feature_transform, log_ap_Kaldi1909.npy.
"""
import os
import shutil
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = os.environ.get("AM_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, os.path.join(REF, "scripts", "util"))
import kw_nn_utils  # noqa: E402
import kw_utils  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20181806


def main():
    rng = np.random.default_rng(SEED)

    # data files
    shutil.copyfile(os.path.join(REF, "kaldi", "final.feature_transform"), os.path.join(HERE, "final.feature_transform"))
    shutil.copyfile(os.path.join(REF, "recog", "log_ap_Kaldi1909.npy"), os.path.join(HERE, "log_ap_Kaldi1909.npy"))

    # 1. ft parse
    ft = kw_nn_utils.loadKaldiFeatureTransform(os.path.join(REF, "kaldi", "final.feature_transform"))
    np.savez(os.path.join(HERE, "ft_parse.npz"), shape=np.asarray(ft["shape"]), shifts=np.asarray(ft["shifts"]),
             addShift=ft["addShift"], rescale=ft["rescale"])

    # 2. splice + prepareBatch
    data = rng.standard_normal((257, 40)).astype(np.float32)
    spliced5 = kw_utils.splicing(data, range(-5, 6))
    spliced2 = kw_utils.splicing(data, range(-2, 3))
    idx_a = np.arange(0, 64)
    idx_b = np.arange(200, 257)
    idx_c = np.arange(100, 140)
    pb_a, _ = kw_nn_utils.prepareBatch(data, [], idx_a.copy(), 11)
    pb_b, _ = kw_nn_utils.prepareBatch(data, [], idx_b.copy(), 11)
    pb_c, _ = kw_nn_utils.prepareBatch(data, [], idx_c.copy(), 5)
    tiny = rng.standard_normal((3, 40)).astype(np.float32)  # file shorter than the window
    tiny5 = kw_utils.splicing(tiny, range(-5, 6))
    np.savez(os.path.join(HERE, "splice.npz"), data=data, spliced5=spliced5, spliced2=spliced2,
             idx_a=idx_a, idx_b=idx_b, idx_c=idx_c, pb_a=pb_a, pb_b=pb_b, pb_c=pb_c, tiny=tiny, tiny5=tiny5)

    # 3. ft apply (bit-exact expected)
    applied = kw_nn_utils.applyKaldiFeatureTransform(spliced5, ft)
    np.savez(os.path.join(HERE, "ft_apply.npz"), applied=applied)

    # 4. logsum
    lp = (3.0 * rng.standard_normal((16, 1909))).astype(np.float32)
    lp[3, :] = -np.inf            # NaN -> -1e20 rule
    lp[5, 77] = 80.0              # outlier
    lp[7, :] = lp[7, 0]           # constant row
    lp[9, 100] = np.inf           # +inf -> NaN -> -1e20
    with np.errstate(all="ignore"):
        lps = kw_utils.logsum(lp, axis=1)
        norm = lp - lps
    np.savez(os.path.join(HERE, "logsum.npz"), lp=lp, lps=lps, norm=norm)

    # 5. saveBin
    y2 = rng.standard_normal((4, 1909)).astype(np.float32)
    y1 = rng.standard_normal(7).astype(np.float32)
    kw_utils.saveBin(os.path.join(HERE, "savebin_2d.bin"), y2)
    kw_utils.saveBin(os.path.join(HERE, "savebin_1d.bin"), y1)
    np.savez(os.path.join(HERE, "savebin.npz"), y2=y2, y1=y1,
             back2=kw_utils.loadBin(os.path.join(HERE, "savebin_2d.bin"), np.float32),
             back1=kw_utils.loadBin(os.path.join(HERE, "savebin_1d.bin"), np.float32))

    # 6. prior subtraction with the real priors (evaluate.py:174, evaluateModelForTest.py:110-112)
    ap = np.load(os.path.join(REF, "recog", "log_ap_Kaldi1909.npy"))
    y = (2.0 * rng.standard_normal((8, 1909))).astype(np.float32)
    out = {}
    for coef in (1.0, 0.5):
        yy = y - coef * ap
        yy = yy - kw_utils.logsum(yy, axis=1)
        out["coef_%g" % coef] = yy
    np.savez(os.path.join(HERE, "prior_sub.npz"), y=y, **{k.replace(".", "p"): v for k, v in out.items()})
    # 7. scoring chain: loadMlf / levenshtein_nist / computeWER / convert_results semantics
    import json
    import tempfile
    import levenshtein  # reference scripts/util/levenshtein.py (numpy-only)
    sys.path.insert(0, os.path.join(REF, "scripts", "util"))
    shutil.copyfile(os.path.join(REF, "recog", "phones.60-48-39.map"), os.path.join(HERE, "phones.60-48-39.map"))
    ref_mlf = kw_utils.loadMlf(os.path.join(REF, "recog", "test_ref.mlf"))
    utts = sorted(ref_mlf.keys(), key=lambda u: ref_mlf[u][3])[:40]
    phones = sorted({w for u in utts for w in ref_mlf[u][0]})
    prng = np.random.default_rng(SEED + 7)
    cases = []
    for u in utts:
        ref_words = list(ref_mlf[u][0])
        hyp = []
        for w in ref_words:
            r = prng.random()
            if r < 0.08:
                continue                                   # deletion
            if r < 0.18:
                hyp.append(phones[prng.integers(len(phones))])  # substitution
            else:
                hyp.append(w)
            if prng.random() < 0.07:
                hyp.append("sil" if prng.random() < 0.5 else phones[prng.integers(len(phones))])  # insertion
        if prng.random() < 0.3 and hyp and hyp[0] == "sil":
            hyp = hyp[1:]                                  # optional leading silence dropped
        if prng.random() < 0.2:
            hyp = hyp[: max(1, len(hyp) // 2)]             # much shorter hypothesis -> flipped alignment
        nist = [int(v) for v in levenshtein.levenshtein_nist(hyp, ref_words)]
        plain = [int(v) for v in levenshtein.levenshtein2(hyp, ref_words)]
        cases.append({"utt": u, "ref": ref_words, "hyp": hyp, "nist": nist, "plain": plain})
    test_mlf = {c["utt"]: [c["hyp"], [], [], i] for i, c in enumerate(cases)}
    sub_ref = {c["utt"]: ref_mlf[c["utt"]] for c in cases}
    wer_nist = levenshtein.computeWER(test_mlf, sub_ref, True)
    wer_plain = levenshtein.computeWER(test_mlf, sub_ref, False)
    # loadMlf on a recogniser-style MLF with times, '#' comments, a blank line and trailing junk after '.'
    rec_text = '#!MLF!#\n"lab\\\\FDHC0_SI1559.lab"\n0 300000 sil\n300000 900000 ax-h\n900000 1200000 el\n.\njunk after dot\n' \
               '"*/MABC0_SX1.lab"\nsil\n\nhh\nzh\n.\n'
    with tempfile.TemporaryDirectory() as td:
        rec = os.path.join(td, "rec.mlf")
        open(rec, "w").write(rec_text)
        loaded = kw_utils.loadMlf(rec)
        # convert_results lives in evaluateModelForTest.py (imports chainer) -> cannot be imported; its expected output is
        # derived here from the map file by the documented rule and checked by hand (3 mapped phones below).
    loaded_json = {k: [v[0], v[1], v[2], v[3]] for k, v in loaded.items()}
    json.dump({"cases": cases, "wer_nist": wer_nist, "wer_plain": wer_plain, "n_ref_utts": len(ref_mlf),
               "first_ref_utt": utts[0], "first_ref_words": ref_mlf[utts[0]][0], "rec_text": rec_text, "rec_loaded": loaded_json},
              open(os.path.join(HERE, "scoring.json"), "w"), indent=0)
    shutil.copyfile(os.path.join(REF, "recog", "test_ref.mlf"), os.path.join(HERE, "test_ref.mlf"))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

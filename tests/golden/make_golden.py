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
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()

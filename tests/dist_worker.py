"""Worker for tests/test_dist_gloo.py: one process per "GPU" (oracle-backed stand-in), launched by torch.distributed.run.

Each rank computes its frame shard (with halo) of one file, rank 0 gathers on the host and compares with the whole-file
oracle -- the N > 1 layout of bench.py / multi-process launches, on CPU with gloo.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nnacousticmodeling_amd import networks, sharding  # noqa: E402
from bench_ranks import Ranks  # noqa: E402
from oracle import am_oracle as orc  # noqa: E402
from tests import fake_backend  # noqa: E402


def main():
    ranks = Ranks()
    rng = np.random.default_rng(123)  # same file on every rank
    ft = orc.load_kaldi_feature_transform(os.path.join(ROOT, "tests", "golden", "final.feature_transform"))
    params = orc.make_mlp_params(rng, 440, 32, 2, 39)
    n = 333
    x = rng.standard_normal((n, 40)).astype(np.float32)
    spec = networks.get_nn("ff", 2, [32], 39).set_params(params)
    shards = sharding.ff_shards(n, ranks.world, 5)
    lo, hi, h_lo, h_hi = shards[ranks.rank]
    runner = sharding.PosteriorRunner([ranks.local_rank], backend=fake_backend)
    ranks.barrier()
    t0 = time.perf_counter()
    # this rank's shard only: the runner sees a "file" made of the shard plus its halo rows
    st = runner._dev(ranks.local_rank)
    pipe = runner._configure(st, networks.as_ensemble(spec), 40, list(range(-5, 6)), ft, 0, None, 0, True)
    y_local = pipe.forward_ff(x[lo - h_lo:hi + h_hi], None, hi - lo, h_lo, h_hi)
    wall = ranks.max_over_ranks(time.perf_counter() - t0)
    ranks.barrier()
    y = ranks.gather_rows(y_local, n)
    if ranks.rank == 0:
        want = orc.predict(orc.MLP(params, 2), x, None, False, 11, 0, ft)
        err = float(np.max(np.abs(y - want)))
        assert y.shape == want.shape and err < 1e-5, err
        assert wall > 0
        print("DIST_OK world=%d err=%.3g" % (ranks.world, err))
    ranks.close()


if __name__ == "__main__":
    main()

"""One-process-per-GPU plumbing of bench.py and tests/dist_worker.py (not part of the product package, which imports no torch): rank discovery, barrier, max-over-ranks and the
host-side gather of per-rank posterior shards.  The data path itself needs no collective (frames / utterances shard
embarrassingly); ``torch.distributed`` (gloo, CPU tensors) is used only for these control-plane steps.
"""
from __future__ import annotations

import os

import numpy as np


class Ranks:
    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self._dist = dist

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def gather_objects(self, obj):
        """Every rank's object, in rank order, on every rank."""
        if self._dist is None:
            return [obj]
        got = [None] * self.world
        self._dist.all_gather_object(got, obj)
        return got

    def gather_rows(self, local, n_total):
        """Host-side gather of contiguous row shards (rank order = row order) into one array on rank 0."""
        local = np.ascontiguousarray(local)
        if self._dist is None:
            assert local.shape[0] == n_total
            return local
        import torch
        counts = [None] * self.world
        self._dist.all_gather_object(counts, int(local.shape[0]))
        assert sum(counts) == n_total, (counts, n_total)
        if self.rank == 0:
            out = np.empty((n_total,) + local.shape[1:], dtype=local.dtype)
            out[:counts[0]] = local
            pos = counts[0]
            for src in range(1, self.world):
                buf = torch.empty((counts[src],) + local.shape[1:], dtype=torch.from_numpy(local[:0]).dtype)
                if counts[src]:
                    self._dist.recv(buf, src=src)
                out[pos:pos + counts[src]] = buf.numpy()
                pos += counts[src]
            return out
        if local.shape[0]:
            self._dist.send(torch.from_numpy(local), dst=0)
        return None

    def close(self):
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None

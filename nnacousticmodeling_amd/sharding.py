"""Shard the posterior computation over the GPUs of one node and gather on the host.

Frames (feed-forward) and utterances (recurrent) are independent, so the path shards with no collective
(SURVEY.md 8e): contiguous frame ranges with a +-context read halo for ``ff`` -- the splice clamps only at the global
ends of the file (reference kw_utils.py:29-33), never at shard or utterance boundaries -- and longest-first bins of
whole utterances for ``lstm``.  One host thread drives each GPU through its own engine (ctypes releases the GIL);
the gather is a plain copy into the single output array.
"""
from __future__ import annotations

import threading

import numpy as np

from ._lib import AM_COMBINE_LOGITS, AM_COMBINE_LOGPOST
from .networks import as_ensemble


# ---------------------------------------------------------------------------------------------------
# pure partitioning helpers
# ---------------------------------------------------------------------------------------------------
def ff_shards(n_frames, n_parts, context):
    """Contiguous ranges [lo, hi) plus the halo rows that exist around them: [(lo, hi, halo_lo, halo_hi), ...].
    Empty ranges are dropped."""
    n_parts = max(1, min(int(n_parts), max(int(n_frames), 1)))
    out = []
    for g in range(n_parts):
        lo = n_frames * g // n_parts
        hi = n_frames * (g + 1) // n_parts
        if hi > lo:
            out.append((lo, hi, min(context, lo), min(context, n_frames - hi)))
    return out


def lstm_bins(offsets, n_parts):
    """Greedy longest-first packing of utterances onto ``n_parts`` bins, balancing total frames.
    -> list of utterance-index arrays (each in file order), empty bins dropped."""
    offsets = np.asarray(offsets, dtype=np.int64)
    lens = np.diff(offsets)
    n_parts = max(1, min(int(n_parts), max(len(lens), 1)))
    order = np.argsort(-lens, kind="stable")
    load = np.zeros(n_parts, dtype=np.int64)
    bins = [[] for _ in range(n_parts)]
    for u in order:
        b = int(np.argmin(load))
        bins[b].append(int(u))
        load[b] += lens[u]
    return [np.sort(np.asarray(b, dtype=np.int64)) for b in bins if b]


def gather_utterances(x, offsets, utts):
    """Rows of the utterances ``utts`` concatenated, plus their local offsets."""
    offsets = np.asarray(offsets, dtype=np.int64)
    parts = [x[offsets[u]:offsets[u + 1]] for u in utts]
    local = np.zeros(len(utts) + 1, dtype=np.int64)
    local[1:] = np.cumsum([p.shape[0] for p in parts])
    cat = np.concatenate(parts, axis=0) if parts else np.zeros((0, x.shape[1]), dtype=x.dtype)
    return np.ascontiguousarray(cat), local


# ---------------------------------------------------------------------------------------------------
# runner
# ---------------------------------------------------------------------------------------------------
class _ShardState:
    """One logical shard: its own engine (streams, workspace), pipeline and resident models on one GPU."""

    def __init__(self, backend, device):
        self.engine = backend.Engine(device)
        self.models = []          # [(spec, Model)]: holds the spec itself, never a bare id() (ids are reused after GC)
        self.pipeline = backend.Pipeline(self.engine)

    def model_for(self, backend, spec):
        for sp, m in self.models:
            if sp is spec:
                return m
        if spec.params is None:
            raise ValueError("network has no parameters loaded (chainer_npz.load_npz)")
        m = backend.Model(self.engine, spec.network, spec.layers, spec.params, spec.activation)
        self.models.append((spec, m))
        return m

    def evict_except(self, keep):
        """Close the device copies of every model that is not part of the ensemble being configured (per-fold mode
        walks through the folds one at a time: the previous fold's weights leave HBM here)."""
        kept = []
        for sp, m in self.models:
            if any(sp is k for k in keep):
                kept.append((sp, m))
            else:
                m.close()
        self.models = kept


def _normalise_shards(devices):
    """``devices``: GPU ids, or (gpu, replica) pairs.  Several logical shards may share one GPU (each gets its own engine:
    two shards of one card overlap one's transfers with the other's compute, and the threaded shard / gather code runs on a
    single-GPU box exactly as it does on eight).  -> [(gpu, replica)] with replicas numbered per GPU."""
    out, seen = [], {}
    for d in devices:
        if isinstance(d, (tuple, list)):
            gpu, rep = int(d[0]), int(d[1])
        else:
            gpu = int(d)
            rep = seen.get(gpu, 0)
        if gpu < 0:
            raise ValueError("negative GPU id (CPU execution) is not supported: this path has no CPU fallback")
        seen[gpu] = max(seen.get(gpu, 0), rep + 1)
        if (gpu, rep) in out:
            raise ValueError("shard (%d, %d) listed twice" % (gpu, rep))
        out.append((gpu, rep))
    if not out:
        raise ValueError("no device given")
    return out


class PosteriorRunner:
    """Runs an ensemble (``NetworkSpec`` or ``NNWithRPL``) on one or more GPUs.

    ``devices``: list of device ids, or of ``(device, replica)`` pairs for several logical shards per GPU.
    ``backend``: module providing Engine/Model/Pipeline (default: the HIP engine; the CPU-only test-suite injects a
    stand-in to exercise this host logic without a GPU)."""

    def __init__(self, devices=(0,), backend=None, precision="f32"):
        if backend is None:
            from . import engine as backend
        self.backend = backend
        self.precision = precision   # "f32" (exact f32 MFMA) or "split_f16" (f32-accurate on the f16 pipe, feed-forward only)
        self.shards = _normalise_shards(devices)
        self.devices = [g for g, _ in self.shards]
        self._state = {}
        self._pinned = {}            # (n, c) -> page-locked blocks handed out by _out
        self._requests = {}          # (n, c) -> how often that shape has been asked for
        self._lock = threading.Lock()

    def _dev(self, shard):
        if not isinstance(shard, tuple):
            shard = (int(shard), 0)
        with self._lock:
            st = self._state.get(shard)
            if st is None:
                st = self._state[shard] = _ShardState(self.backend, shard[0])
                if self.precision != "f32":
                    st.engine.set_precision(self.precision)
            return st

    def _out(self, n, c):
        """The gathered [N, C] array.  Page-locked when the backend offers it (am_host_alloc): every shard's download then
        runs at PCIe rate straight into its slice, instead of being staged through the driver's bounce buffers.
        Page-locking 442 MB costs ~0.1 s, more than a warm pass over a core-test-sized file, so the blocks are pooled: a block
        is handed out again once nothing but the pool refers to it (the caller has dropped the array and every view of it)."""
        alloc = getattr(self.backend, "pinned_empty", None)
        if alloc is None or n * c == 0:
            return np.empty((n, c), dtype=np.float32)
        import sys
        with self._lock:
            seen = self._requests[(n, c)] = self._requests.get((n, c), 0) + 1
            if len(self.shards) == 1 and seen == 1:
                # one shard, first request of this shape: a one-shot call is faster into pageable memory (43 ms for the
                # core-test file) than into memory it first has to page-lock (~100 ms); a repeated workload gets the pool
                return np.empty((n, c), dtype=np.float32)
            pool = self._pinned.setdefault((n, c), [])
            for y in pool:
                if sys.getrefcount(y) <= 3:      # the pool's list, the loop variable, getrefcount's argument
                    return y
            y = alloc((n, c), np.float32)
            pool.append(y)
            if len(pool) > 4:                    # callers that keep many results alive: stop pooling the oldest
                pool.pop(0)
            return y

    def _configure(self, st, ensemble, D, shifts, ft, d_iv, prior, combine, normalise):
        specs = list(ensemble.members())
        members = [st.model_for(self.backend, spec) for spec in specs]
        p = st.pipeline
        add, scale = (None, None) if ft is None else (ft["addShift"], ft["rescale"])
        p.set_frontend(D, shifts, add, scale, d_iv)
        p.set_members(members, has_master=ensemble.has_master and combine == AM_COMBINE_LOGITS, combine=combine)
        st.evict_except(specs)   # after set_members: the pipeline no longer refers to the models closed here
        rpl = ensemble.rpl
        if rpl is not None:
            p.set_rpl4(rpl.W, rpl.b, rpl.lb)
        else:
            p.set_rpl4(None, None, None)
        p.set_prior(prior)
        p.set_normalise(normalise)
        return p

    def _run_parallel(self, jobs):
        """jobs: list of (shard, callable).  One thread per job; first exception is re-raised."""
        if len(jobs) == 1:
            jobs[0][1]()
            return
        errors = []

        def wrap(fn):
            try:
                fn()
            except BaseException as exc:  # noqa: BLE001 - re-raised below
                errors.append(exc)

        threads = [threading.Thread(target=wrap, args=(fn,)) for _, fn in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    # feed-forward ----------------------------------------------------------------------------------
    def forward_ff(self, model, x, shifts=(0,), ft=None, ivectors=None, prior=None, combine=AM_COMBINE_LOGITS, normalise=True):
        """x: [N, D] raw frames of the whole file.  Returns float32 [N, C] in file order."""
        ens = as_ensemble(model)
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, D = x.shape
        iv = None if ivectors is None else np.ascontiguousarray(ivectors, dtype=np.float32)
        d_iv = 0 if iv is None else iv.shape[1]
        shifts = [int(s) for s in shifts]
        ctx = max(abs(s) for s in shifts)
        y = self._out(n, ens.num_classes)
        shards = ff_shards(n, len(self.shards), ctx)
        jobs = []
        for (lo, hi, h_lo, h_hi), shard in zip(shards, self.shards):
            def job(lo=lo, hi=hi, h_lo=h_lo, h_hi=h_hi, shard=shard):
                st = self._dev(shard)
                p = self._configure(st, ens, D, shifts, ft, d_iv, prior, combine, normalise)
                p.forward_ff(x[lo - h_lo:hi + h_hi], None if iv is None else iv[lo:hi], hi - lo, h_lo, h_hi, out=y[lo:hi])
            jobs.append((shard, job))
        if jobs:
            self._run_parallel(jobs)
        return y

    # recurrent -------------------------------------------------------------------------------------
    def forward_lstm(self, model, x, offsets, ft=None, ivectors=None, timedelay=0, prior=None, combine=AM_COMBINE_LOGITS, normalise=True):
        """x: [N, D] frames of whole utterances delimited by ``offsets``.  Returns float32 [N, C] in file order."""
        ens = as_ensemble(model)
        x = np.ascontiguousarray(x, dtype=np.float32)
        offsets = np.asarray(offsets, dtype=np.int64)
        n, D = x.shape
        iv = None if ivectors is None else np.ascontiguousarray(ivectors, dtype=np.float32)
        d_iv = 0 if iv is None else iv.shape[1]
        y = self._out(n, ens.num_classes)   # every row belongs to an utterance of some bin: all of it is overwritten
        bins = lstm_bins(offsets, len(self.shards))
        jobs = []
        for utts, shard in zip(bins, self.shards):
            def job(utts=utts, shard=shard):
                st = self._dev(shard)
                p = self._configure(st, ens, D, [0], ft, d_iv, prior, combine, normalise)
                if len(bins) == 1:
                    p.forward_lstm(x, offsets, iv, timedelay, out=y)
                    return
                xs, local = gather_utterances(x, offsets, utts)
                ivs = None if iv is None else gather_utterances(iv, offsets, utts)[0]
                ys = p.forward_lstm(xs, local, ivs, timedelay, out=self._out(xs.shape[0], ens.num_classes))
                for j, u in enumerate(utts):
                    y[offsets[u]:offsets[u + 1]] = ys[local[j]:local[j + 1]]
            jobs.append((shard, job))
        if jobs:
            self._run_parallel(jobs)
        return y

    # device-resident variants (single GPU): the posteriors stay in the pipeline's device memory -------------------
    def resident(self, model, x, offsets=None, shifts=(0,), ft=None, ivectors=None, prior=None, recurrent=False, normalise=True):
        """Upload + run without the download: returns the pipeline that now holds the [N, C] posteriors on the GPU
        (``PhoneRecog.recognise_pipeline`` decodes them in place).  One shard only.  In split-f16 mode ``ff_run`` itself
        repeats the pass on the exact f32 path when an activation does not fit f16 (am_ff_run)."""
        if len(self.shards) != 1:
            raise ValueError("device-resident posteriors need a single-GPU runner")
        ens = as_ensemble(model)
        x = np.ascontiguousarray(x, dtype=np.float32)
        iv = None if ivectors is None else np.ascontiguousarray(ivectors, dtype=np.float32)
        d_iv = 0 if iv is None else iv.shape[1]
        st = self._dev(self.shards[0])
        shifts = [0] if recurrent else [int(s) for s in shifts]
        p = self._configure(st, ens, x.shape[1], shifts, ft, d_iv, prior, AM_COMBINE_LOGITS, normalise)
        if recurrent:
            p.lstm_upload(x, iv, np.asarray(offsets, dtype=np.int64), 0)
            p.lstm_run()
        else:
            p.ff_upload(x, iv, x.shape[0], 0, 0)
            p.ff_run()
        return p

    def close(self):
        for st in self._state.values():
            st.pipeline.close()
            for _, m in st.models:
                m.close()
            st.engine.close()
        self._state = {}
        self._pinned = {}


_default_runners = {}


_default_precision = "f32"


def set_default_precision(mode):
    """Precision of runners created by get_runner() from now on (the CLIs' --precision flag)."""
    global _default_precision
    if mode not in ("f32", "split_f16"):
        raise ValueError("precision must be 'f32' or 'split_f16'")
    _default_precision = mode


def get_runner(devices):
    """Process-wide cache of runners keyed by the shard tuple and precision (models stay resident across calls)."""
    key = tuple(tuple(d) if isinstance(d, (tuple, list)) else int(d) for d in devices) + (_default_precision,)
    r = _default_runners.get(key)
    if r is None:
        r = _default_runners[key] = PosteriorRunner(key[:-1], precision=_default_precision)
    return r


def parse_devices(gpu, n_gpus=None):
    """``--gpu G`` (reference: one device id) plus the extension ``--gpus N``: devices G..G+N-1."""
    if gpu < 0:
        raise SystemExit("--gpu %d: CPU execution is not supported by the MI355X engine (no CPU fallback)" % gpu)
    n = 1 if not n_gpus else int(n_gpus)
    return list(range(gpu, gpu + n))

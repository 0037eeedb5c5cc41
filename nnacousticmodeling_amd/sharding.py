"""Shard the posterior computation over the GPUs of one node; every shard delivers its own rows.

Frames (feed-forward) and utterances (recurrent) are independent, so the path shards with no collective
(SURVEY.md 8e): contiguous frame ranges with a +-context read halo for ``ff`` -- the splice clamps only at the global
ends of the file (reference kw_utils.py:29-33), never at shard or utterance boundaries -- and longest-first bins of
whole utterances for ``lstm``.  One host thread drives each logical shard through its own engine (ctypes releases the GIL).

Two ways out of a shard:
* ``forward_ff`` / ``forward_lstm``: the reference's contract, ONE ``[N, C]`` array in file order (host-side gather; every
  shard downloads straight into its slice of one page-locked block);
* ``map_shards``: each shard thread hands its rows to a caller-supplied function while they are still on its GPU and/or in
  its own page-locked block -- the ``.lab`` writer, the in-process decoder and the ``.npy`` writer use this, so N GPUs do not
  funnel 7.6 KB per frame into one array (and ``evaluate --no-lab`` never downloads the posteriors at all).
"""
from __future__ import annotations

import os
import threading
import weakref

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


def ff_shards_aligned(offsets, n_parts, context):
    """Frame ranges cut at UTTERANCE boundaries (each shard owns whole utterances, so it can write their ``.lab`` files and
    decode them by itself); the read halo still crosses the boundary, as the reference's splice does.
    -> [(lo, hi, halo_lo, halo_hi, u_lo, u_hi), ...]: frames [lo, hi) = utterances [u_lo, u_hi).  Empty shards are dropped."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n_utts, n = len(offsets) - 1, int(offsets[-1])
    n_parts = max(1, min(int(n_parts), max(n_utts, 1)))
    cuts = [0]
    for g in range(1, n_parts):
        u = int(np.searchsorted(offsets, n * g // n_parts, side="left"))   # first boundary at or after the even cut
        cuts.append(min(max(u, cuts[-1]), n_utts))
    cuts.append(n_utts)
    out = []
    for u_lo, u_hi in zip(cuts[:-1], cuts[1:]):
        lo, hi = int(offsets[u_lo]), int(offsets[u_hi])
        if u_hi > u_lo:
            out.append((lo, hi, min(context, lo), min(context, n - hi), u_lo, u_hi))
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


def check_offsets(offsets, n):
    """The utterance table must tile the file: start at 0, never decrease, end at the frame count (the reference checks the
    last entry, evaluateModelForTest.py:39; rows no utterance covers would otherwise come back uninitialised)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    if offsets.ndim != 1 or len(offsets) < 1 or offsets[0] != 0 or offsets[-1] != n or np.any(np.diff(offsets) < 0):
        raise ValueError("offsets must start at 0, be non-decreasing and end at the number of frames (%d)" % n)
    return offsets


# ---------------------------------------------------------------------------------------------------
# page-locked result blocks
# ---------------------------------------------------------------------------------------------------
def _default_pool_bytes():
    v = os.environ.get("AM_PINNED_POOL_MB")
    if v:
        return int(float(v) * (1 << 20))
    try:
        ram = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES")
    except (ValueError, OSError):
        ram = 64 << 30
    return int(min(16 << 30, ram // 4))


class PinnedPool:
    """Page-locked host blocks for posteriors, handed out as explicit LEASES.

    ``take(n, c)`` returns an ``[n, c]`` float32 array on page-locked memory (a view of a block with at least ``n * c``
    elements) -- a block somebody gave back when one is large enough, else a fresh one.  The array belongs to the caller: it
    is never handed out again unless the caller returns it with ``give(y)`` (after which ``y`` must not be used).  Blocks are
    matched by capacity, not by shape, so a corpus of differently sized files re-uses a few blocks.  The pool's page-locked
    bytes (given-back blocks + outstanding leases) are capped: over the cap the least recently returned blocks are freed
    first, and a request that still does not fit gets ordinary pageable memory (slower download, never an error)."""

    def __init__(self, alloc, cap_bytes=None):
        self._alloc = alloc                      # (shape, dtype) -> ndarray on page-locked memory
        self.cap = _default_pool_bytes() if cap_bytes is None else int(cap_bytes)
        self._free = []                          # blocks given back, least recently returned first
        self._leased = {}                        # id(block) -> (weakref to the block, bytes)
        self._dead = []                          # (id, weakref) of leases whose array was dropped without give(): reaped under the lock
        self._dropped = False                    # a lease has been dropped (not given back) since the caller's last give()
        self._pending = 0                        # bytes reserved by take() calls that are page-locking their block right now
        self._lock = threading.Lock()

    @staticmethod
    def _root(y):
        while isinstance(getattr(y, "base", None), np.ndarray):
            y = y.base
        return y

    def _reap(self):
        """(lock held)  Forget leases whose arrays have been garbage-collected: their blocks are already unpinned.  The weakref
        callback only queues -- it may run inside any allocation, also while this thread holds the lock."""
        while self._dead:
            key, ref = self._dead.pop()
            ent = self._leased.get(key)
            if ent is not None and ent[0] is ref:
                del self._leased[key]
                self._dropped = True

    def leases_get_dropped(self):
        """True when the caller's pattern is take -> drop (the reference's ``predict`` contract has no release): the block of a
        dropped lease is unpinned with its array, so the next request would page-lock a fresh one (~0.2 ms per MB, 0.1 s for the
        core-test file's 442 MB) -- more than the page-locked download saves."""
        with self._lock:
            self._reap()
            return self._dropped

    def _bytes(self):
        return sum(b.nbytes for b in self._free) + sum(nb for _, nb in self._leased.values()) + self._pending

    def pinned_bytes(self):
        with self._lock:
            self._reap()
            return self._bytes()

    def free_blocks(self):
        with self._lock:
            return len(self._free)

    def take(self, n, c):
        need = int(n) * int(c)
        if need == 0 or self._alloc is None:
            return np.empty((n, c), dtype=np.float32)
        with self._lock:
            self._reap()
            best = None
            for i, b in enumerate(self._free):
                if b.size >= need and (best is None or b.size < self._free[best].size):
                    best = i
            if best is not None:
                block = self._free.pop(best)
                reserved = block.nbytes              # (out of _free, not yet in _leased: still counted)
            else:
                used = self._bytes()
                while self._free and used + 4 * need > self.cap:
                    used -= self._free.pop(0).nbytes            # least recently returned first
                if used + 4 * need > self.cap:
                    return np.empty((n, c), dtype=np.float32)   # over the cap: pageable memory
                block = None
                reserved = 4 * need                  # N shard threads asking at once each see the others' reservations
            self._pending += reserved
        if block is None:
            try:
                block = self._alloc((need,), np.float32)        # (outside the lock: page-locking takes ~0.2 ms per MB)
            except Exception:                                   # the host would not page-lock more: pageable, never an error in a shard thread
                with self._lock:
                    self._pending -= reserved
                return np.empty((n, c), dtype=np.float32)
        key = id(block)
        ref = weakref.ref(block, lambda r, k=key: self._dead.append((k, r)))
        with self._lock:
            self._pending -= reserved
            self._leased[key] = (ref, block.nbytes)
        return block[:need].reshape(n, c)

    def give(self, y, from_caller=False):
        """Return a leased array's block to the pool.  Arrays that are not leases of this pool (pageable ones, slices of
        somebody else's memory) are ignored.  The caller must not touch ``y`` (or views of it) afterwards.
        ``from_caller``: the array is a RESULT the runner's caller hands back (``PosteriorRunner.release``) -- only that says the
        caller's pattern is take -> give and switches the pool back on after dropped leases; the runner's own per-shard
        temporaries come back through here too and must not."""
        block = self._root(y)
        with self._lock:
            self._reap()
            ent = self._leased.get(id(block))
            if ent is None or ent[0]() is not block:
                return False
            del self._leased[id(block)]
            if from_caller:
                self._dropped = False
            self._free.append(block)
            total = self._bytes()
            while len(self._free) > 1 and total > self.cap:
                total -= self._free.pop(0).nbytes
        return True

    def clear(self):
        with self._lock:
            self._free = []
            self._leased = {}
            self._dead = []


# ---------------------------------------------------------------------------------------------------
# runner
# ---------------------------------------------------------------------------------------------------
class _ShardState:
    """One logical shard: its own engine (streams, workspace), pipeline and resident models on one GPU."""

    def __init__(self, backend, device):
        self.engine = backend.Engine(device)
        self.models = []          # [(spec, Model)]: holds the spec itself, never a bare id() (ids are reused after GC)
        self.pipeline = backend.Pipeline(self.engine)
        self.decoders = {}        # key -> decoder bound to this shard's engine (ShardContext.decoder)

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


class ShardContext:
    """What ``map_shards`` hands to the per-shard function, in the shard's own thread, after its pass has run:

    ``index`` / ``shard``   position in the runner's shard list / the (gpu, replica) pair
    ``engine``, ``pipeline`` the shard's engine and its pipeline, which holds this shard's posteriors ON THE DEVICE
    ``utts``                 indices (file order) of the utterances this shard owns, or None (feed-forward without offsets)
    ``local_offsets``        their offsets inside this shard's rows (``len(utts) + 1`` entries), or None
    ``lo``, ``hi``           feed-forward: the shard's frame range in the file; recurrent: None
    ``n``                    rows of this shard
    ``download()``           -> float32 ``[n, C]`` on the host (page-locked lease of the runner's pool; ``release`` gives it back)
    ``decoder(key, make)``   a per-engine cache for objects bound to this shard's GPU (the Viterbi decoder)."""

    def __init__(self, runner, index, shard, st, recurrent, utts, local_offsets, lo, hi, n):
        self._runner, self._st, self._recurrent = runner, st, recurrent
        self.index, self.shard = index, shard
        self.engine, self.pipeline = st.engine, st.pipeline
        self.utts, self.local_offsets, self.lo, self.hi, self.n = utts, local_offsets, lo, hi, n
        self._host = None

    def download(self):
        if self._host is None:
            y = self._runner._out(self.n, self.pipeline.num_classes)
            self._host = self.pipeline.lstm_download(out=y) if self._recurrent else self.pipeline.ff_download(out=y)
        return self._host

    def release(self):
        if self._host is not None:
            self._runner.pool.give(self._host)
            self._host = None

    def decoder(self, key, make):
        d = self._st.decoders.get(key)
        if d is None:
            d = self._st.decoders[key] = make(self.engine)
        return d


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
    stand-in to exercise this host logic without a GPU).
    Result arrays of ``forward_ff`` / ``forward_lstm`` are the caller's; ``release(y)`` offers a page-locked one back to the
    pool for the next call (repeated workloads: page-locking 442 MB costs ~0.1 s, more than a warm pass)."""

    def __init__(self, devices=(0,), backend=None, precision="f32", pool_bytes=None):
        if backend is None:
            from . import engine as backend
        self.backend = backend
        self.precision = precision   # "f32" (exact f32 MFMA) or "split_f16" (f32-accurate on the f16 pipe, feed-forward only)
        self.shards = _normalise_shards(devices)
        self.devices = [g for g, _ in self.shards]
        self._state = {}
        self.pool = PinnedPool(getattr(backend, "pinned_empty", None), pool_bytes)
        self._requests = 0
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
        """The gathered [N, C] array of forward_ff / forward_lstm.  Page-locked (every shard's download then runs at PCIe
        rate straight into its slice instead of through the driver's bounce buffers) -- except for the very first request
        of a single-shard runner: a one-shot call is faster into pageable memory (43 ms for the core-test file) than into
        memory it first has to page-lock (~100 ms); from the second request on the pool is used -- as long as the caller gives
        its results back (``release``); one that drops them instead gets pageable memory again."""
        with self._lock:
            self._requests += 1
            first = self._requests == 1
        if (len(self.shards) == 1 and first) or self.pool.leases_get_dropped():
            return np.empty((n, c), dtype=np.float32)   # (a caller that never releases: pageable every time beats page-locking every time, on any number of shards)
        return self.pool.take(n, c)

    def release(self, y):
        """Offer a result array back to the pool (no-op for pageable arrays).  ``y`` must not be used afterwards."""
        return self.pool.give(y, from_caller=True)

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
        n, D = x.shape
        offsets = check_offsets(offsets, n)   # every row then belongs to an utterance of some bin: all of y is overwritten
        iv = None if ivectors is None else np.ascontiguousarray(ivectors, dtype=np.float32)
        d_iv = 0 if iv is None else iv.shape[1]
        bins = lstm_bins(offsets, len(self.shards))
        # several bins: y is only ever written by the host copies below (never by DMA), so page-locking it buys nothing
        y = self._out(n, ens.num_classes) if len(bins) == 1 else np.empty((n, ens.num_classes), dtype=np.float32)
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
                ys = p.forward_lstm(xs, local, ivs, timedelay, out=self.pool.take(xs.shape[0], ens.num_classes))
                for j, u in enumerate(utts):
                    y[offsets[u]:offsets[u + 1]] = ys[local[j]:local[j + 1]]
                self.pool.give(ys)            # a temporary of this call: straight back to the pool
            jobs.append((shard, job))
        if jobs:
            self._run_parallel(jobs)
        return y

    # every shard delivers its own rows ---------------------------------------------------------------
    def map_shards(self, model, x, fn, offsets=None, shifts=(0,), ft=None, ivectors=None, prior=None, recurrent=False, timedelay=0,
                   combine=AM_COMBINE_LOGITS, normalise=True, host=False):
        """Run the pass shard by shard and call ``fn(ctx)`` (``ShardContext``) in each shard's thread once its posteriors are
        on its GPU; -> the list of ``fn``'s return values in shard order.

        Feed-forward with ``offsets``: frame ranges cut at utterance boundaries (``ff_shards_aligned``), so a shard owns whole
        utterances; without: plain even ranges.  Recurrent: longest-first utterance bins.  Nothing is gathered: ``fn`` writes,
        decodes or copies its shard's rows wherever they have to go (``ctx.download()`` for a host copy).  ``host=True`` says the
        host copy will be wanted: feed-forward shards then run the pipelined form (the download of chunk c beside the compute of
        chunk c + 1, am_forward_ff) straight into the shard's block; the posteriors stay resident on the GPU as well."""
        ens = as_ensemble(model)
        x = np.ascontiguousarray(x, dtype=np.float32)
        n, D = x.shape
        iv = None if ivectors is None else np.ascontiguousarray(ivectors, dtype=np.float32)
        d_iv = 0 if iv is None else iv.shape[1]
        if offsets is not None:
            offsets = check_offsets(offsets, n)
        if recurrent and offsets is None:
            raise ValueError("recurrent networks need the utterance offsets")
        results = [None] * len(self.shards)
        jobs = []
        if recurrent:
            bins = lstm_bins(offsets, len(self.shards))
            for i, (utts, shard) in enumerate(zip(bins, self.shards)):
                def job(i=i, utts=utts, shard=shard):
                    st = self._dev(shard)
                    p = self._configure(st, ens, D, [0], ft, d_iv, prior, combine, normalise)
                    if len(bins) == 1:
                        xs, local, ivs = x, offsets, iv
                    else:
                        xs, local = gather_utterances(x, offsets, utts)
                        ivs = None if iv is None else gather_utterances(iv, offsets, utts)[0]
                    p.lstm_upload(xs, ivs, local, timedelay)
                    p.lstm_run()
                    ctx = ShardContext(self, i, shard, st, True, utts, local, None, None, xs.shape[0])
                    try:
                        results[i] = fn(ctx)
                    finally:
                        ctx.release()
                jobs.append((shard, job))
            results = results[:len(bins)]
        else:
            sh = [int(s) for s in shifts]
            ctxw = max(abs(s) for s in sh)
            if offsets is not None:
                ranges = ff_shards_aligned(offsets, len(self.shards), ctxw)
            else:
                ranges = [r + (None, None) for r in ff_shards(n, len(self.shards), ctxw)]
            for i, ((lo, hi, h_lo, h_hi, u_lo, u_hi), shard) in enumerate(zip(ranges, self.shards)):
                def job(i=i, lo=lo, hi=hi, h_lo=h_lo, h_hi=h_hi, u_lo=u_lo, u_hi=u_hi, shard=shard):
                    st = self._dev(shard)
                    p = self._configure(st, ens, D, sh, ft, d_iv, prior, combine, normalise)
                    utts = None if u_lo is None else np.arange(u_lo, u_hi, dtype=np.int64)
                    local = None if u_lo is None else offsets[u_lo:u_hi + 1] - lo
                    ctx = ShardContext(self, i, shard, st, False, utts, local, lo, hi, hi - lo)
                    if host:
                        ctx._host = p.forward_ff(x[lo - h_lo:hi + h_hi], None if iv is None else iv[lo:hi], hi - lo, h_lo, h_hi,
                                                 out=self._out(hi - lo, ens.num_classes))
                    else:
                        p.ff_upload(x[lo - h_lo:hi + h_hi], None if iv is None else iv[lo:hi], hi - lo, h_lo, h_hi)
                        p.ff_run()
                    try:
                        results[i] = fn(ctx)
                    finally:
                        ctx.release()
                jobs.append((shard, job))
            results = results[:len(ranges)]
        if jobs:
            self._run_parallel(jobs)
        return results

    def close(self):
        for st in self._state.values():
            for d in st.decoders.values():
                if hasattr(d, "close"):
                    d.close()
            st.pipeline.close()
            for _, m in st.models:
                m.close()
            st.engine.close()
        self._state = {}
        self.pool.clear()


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

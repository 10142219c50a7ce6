"""HIP-graph replay of the fused training step.

The fused step is ~2,600 kernel launches of 10-300 us each; issued one by one from Python it is only as fast as
the host can issue them, and on a slow or busy host the GPU idles between kernels.  The step's launch sequence
is static for given batch shapes, so it is captured once into HIP graphs (torch.cuda.CUDAGraph is a thin wrapper
over hipGraph) and replayed:

  forward  : one graph  (salt bump, text encoder with the table/image encoders on a parallel branch, leave-one-out
             decoder, LM head + loss)
  backward : one graph per gradient segment (modules._segment_layers: three decoder layers at a time, then three encoder
             layers at a time -- the first of those with the image/table backward on a parallel branch, the last with the
             embeddings), so the data-parallel exchange of a finished segment (parallel.DistributedDataParallel) overlaps
             the next segment's kernels and only the last ~0.36 GB of gradients (the encoder's bottom layers + the tied
             embedding) is reduced in the open -- the collectives stay outside the graphs, on their own stream.

What stays eager: weight shadow refresh (engine.sync_weights), gradient-buffer preparation, clipping and the
optimiser (a handful of launches whose scalars -- lr, bias corrections -- change every step).

ONE set of graphs serves every batch of a given SHAPE.  What varies from batch to batch inside a shape -- how many
review tokens are real, how many image slots are filled -- only changes how many rows the padding-free parts of the
step work on, and those counts are device-side scalars (engine.row_maps -> the kernels' `live_rows` argument): the
graph is captured for the row capacity (all rows) and every kernel reads the live count when it runs.  So a real
loader with varying token counts never triggers a re-capture, and graph memory is one set of activations.  A second
shape (e.g. the last, smaller batch of an epoch when drop_last is off) gets its own set; at most `max_live` sets are
kept, the least recently used one is dropped first.

Dropout: a captured graph replays the seed arguments it was captured with; the kernels mix in a device-resident
salt (their `salt` argument, engine.salt) that the forward graph bumps first, so every replay draws fresh masks and
the backward graphs of the same step see the same ones.

The upstream gradient of the loss (modules._StepFn.backward) is a device scalar the captured LM-head backward
products multiply by; it is copied into a static buffer before the backward graphs replay.

The first call with new shapes runs eagerly (warm-up: allocator, kernel attributes), the second captures, later ones
replay.
"""
import torch

from . import kernels as kn
from ._lib import check, lib


def _flatten(batch):
    flat, spec = [], []
    for x in batch:
        if isinstance(x, (list, tuple)):
            spec.append(len(x))
            flat.extend(x)
        else:
            spec.append(None)
            flat.append(x)
    return flat, spec


def _unflatten(flat, spec):
    out, i = [], 0
    for n in spec:
        if n is None:
            out.append(flat[i])
            i += 1
        else:
            out.append(list(flat[i:i + n]))
            i += n
    return tuple(out)


class _Entry:
    __slots__ = ("state", "static", "fwd", "bwd", "saved", "serial", "upstream")

    def __init__(self):
        self.state, self.static, self.fwd, self.bwd, self.saved, self.serial, self.upstream = 0, None, None, [], None, 0, None


class StepGraphs:
    """Owned by a step module (MultimodalSum / TextSupervised); used by modules._StepFn."""

    def __init__(self, model, max_live=2):
        self.max_live = max(1, int(max_live))     # captured graph sets kept at once: each pins its own activations in HBM
        self.model = model
        self.engine = model._engine
        self.entries = {}                         # insertion order = least recently used first
        self.pool = None
        self.captures = 0                         # how many times a set was captured (bench.py reports it)
        self.salt = torch.zeros(1, dtype=torch.int64, device=self.engine.device)

    def _key(self, flat):
        e = self.engine
        from .modules import _compact
        return tuple((tuple(t.shape), t.dtype) for t in flat) + (e.training, e.p_drop(), _compact(self.model))

    def forward(self, batch):
        """Returns the entry whose .saved holds this step's forward state, or None (caller runs eagerly)."""
        flat, spec = _flatten(batch)
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
            return None
        key = self._key(flat)
        ent = self.entries.get(key)
        if ent is None:
            self.entries[key] = _Entry()
            self._evict(keep=key)
            return None                          # first sight of these shapes: eager warm-up
        if ent.state == -1:
            return None                          # capture failed for these shapes before: stay eager
        self.entries[key] = self.entries.pop(key)          # most recently used last
        if ent.state == 0:
            self._evict(keep=key)
            try:
                self._capture(ent, flat, spec)
            except Exception as exc:             # a failed capture must not take the training run down: fall back to eager launches
                import warnings
                warnings.warn("multimodalsum_amd: HIP-graph capture of the step failed (%r); continuing with eager launches" % (exc,))
                ent.state, ent.fwd, ent.bwd, ent.saved = -1, None, [], None
                try:
                    torch.cuda.synchronize()
                except Exception:
                    pass
                return None
        else:
            for dst, src in zip(ent.static, flat):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
        ent.fwd.replay()
        ent.serial += 1
        return ent

    def _evict(self, keep):
        """Drop least recently used entries until at most max_live remain (the one being served included)."""
        victims = [k for k in self.entries if k != keep]
        dropped = False
        while len(self.entries) > self.max_live and victims:
            old = self.entries.pop(victims.pop(0))
            dropped = dropped or old.state == 1
            old.state, old.fwd, old.bwd, old.saved, old.static = 0, None, [], None, None
        if dropped:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()              # hand the dropped set's pool blocks back before the next capture sizes its own
        if not any(en.state == 1 for en in self.entries.values()):
            self.pool = None                      # the allocator drops a graph pool with its last graph: start a new one

    def _capture(self, ent, flat, spec):
        e, m = self.engine, self.model
        from .modules import _compact
        ent.static = [t.clone() for t in flat]
        ent.upstream = torch.ones(1, dtype=torch.float32, device=e.device)
        torch.cuda.synchronize()
        e.salt = self.salt                       # the dropout kernels captured below mix this device counter into their seeds
        try:
            ent.fwd = torch.cuda.CUDAGraph()
            # thread_local: other threads keep their HIP calls while this one captures (the RCCL process group's watchdog thread
            # polls its events every few hundred milliseconds; under the default global mode that poll fails the capture -- and
            # kills the watchdog -- once capturing takes longer than its period)
            with torch.cuda.graph(ent.fwd, pool=self.pool, capture_error_mode="thread_local"):
                check(lib.mmsum_bump_u64(self.salt.data_ptr(), 1, kn._stream()), "mmsum_bump_u64")
                ent.saved = m._step_fwd(*_unflatten(ent.static, spec), compact=_compact(m))
            if self.pool is None:
                self.pool = ent.fwd.pool()
            # the backward graphs are captured here too (capture records, it does not execute): autograd would
            # otherwise run the capture on its worker thread
            keep_touched = e.touched
            for fn, prefixes in m._step_bwd_segments(ent.saved, release=False, upstream=ent.upstream):
                e.touched = set()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
                    fn()
                ent.bwd.append((g, frozenset(e.touched), prefixes))
            e.touched = keep_touched
        finally:
            e.salt = None
        self.captures += 1
        ent.state = 1

    def backward(self, ent, begin_backward, end_backward, serial=None, upstream=None):
        e = self.engine
        if ent.state != 1 or (serial is not None and serial != ent.serial):
            # the set's activation buffers belong to its latest forward replay; the training loop this path serves
            # (multimodal_train.py:355-373) always runs backward before the next forward
            raise RuntimeError("multimodalsum_amd: backward of a step whose captured forward state was overwritten by a later "
                               "forward (or evicted); call backward before the next forward, or enable_step_graphs(False)")
        if upstream is not None:
            ent.upstream.copy_(upstream, non_blocking=True)
        else:
            ent.upstream.fill_(1.0)
        begin_backward(e)
        for g, touched, prefixes in ent.bwd:
            g.replay()
            e.touched = set(touched)
            end_backward(e)
            e.segment_ready(prefixes)

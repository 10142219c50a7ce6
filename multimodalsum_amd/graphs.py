"""HIP-graph replay of the fused training step.

The fused step is ~2,600 kernel launches of 10-300 us each; issued one by one from Python it is only as fast as
the host can issue them, and on a slow or busy host the GPU idles between kernels.  The step's launch sequence
is static for given batch shapes, so it is captured once into HIP graphs (torch.cuda.CUDAGraph is a thin wrapper
over hipGraph) and replayed:

  forward  : one graph  (salt bump, text encoder with the table/image encoders on a parallel branch, leave-one-out
             decoder, LM head + loss)
  backward : one graph per gradient segment (decoder | upper half of the text encoder with the image/table backward on
             a parallel branch | lower half + embeddings), so the data-parallel all-reduce of a finished segment
             (parallel.DistributedDataParallel) overlaps the next segment's kernels and only the last ~0.4 GB of
             gradients is reduced in the open -- the collectives stay outside the graphs, on their own stream.

What stays eager: weight shadow refresh (engine.sync_weights), gradient-buffer preparation, clipping and the
optimiser (a handful of launches whose scalars -- lr, bias corrections -- change every step).

Dropout: a captured graph replays the seed arguments it was captured with; the kernels mix in a device-resident
salt that the forward graph bumps first (include/mmsum_hip.h: mmsum_set_dropout_salt), so every replay draws
fresh masks and the backward graphs of the same step see the same ones.

Shapes: one set of graphs per distinct (input shapes, dtypes, training flag, row capacity of the padding-free encoder); the first call with new shapes runs
eagerly (warm-up), the second captures, later ones replay.
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
    __slots__ = ("state", "static", "fwd", "bwd", "saved", "serial", "nbytes")

    def __init__(self):
        self.state, self.static, self.fwd, self.bwd, self.saved, self.serial, self.nbytes = 0, None, None, [], None, 0, 0


class StepGraphs:
    """Owned by a step module (MultimodalSum / TextSupervised); used by modules._StepFn."""

    def __init__(self, model, max_shapes=8, max_live=4):
        self.max_live = max_live                  # captured graph sets kept at once: each pins its own activations in HBM
        self.model = model
        self.engine = model._engine
        self.entries = {}
        self.pool = None
        self.max_shapes = max_shapes
        self.salt = torch.zeros(1, dtype=torch.int64, device=self.engine.device)

    def _key(self, flat, extra):
        e = self.engine
        return tuple((tuple(t.shape), t.dtype) for t in flat) + (e.training, e.p_drop(), extra)

    def forward(self, batch, extra=None):
        """Returns the entry whose .saved holds this step's forward state, or None (caller runs eagerly)."""
        flat, spec = _flatten(batch)
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in flat):
            return None
        key = self._key(flat, extra)
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) >= self.max_shapes:
                return None                      # too many distinct shapes: stay eager rather than hoard graph memory
            self.entries[key] = _Entry()
            self._make_room(for_capture=False)   # the eager warm-up needs as much memory as a captured set pins
            return None                          # first sight of these shapes: eager warm-up
        if ent.state == -1:
            return None                          # capture failed for these shapes before: stay eager
        if ent.state == 0:
            self._make_room(for_capture=True)
            try:
                self._capture(ent, flat, spec, extra)
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
                    dst.copy_(src)
        self.entries[key] = self.entries.pop(key)          # most recently used last
        ent.fwd.replay()
        ent.serial += 1
        return ent

    def _make_room(self, for_capture):
        """Evict least recently used graph sets (dicts keep insertion order; replays re-insert): before a capture when max_live
        are captured already, and -- capture or eager warm-up of new shapes -- while the device lacks room for another set the
        size of the largest one captured so far."""
        live = [k for k, en in self.entries.items() if en.state == 1]
        # a captured set needs what the largest one so far took from the pool; an eager step of the same shapes holds more
        # (every activation until its backward kernel ran, plus the allocator's rounding): measured ~1.5x
        need = max([self.entries[k].nbytes for k in self.entries], default=0) * (1.0 if for_capture else 1.4)
        if live and self._free_bytes() < 1.15 * need:
            torch.cuda.empty_cache()              # blocks the allocator merely caches count as free: return them, then measure again
        while live and ((for_capture and len(live) >= self.max_live) or self._free_bytes() < 1.15 * need):
            old = self.entries[live.pop(0)]
            old.state, old.fwd, old.bwd, old.saved, old.static = 0, None, [], None, None
            if self._free_bytes() < 1.15 * need:
                torch.cuda.synchronize()
                torch.cuda.empty_cache()          # hand the evicted set's pool back before the new one is sized
        if not live:
            self.pool = None                      # the allocator drops a graph pool with its last graph: start a new one

    def _pool_bytes(self):
        """Bytes the allocator holds in this object's graph pool (segments tagged with the pool id)."""
        if self.pool is None:
            return 0
        try:
            return sum(seg["total_size"] for seg in torch.cuda.memory_snapshot() if tuple(seg.get("segment_pool_id", (0, 0))) == tuple(self.pool))
        except Exception:
            return 0

    def _free_bytes(self):
        """Device memory the allocator could still obtain (blocks cached inside graph pools are not counted: conservative)."""
        return torch.cuda.mem_get_info(self.engine.device)[0]

    def _capture(self, ent, flat, spec, extra):
        e, m = self.engine, self.model
        before = self._pool_bytes()
        ent.static = [t.clone() for t in flat]
        torch.cuda.synchronize()
        check(lib.mmsum_set_dropout_salt(self.salt.data_ptr()), "mmsum_set_dropout_salt")
        try:
            ent.fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ent.fwd, pool=self.pool):
                check(lib.mmsum_bump_u64(self.salt.data_ptr(), 1, kn._stream()), "mmsum_bump_u64")
                ent.saved = m._step_fwd(*_unflatten(ent.static, spec), capacity=extra)
            if self.pool is None:
                self.pool = ent.fwd.pool()
            # the backward graphs are captured here too (capture records, it does not execute): autograd would
            # otherwise run the capture on its worker thread
            keep_touched = e.touched
            for fn, prefixes in m._step_bwd_segments(ent.saved, release=False):
                e.touched = set()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=self.pool):
                    fn()
                ent.bwd.append((g, frozenset(e.touched), prefixes))
            e.touched = keep_touched
        finally:
            check(lib.mmsum_set_dropout_salt(None), "mmsum_set_dropout_salt")
        ent.nbytes = max(0, self._pool_bytes() - before)                         # what this set added to the graph pool
        ent.state = 1

    def backward(self, ent, begin_backward, end_backward, serial=None):
        e = self.engine
        if ent.state != 1 or (serial is not None and serial != ent.serial):
            # the set's activation buffers belong to its latest forward replay; the training loop this path serves
            # (multimodal_train.py:355-373) always runs backward before the next forward
            raise RuntimeError("multimodalsum_amd: backward of a step whose captured forward state was overwritten by a later "
                               "forward (or evicted); call backward before the next forward, or enable_step_graphs(False)")
        begin_backward(e)
        for g, touched, prefixes in ent.bwd:
            g.replay()
            e.touched = set(touched)
            end_backward(e)
            e.segment_ready(prefixes)

"""Data-parallel gradient synchronisation over the flat gradient arena (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

Replaces apex.parallel.DistributedDataParallel(model, delay_allreduce=True)
(/root/reference/src/multimodal_train.py:12,473-474) and reduce_tensor (/root/reference/src/utils.py:8-12).

The reference flattens every gradient into one buffer AFTER backward and all-reduces it with no
overlap.  Here the gradients already live in one arena, so there is nothing to flatten, and the
fused step tells us when a parameter segment is final (modules._segment_layers: three decoder /
encoder layers at a time, the image + table encoders with the encoder's top layers, the tied
embedding with its bottom ones): its arena ranges are exchanged on a side stream in large buckets
while the next segment's backward keeps the compute stream busy, so only the last segment (the
encoder's bottom layers + the 206 MB tied embedding, which the encoder's embedding backward is the
last to touch) is exchanged in the open.  Parameters that never receive a gradient (ResNet
stem/layer1/layer2/layer4/fc) are simply absent from the ranges.

`mode`: "all_reduce" (one collective per bucket) or "reduce_scatter" (reduce-scatter of the bucket into
this rank's 1/N shard, then all-gather of the shards back: the two halves of a ring all-reduce as
separate collectives -- xGMI is point-to-point, 7 links per GPU, and the two phases of consecutive
buckets overlap on the links; it is also the hook for a sharded optimiser step between the phases).
Both give every rank the same mean.

The mean is taken by the collective itself (ReduceOp.AVG on RCCL: no extra pass over the 1.95 GB of
gradients); backends without AVG (gloo) get SUM followed by one scaling pass.  `grad_dtype=torch.bfloat16`
sends bf16 buckets (half the bytes on the xGMI links; the f32 arena stays the accumulator) -- off by
default because the reference reduces in f32.
"""
import collections

import torch
import torch.distributed as dist
import torch.nn as nn

from .optim import _ranges


def reduce_tensor(tensor, world_size):
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= world_size
    return rt


class DistributedDataParallel(nn.Module):
    def __init__(self, module, delay_allreduce=True, bucket_elems=64 * 1024 * 1024, overlap=True, process_group=None,
                 always_reduce=False, grad_dtype=None, collect_stats=False, mode="all_reduce", stats_window=64):
        super().__init__()
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError("mode must be 'all_reduce' or 'reduce_scatter'")
        self.mode = mode
        self.module = module
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_elems)
        self.engine = module._engine
        self.arena = self.engine.arena
        self.overlap = overlap and self.arena.grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.overlap else None
        self.grad_dtype = grad_dtype
        # ReduceOp.AVG exists on RCCL only -- and is asked for only when there is something to average: RCCL 2.26.6's ONE-rank AVG path
        # (a copy kernel with a pre-multiplier) leaves the last 16 bytes of reduce_scatter's output unwritten for some sizes (19,584 f32:
        # tools/rccl_one_rank_avg_probe.py on a 1-GPU box; SUM and every size tried at AVG in place are fine).  A one-rank mean is the sum.
        self.native_avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl" and self.world_size > 1
        self.collect_stats = bool(collect_stats) and self.arena.grad.is_cuda
        self.stats = collections.deque(maxlen=max(1, int(stats_window)))     # the last steps only: (bucket events, bytes, end events); comm_stats() reads them
        self._events, self._bytes = [], 0
        self._done = set()
        # persistent staging (collectives of one rank run in order on ONE stream, so one buffer serves every bucket): the bucket in
        # the wire dtype when that is not the arena's f32, and this rank's shard between the two phases of reduce_scatter mode
        self._wire = None
        if grad_dtype is not None and grad_dtype != self.arena.grad.dtype:
            self._wire = torch.empty(self.bucket_elems, dtype=grad_dtype, device=self.arena.grad.device)
        self._shard = None
        if mode == "reduce_scatter":
            self._shard = torch.empty(-(-self.bucket_elems // max(1, self.world_size)), dtype=grad_dtype or self.arena.grad.dtype,
                                      device=self.arena.grad.device)
        if self.world_size > 1 or (always_reduce and dist.is_initialized()):      # always_reduce: exercise the path at world size 1
            dist.broadcast(self.arena.data, 0, group=self.group)          # C2: parameters from rank 0
            for b in self.engine.buffers.values():
                if b.is_floating_point():
                    dist.broadcast(b, 0, group=self.group)
            self.engine.mark_weights_changed()
            self.engine.segment_hooks.append(self._segment_ready)
            self.engine.post_backward_hooks.append(self._finish)

    def forward(self, *args, **kwargs):
        self._done = set()
        self._events, self._bytes = [], 0
        return self.module(*args, **kwargs)

    # ---- gradient all-reduce ----------------------------------------------------------------------
    def _all_reduce_mean(self, chunk):
        buf = chunk
        if self._wire is not None:
            buf = self._wire[:chunk.numel()]
            buf.copy_(chunk)                     # f32 -> wire dtype into the persistent staging buffer (no allocation per bucket)
        op = dist.ReduceOp.AVG if self.native_avg else dist.ReduceOp.SUM
        W = self.world_size
        n = (buf.numel() // W) * W if self.mode == "reduce_scatter" else 0
        if n:
            # the bucket as W equal shards: this rank reduces shard `rank` into the persistent shard buffer, then every rank gathers
            # all of them back into the bucket.  (An in-place all-gather whose input aliases its own slice of the output is legal on
            # RCCL but not on every backend this class runs on -- gloo in the CPU tests --, and the copy it saves is 1/W of a bucket.)
            head = buf[:n]
            shard = self._shard[:n // W]
            dist.reduce_scatter_tensor(shard, head, op=op, group=self.group)
            dist.all_gather_into_tensor(head, shard, group=self.group)
        if n < buf.numel():                       # all_reduce mode, or the few elements a bucket has beyond a multiple of W
            dist.all_reduce(buf[n:], op=op, group=self.group)
        if not self.native_avg and W > 1:
            buf.mul_(1.0 / W)
        if buf is not chunk:
            chunk.copy_(buf)
        return buf.numel() * buf.element_size()

    def _reduce_ranges(self, ranges):
        g = self.arena.grad
        for s, e in ranges:
            for b0 in range(s, e, self.bucket_elems):
                chunk = g[b0:min(e, b0 + self.bucket_elems)]
                if self.collect_stats:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    self._bytes += self._all_reduce_mean(chunk)
                    e1.record()
                    self._events.append((e0, e1))
                else:
                    self._all_reduce_mean(chunk)

    def _segment_ready(self, prefixes):
        params = [p for n, p in self.arena.params.items() if p.grad is not None and n not in self._done
                  and any(n.startswith(px) for px in prefixes)]
        if not params:
            return
        self._done.update(p._mmsum_name for p in params)
        ranges = _ranges(self.arena, params)
        if self.overlap:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self._reduce_ranges(ranges)
        else:
            self._reduce_ranges(ranges)

    def _finish(self):
        """End of the whole backward pass: reduce whatever no segment notification covered (the
        coarse, un-fused module path) and make the compute stream wait for the side stream."""
        rest = [p for n, p in self.arena.params.items() if p.grad is not None and n not in self._done]
        if rest:
            self._done.update(p._mmsum_name for p in rest)
            if self.overlap:          # on the communication stream like every other bucket: the staging buffers are ordered by that ONE stream
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    self._reduce_ranges(_ranges(self.arena, rest))
            else:
                self._reduce_ranges(_ranges(self.arena, rest))
        if self.overlap:
            if self.collect_stats:
                bwd_done = torch.cuda.Event(enable_timing=True)
                bwd_done.record()                                  # compute stream: the backward kernels end here
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            if self.collect_stats:
                all_done = torch.cuda.Event(enable_timing=True)
                all_done.record()                                  # ... and here the last collective has landed too
                self.stats.append((self._events, self._bytes, bwd_done, all_done))
        elif self.collect_stats:
            self.stats.append((self._events, self._bytes, None, None))

    def comm_stats(self, skip=0):
        """Averages over the recorded steps (after a device synchronise): milliseconds inside the collectives per step,
        milliseconds of them left exposed after the backward's last kernel, bytes reduced per step, and the bus bandwidth
        2 (N-1)/N * bytes / time that a ring all-reduce's links saw."""
        steps = list(self.stats)[skip:]
        if not steps:
            return None
        comm = sum(sum(a.elapsed_time(b) for a, b in ev) for ev, _, _, _ in steps) / len(steps)
        exposed = sum((bd.elapsed_time(ad) if bd is not None else 0.0) for _, _, bd, ad in steps) / len(steps)
        nbytes = sum(nb for _, nb, _, _ in steps) / len(steps)
        n = self.world_size
        bus = (2.0 * (n - 1) / n * nbytes / (comm * 1e-3) / 1e9) if (comm > 0 and n > 1) else 0.0
        return {"allreduce_ms": comm, "exposed_ms": exposed, "overlap_frac": (1.0 - exposed / comm) if comm > 0 else None,
                "bytes_per_step": nbytes, "bus_gb_s": bus, "buckets_per_step": sum(len(ev) for ev, _, _, _ in steps) / len(steps),
                "grad_dtype": str(self.grad_dtype or torch.float32), "mode": self.mode}


def bus_microbench(device, sizes_elems=(64 * 1024 * 1024, 16 * 1024 * 1024), dtypes=(torch.float32, torch.bfloat16), iters=5, group=None):
    """Times the two exchange forms of DistributedDataParallel on its own bucket sizes, outside any training step: all_reduce against
    reduce_scatter + all_gather, f32 and bf16 buckets.  Bus bandwidth = 2 (N-1)/N * bytes / time (what a ring's links carry), to be
    read against the 7 x ~153 GB/s of xGMI links a MI355X has.  Every rank calls it; returns a list of dicts (identical on all ranks
    up to timing).  `python bench.py --gpus N` attaches it to the JSON line as comm.microbench; tools/rccl_bus_bench.py prints it."""
    import time
    W = dist.get_world_size(group)
    rank = dist.get_rank(group)
    cuda = torch.device(device).type == "cuda"
    avg = dist.get_backend(group) == "nccl" and W > 1          # see DistributedDataParallel.native_avg
    op = dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    out = []
    for dt in dtypes:
        for n in sizes_elems:
            n = (n // W) * W
            buf = torch.ones(n, dtype=dt, device=device)
            shard = torch.empty(n // W, dtype=dt, device=device)

            def ar():
                dist.all_reduce(buf, op=op, group=group)

            def rs_ag():
                dist.reduce_scatter_tensor(shard, buf, op=op, group=group)
                dist.all_gather_into_tensor(buf, shard, group=group)
            row = {"elements": n, "dtype": str(dt).replace("torch.", ""), "bytes": n * buf.element_size()}
            for name, fn in (("all_reduce", ar), ("reduce_scatter_all_gather", rs_ag)):
                fn()
                if cuda:
                    torch.cuda.synchronize()
                dist.barrier(group=group)
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                if cuda:
                    torch.cuda.synchronize()
                dt_s = torch.tensor([(time.perf_counter() - t0) / iters], dtype=torch.float64, device=device)
                dist.all_reduce(dt_s, op=dist.ReduceOp.MAX, group=group)
                sec = float(dt_s.item())
                row[name + "_ms"] = sec * 1e3
                row[name + "_bus_gb_s"] = (2.0 * (W - 1) / W * row["bytes"] / sec / 1e9) if W > 1 else 0.0
            out.append(row)
    return out

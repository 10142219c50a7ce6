"""Side-stream host->HBM prefetchers.

Drop-in for yelp_data_prefetcher / amazon_data_prefetcher (/root/reference/src/multimodal_train.py:196-268, 271-343)
and data_prefetcher (/root/reference/src/text_pretrain.py:116-150): the next batch is copied to the GPU on a private
HIP stream while the current step computes, `next()` makes the compute stream wait for that copy and returns the batch
regrouped exactly as the reference returns it (table fields as one list), or Nones when the loader is exhausted.

One difference: host tensors that are not page-locked yet are pinned first (`DataLoader(pin_memory=True)` batches
already are), because a pageable source turns `non_blocking=True` into a synchronous staged copy on ROCm and the
overlap is lost.  The fused step replays from HIP graphs whose inputs are static buffers (graphs.StepGraphs copies the
prefetched batch into them on the compute stream), so prefetched tensors can be freed as soon as the step has started.
"""
import torch


class _Prefetcher:
    n_fields = 11
    table = slice(3, 9)        # positions of the six table tensors inside a loader batch

    def __init__(self, loader, device=None):
        self.loader = iter(loader)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.batch = None
        self.preload()

    def _to_device(self, t):
        if not t.is_cuda and not t.is_pinned():
            t = t.pin_memory()
        return t.to(self.device, non_blocking=True)

    def preload(self):
        try:
            host = next(self.loader)
        except StopIteration:
            self.batch = None
            return
        assert len(host) == self.n_fields, "loader batch has %d tensors, expected %d" % (len(host), self.n_fields)
        with torch.cuda.stream(self.stream):
            self.batch = [self._to_device(t) for t in host]

    def _regroup(self, b):
        raise NotImplementedError

    def next(self):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.stream)
        b = self.batch
        if b is None:
            return self._regroup(None)
        for t in b:
            t.record_stream(cur)
        self.preload()
        return self._regroup(b)


class yelp_data_prefetcher(_Prefetcher):
    """batch = (reviews, reviews_mask, reviews_rating, name, category, str_categorical, str_boolean, rating, hours, img, img_mask)
    -> reviews, reviews_mask, reviews_rating, [name, category, str_categorical, str_boolean, rating, hours], img, img_mask."""

    def _regroup(self, b):
        if b is None:
            return None, None, None, [None] * 6, None, None
        return b[0], b[1], b[2], list(b[3:9]), b[9], b[10]


class amazon_data_prefetcher(yelp_data_prefetcher):
    """batch = (reviews, reviews_mask, reviews_rating, price, rating, brand, name, category, description, img, img_mask)
    -> reviews, reviews_mask, reviews_rating, [price, rating, brand, name, category, description], img, img_mask."""


class data_prefetcher(_Prefetcher):
    """text_pretrain.py:116-150: batch = (reviews, reviews_mask, reviews_rating)."""
    n_fields = 3

    def _regroup(self, b):
        return (None, None, None) if b is None else (b[0], b[1], b[2])

"""Host-side statement of the device dropout masks (csrc/mmsum_device.h: dropout_keep / hash_u32, csrc/rowwise.hip: salted_seed /
keep_threshold).

The reference draws its dropout masks from torch's Philox stream (F.dropout at modeling_multimodalsum.py:294,305,371,458,474,486,596);
the HIP kernels use a counter-based hash instead -- the keep decision of element (row, column) of a dropout site is a pure function of
(seed of the site, salt of the replay, row * D + column) -- so that the backward pass regenerates a mask instead of storing it and a
captured graph draws fresh masks on every replay.  The two streams cannot agree, but the masks are reproducible on the host: this module
restates the hash so that a checker can hand the SAME masks to a CPU statement of the reference and compare tensors of a training step
with dropout on (tests/test_dropout_parity_gpu.py).  Nothing on the product path calls it.
"""
import numpy as np
import torch

_M32 = np.uint64(0xFFFFFFFF)
_GOLDEN = 0x9E3779B97F4A7C15


def _hash_u32(x):
    """x: uint64 array holding 32-bit values."""
    x = x ^ (x >> np.uint64(16))
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x = x ^ (x >> np.uint64(15))
    x = (x * np.uint64(0x846CA68B)) & _M32
    x = x ^ (x >> np.uint64(16))
    return x


def keep_threshold(p_drop):
    if p_drop <= 0.0:
        return 0xFFFFFFFF
    t = (1.0 - float(np.float32(p_drop))) * 4294967296.0          # the kernels receive p as a float
    return 0xFFFFFFFF if t >= 4294967295.0 else int(t)


def salted_seed(seed, salt=None):
    return int(seed) if salt is None else (int(seed) + int(salt) * _GOLDEN) & 0xFFFFFFFFFFFFFFFF


def keep_mask(seed, rows, D, p_drop, salt=None):
    """Keep decisions of one dropout site.  rows: int (rows 0 .. rows-1) or an int64 array / tensor of ROW INDICES AS THE KERNEL SAW THEM
    (the padding-free encoder runs on compact rows: pass each logical row's compact index).  -> bool tensor [len(rows), D]."""
    if isinstance(rows, int):
        rows = np.arange(rows, dtype=np.uint64)
    else:
        rows = np.asarray(torch.as_tensor(rows).cpu().numpy(), dtype=np.int64).astype(np.uint64)
    s = salted_seed(seed, salt)
    idx = rows[:, None] * np.uint64(D) + np.arange(D, dtype=np.uint64)[None, :]
    lo, hi = idx & _M32, idx >> np.uint64(32)
    r = _hash_u32(lo ^ _hash_u32(hi ^ np.uint64(s & 0xFFFFFFFF)) ^ np.uint64(s >> 32))
    return torch.from_numpy(r < np.uint64(keep_threshold(p_drop)))

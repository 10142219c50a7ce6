"""RNG-free closed-form weight initialisation.

Both sides of every parity test (the reference run in the development container, the CPU
oracle, and the HIP path on the GPU box) regenerate identical weights from (tensor name, shape)
alone, so no weight blob ever has to be shipped.  value[i] = std*sqrt(3)*(2*u-1) where u is a
32-bit integer hash of (crc32(name), flat index i) -- uniform with the requested std.

This replaces, for test/bench purposes only, the reference's `normal_(0, 0.02)` init
(/root/reference/src/transformer/modeling_multimodalsum.py:188-199); there is no network to
fetch `facebook/bart-large`, so random-init weights of the right shape are what the bench uses.
"""
import math
import zlib

import torch


def formula_tensor(name, shape, std=0.02, mean=0.0, device="cpu", dtype=torch.float32):
    """Deterministic pseudo-random tensor: a pure function of (name, flat index)."""
    n = 1
    for s in shape:
        n *= int(s)
    seed = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    i = torch.arange(n, dtype=torch.int64, device=device)
    x = (i * 0x9E3779B1 + seed * 0x85EBCA6B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    u = x.to(torch.float64) / 4294967296.0
    v = (2.0 * u - 1.0) * (math.sqrt(3.0) * std) + mean
    return v.to(dtype).reshape(tuple(shape))


def formula_state_dict(shapes, std=0.02, overrides=None, device="cpu"):
    """shapes: {name: shape}.  LayerNorm/BatchNorm weights get mean 1 (small spread), biases a
    small spread around 0 so that bias/affine code paths are exercised by parity tests."""
    overrides = overrides or {}
    out = {}
    for name, shape in shapes.items():
        if name in overrides:
            s, m = overrides[name]
        elif name.endswith("layer_norm.weight") or name.endswith("layernorm_embedding.weight") \
                or (".bn" in name and name.endswith(".weight")) or name.endswith("downsample.1.weight"):
            s, m = 0.05, 1.0
        elif name.endswith("running_var"):
            s, m = 0.0, 1.0
        elif name.endswith("running_mean") or name.endswith("num_batches_tracked") \
                or name.endswith("final_logits_bias"):
            s, m = 0.0, 0.0
        else:
            s, m = std, 0.0
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros((), dtype=torch.int64, device=device)
        else:
            out[name] = formula_tensor(name, shape, s, m, device=device)
    return out

"""CPU: the C-ABI library loads and exports every symbol include/mmsum_hip.h declares; argument
validation paths that need no GPU return the documented error codes."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmsum_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long)\s+(mmsum_\w+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from multimodalsum_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), "libmmsum_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.lib.mmsum_abi_version() == 1


def test_argument_validation_without_gpu():
    from multimodalsum_amd import _lib
    lib = _lib.lib
    # bad dtype / shape are rejected before anything touches the device
    assert lib.mmsum_gemm(7, None, 0, None, 0, 0, None, 0, None, 0, None, None, 0, 8, 8, 8, 1.0, 0, 1, None) == -2
    assert lib.mmsum_gemm(_lib.BF16, None, 0, None, 0, 0, None, 0, None, 0, None, None, 0, 0, 8, 8, 1.0, 0, 1, None) == -1
    assert lib.mmsum_gemm(_lib.BF16, None, 8, None, 0, 0, None, 8, None, 8, None, None, 0, 8, 8, 12, 1.0, 0, 1, None) == -1  # K % 8
    d = _lib.AttnDesc()
    d.T, d.S, d.N, d.H, d.qpb, d.n_qblocks = 200, 10, 1, 1, 1, 1
    assert lib.mmsum_attn_fwd(_lib.BF16, ctypes.byref(d), None) == -1  # T > 128
    assert lib.mmsum_add_ln_fwd(_lib.F32, None, None, None, None, None, None, None, 4, 100, 1e-5, 0.0, 0, None) == -1  # D unsupported


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from multimodalsum_amd import kernels as kn
    a = torch.zeros(8, 8)
    with pytest.raises(RuntimeError):
        kn.gemm(a, a, a.clone())

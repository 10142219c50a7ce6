"""CPU: the C-ABI library loads and exports every symbol include/mmsum_hip.h declares; argument
validation paths that need no GPU return the documented error codes."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmsum_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long|char\*)\s+(mmsum_\w+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from multimodalsum_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(_lib.lib, s), "libmmsum_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.lib.mmsum_abi_version() == _lib.ABI_VERSION == 10


def test_argument_validation_without_gpu():
    from multimodalsum_amd import _lib
    lib = _lib.lib
    # bad dtype / shape are rejected before anything touches the device
    assert lib.mmsum_gemm(7, None, 0, None, 0, 0, None, 0, None, 0, None, None, 0, 8, 8, 8, 1.0, None, 0, 1, None, None, 0, None) == -2
    assert lib.mmsum_gemm(_lib.BF16, None, 0, None, 0, 0, None, 0, None, 0, None, None, 0, 0, 8, 8, 1.0, None, 0, 1, None, None, 0, None) == -1
    assert lib.mmsum_gemm(_lib.BF16, None, 8, None, 0, 0, None, 8, None, 8, None, None, 0, 8, 8, 12, 1.0, None, 0, 1, None, None, 0, None) == -1  # K % 8
    d = _lib.AttnDesc()
    d.T, d.S, d.N, d.H, d.qpb, d.n_qblocks = 200, 10, 1, 1, 1, 1
    assert lib.mmsum_attn_fwd(_lib.BF16, ctypes.byref(d), None) == -1  # T > 128
    assert lib.mmsum_add_ln_fwd(_lib.F32, None, None, None, None, None, None, None, 4, 100, 1e-5, 0.0, 0, None, None, None, None) == -1  # D unsupported
    # live-image window: rows_per_image must divide the row count; the plan takes at most 8192 slots and 8 row kinds
    assert lib.mmsum_bn_apply(_lib.BF16, None, None, None, None, None, None, None, None, None, 100, 64, 1e-5, 0.1, 1, 1, 0, 0, 1, 7, None) == -1
    assert lib.mmsum_image_plan(None, 10, None, 9000, 4, None, None, 0, None, None, None, None, None, None) == -1
    assert lib.mmsum_image_plan(None, 10, None, 16, 4, None, None, 2, None, None, None, None, None, None) == -1   # row kinds announced, none given


def test_gemm_plan_is_pure_and_reaches_the_benchmarked_kernels():
    """mmsum_gemm_plan needs no GPU: the tile chooser sends the bench shapes (B=56: 64,512 decoder rows, 38,912-ish encoder
    rows) to the persistent 256x256 ring kernels -- the -m gpu parity tests assert the same plan before they compare."""
    from multimodalsum_amd import _lib
    lib = _lib.lib
    plan = (ctypes.c_int * 4)()
    fake = 1 << 20                       # any 16-byte aligned non-NULL address: the plan never dereferences

    def p(M, N, K, flags=0, splitk=1, lda=None, ldb=None, live=None, alpha_dev=None, ws=0):
        at, bt = flags & _lib.GEMM_A_T, flags & _lib.GEMM_B_T
        lda = lda or (M if at else K)
        ldb = ldb or (N if bt else K)
        rc = lib.mmsum_gemm_plan(_lib.BF16, fake, lda, None, 0, 0, fake, ldb, fake, N, None, None, 0, M, N, K, flags, splitk, live, alpha_dev, ws, plan)
        assert rc == 0, rc
        return tuple(plan)

    for M in (38912, 64512, 76288, 129024):
        for N, K in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096), (2048, 1024)):
            k, bm, bn, grid = p(M, N, K)
            assert (k, bm, bn) == (_lib.PLAN_NT_RING, 256, 256), (M, N, K, k, bm, bn)
            assert grid == 256 < (M // 256) * (N // 256)          # persistent: one workgroup per CU walks the tile list
    assert p(64512, 50265, 1024)[:3] == (_lib.PLAN_NT_RING, 256, 256)          # LM head forward (ragged N)
    assert p(64512, 1024, 50304)[:3] == (_lib.PLAN_NT_RING, 256, 256)          # LM head input gradient
    tn = _lib.GEMM_A_T | _lib.GEMM_B_T | _lib.GEMM_OUT_F32 | _lib.GEMM_SLABS
    assert p(4096, 1024, 64512, tn, 8)[:3] == (_lib.PLAN_TN_RING, 256, 256)     # fc1 weight gradient, split-K slabs
    assert p(1000, 520, 128)[0] == _lib.PLAN_NT_RING and p(1000, 520, 128)[1:3] == (128, 128)
    # the small-batch step (1,152 decoder rows at the per-GPU batch 1 of multimodal_train.py:420): one round of 128x128 tiles; with a lent
    # workspace the K = 4,096 products are cut into three reduction slices per tile, the K = 1,024 ones are not (mmsum_gemm)
    ws = 4096 + 256 * 128 * 128 * 4                                             # MMSUM_GEMM_WORKSPACE_BYTES
    assert p(1152, 1024, 4096) == (_lib.PLAN_NT_RING, 128, 128, 72) and p(1152, 1024, 4096, ws=ws) == (_lib.PLAN_NT_RING, 128, 128, 216)
    assert p(1152, 1024, 1024, ws=ws) == (_lib.PLAN_NT_RING, 128, 128, 72) and p(640, 1024, 4096, ws=ws)[3] == 160
    assert p(1152, 1024, 4096, ws=4096 + 100 * 128 * 128 * 4)[3] == 72          # a workspace too small for the slabs: unsplit
    assert p(32, 4096, 1024)[0] == _lib.PLAN_SKINNY                              # decode-step rows
    # the decode LM head: f32 logits, and (bf16 mode) the final LayerNorm's f32 output as the A operand -- weight-streaming kernel only
    assert p(32, 50265, 1024, _lib.GEMM_OUT_F32, ldb=1024)[0] == _lib.PLAN_SKINNY
    assert p(32, 50265, 1024, _lib.GEMM_OUT_F32 | _lib.GEMM_A_F32, ldb=1024)[0] == _lib.PLAN_SKINNY
    assert lib.mmsum_gemm_plan(_lib.BF16, fake, 1024, None, 0, 0, fake, 1024, fake, 4096, None, None, 0, 4096, 4096, 1024,
                               _lib.GEMM_OUT_F32 | _lib.GEMM_A_F32, 1, None, None, 0, plan) == -2      # no tiled kernel reads an f32 A
    # a live row count or a device-side scale takes the product off the weight-streaming kernel: the plan must say so
    assert p(32, 4096, 1024, live=fake)[0] == _lib.PLAN_NT_RING and p(32, 4096, 1024, alpha_dev=fake)[0] == _lib.PLAN_NT_RING
    assert lib.mmsum_gemm_plan(_lib.F32, fake, 1024, None, 0, 0, fake, 1024, fake, 512, None, None, 0, 512, 512, 1024, 0, 1, None, None, 0, plan) == 0
    assert plan[0] == _lib.PLAN_GENERIC


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from multimodalsum_amd import kernels as kn
    a = torch.zeros(8, 8)
    with pytest.raises(RuntimeError):
        kn.gemm(a, a, a.clone())

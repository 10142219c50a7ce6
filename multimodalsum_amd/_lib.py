"""ctypes binding of libmmsum_hip.so (the C ABI declared in include/mmsum_hip.h).

The product path has no fallback: if the shared library is missing or fails to load, importing
this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C multimodalsum_amd/csrc`.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_long, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMSUM_LIB") or os.path.join(_HERE, "csrc", "libmmsum_hip.so")     # MMSUM_LIB: A/B a second build

F32, BF16 = 0, 1
OK = 0
ERRORS = {-1: "bad shape", -2: "bad dtype", -3: "bad alignment", -4: "workspace missing/too small", -5: "HIP launch error"}

GEMM_A_T, GEMM_B_T, GEMM_BIAS = 0x1, 0x2, 0x4
EPI_NONE, EPI_GELU, EPI_GELU_BWD, EPI_RELU, EPI_RELU_BWD = 0, 1, 2, 3, 4
GEMM_ACCUM, GEMM_OUT_F32, GEMM_SLABS, GEMM_COLSUM, GEMM_COLSUM2, GEMM_A_F32 = 0x40, 0x80, 0x100, 0x200, 0x400, 0x800
PLAN_GENERIC, PLAN_NT_RING, PLAN_TN_RING, PLAN_SKINNY = 0, 1, 2, 3
ABI_VERSION = 10


def gemm_epi(e):
    return e << 3


class GemmOperands(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("A2", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p),
                ("lda", c_long), ("lda2", c_long), ("ldb", c_long), ("ldc", c_long)]


class XattnMemory(ctypes.Structure):
    _fields_ = [("k", c_void_p), ("v", c_void_p), ("pad", c_void_p), ("null_entity", c_void_p), ("N", c_int), ("S", c_int)]


class AttnDesc(ctypes.Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p),
                ("ldq", c_long), ("ldk", c_long), ("ldv", c_long), ("ldo", c_long),
                ("pad", c_void_p), ("null_entity", c_void_p),
                ("n_qblocks", c_int), ("T", c_int), ("qpb", c_int), ("N", c_int), ("S", c_int), ("H", c_int),
                ("exclude_self", c_int), ("causal", c_int), ("scale", c_float),
                ("q_rows", c_void_p), ("kv_rows", c_void_p), ("causal_q0", c_int)]


# name -> (restype, argtypes); mirrors include/mmsum_hip.h one to one
SIGNATURES = {
    "mmsum_abi_version": (c_int, []),
    "mmsum_build_id": (ctypes.c_char_p, []),
    "mmsum_gemm": (c_int, [c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p,
                           c_void_p, c_long, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_void_p, c_void_p, c_long, c_void_p]),
    "mmsum_gemm_plan": (c_int, [c_int, c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p,
                                c_long, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_long, ctypes.POINTER(c_int)]),
    "mmsum_decode_cross_attn_workspace": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "mmsum_decode_cross_attn": (c_int, [c_int, c_void_p, c_long, ctypes.POINTER(XattnMemory), c_int, c_long, c_void_p, c_long, c_int, c_int, c_int,
                                        c_float, c_void_p, c_void_p]),
    "mmsum_gemm_pair": (c_int, [ctypes.POINTER(GemmOperands), c_int, c_int, c_int, c_int, c_void_p]),
    "mmsum_dec_gemm_workspace": (c_long, [c_int, c_int, c_int]),
    "mmsum_dec_gemm": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long,
                               c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmsum_slab_reduce": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_long, c_int, c_void_p]),
    "mmsum_colsum_workspace": (c_long, [c_int]),
    "mmsum_colsum": (c_int, [c_int, c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mmsum_embed_ln_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_uint64, c_void_p, c_void_p]),
    "mmsum_embed_ln_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_float, c_uint64, c_void_p, c_void_p]),
    "mmsum_add_ln_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_float, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmsum_add_ln_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p, c_int, c_int, c_float, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmsum_entity_null": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mmsum_attn_fwd": (c_int, [c_int, ctypes.POINTER(AttnDesc), c_void_p]),
    "mmsum_attn_bwd_workspace": (c_long, [ctypes.POINTER(AttnDesc)]),
    "mmsum_attn_bwd": (c_int, [c_int, ctypes.POINTER(AttnDesc), c_void_p, c_long, c_void_p, c_long, c_int, c_void_p, c_long,
                               c_void_p, c_long, c_void_p, c_void_p]),
    "mmsum_gate_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                               c_int, c_int, c_void_p]),
    "mmsum_gate_add_ln_fwd": (c_int, [c_int] + [c_void_p] * 11 + [c_int, c_int, c_int, c_float, c_void_p]),
    "mmsum_gate_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mmsum_ls_loss": (c_int, [c_int, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_int, c_void_p]),
    "mmsum_segment_sum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]),
    "mmsum_l2_workspace": (c_long, []),
    "mmsum_l2norm_sq": (c_int, [c_void_p, c_long, c_void_p, c_int, c_void_p, c_void_p]),
    "mmsum_adamw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_float, c_float,
                            c_float, c_void_p]),
    "mmsum_cast": (c_int, [c_int, c_void_p, c_int, c_void_p, c_long, c_void_p]),
    "mmsum_scale_by_clip": (c_int, [c_void_p, c_long, c_void_p, c_float, c_void_p]),
    "mmsum_rows_gather": (c_int, [c_void_p, c_long, c_int, c_void_p, c_long, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mmsum_bump_u64": (c_int, [c_void_p, ctypes.c_ulonglong, c_void_p]),
    "mmsum_transpose_bf16": (c_int, [c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmsum_transpose_bf16_batched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mmsum_im2col": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p, c_void_p]),
    "mmsum_col2im": (c_int, [c_int, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p, c_void_p]),
    "mmsum_conv_weight_permute": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mmsum_bn_workspace": (c_long, [c_int]),
    "mmsum_bn_reduce": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mmsum_bn_rep_fix": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mmsum_image_plan_workspace": (c_long, [c_int]),
    "mmsum_image_plan": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "mmsum_bn_stats_from_sums": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "mmsum_bn_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                               c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mmsum_bn_bwd_reduce": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p,
                                    c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mmsum_bn_bwd_apply": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "mmsum_conv3x3_gemm": (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmsum_conv3x3_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmsum_maxpool3x3s2": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mmsum_nchw_to_nhwc": (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mmsum_table_gather": (c_int, [c_int] + [c_void_p] * 12 + [c_int, c_int, c_int, c_void_p]),
    "mmsum_table_gather_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mmsum_amazon_table_gather": (c_int, [c_int] + [c_void_p] * 12 + [c_int, c_int, c_int, c_void_p]),
    "mmsum_amazon_table_gather_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mmsum_beam_topk_workspace": (c_long, [c_int, c_int, c_int]),
    "mmsum_beam_topk": (c_int, [c_int, c_void_p, c_long, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_void_p]),
    "mmsum_decode_self_attn": (c_int, [c_int, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int,
                                       c_int, c_float, c_void_p, c_void_p, c_long, c_void_p]),
}


from ._build_id import source_build_id  # noqa: E402,F401


def load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmmsum_hip.so not found at %s -- the HIP extension is mandatory (no CPU/torch fallback). "
                           "Build it: make -C multimodalsum_amd/csrc" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    if lib.mmsum_abi_version() != ABI_VERSION:
        raise RuntimeError("libmmsum_hip.so at %s has ABI version %d, this package binds version %d: rebuild it (make -C multimodalsum_amd/csrc)"
                           % (LIB_PATH, lib.mmsum_abi_version(), ABI_VERSION))
    return lib


lib = load()


def check(rc, what):
    if rc != OK:
        raise RuntimeError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "unknown"), rc))

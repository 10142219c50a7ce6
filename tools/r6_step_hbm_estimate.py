#!/usr/bin/env python3
"""HBM-side bytes of one training step (B = 128) put together from counters: the whole-step FETCH_SIZE / WRITE_SIZE passes do not
finish on this pool (tools/r6_step_hbm.sh: 25-minute limit / rocprofv3 segfault), so ONE representative launch of every kernel family was profiled
in isolation at the bench sizes (tools/r6_family_pmc.sh -> r6_family_pmc*.json) and the step is assembled from
    GEMMs          : algorithmic bytes of every product of the step (bench.py's families.*.by_shape: shape x launches) x the measured
                     traffic / algorithmic ratio of the representative of its class;
    everything else: the representative launch's measured byte rate (GB/s under the profiler) x the family's kernel time per step in the
                     steady-state rocprofv3 summary.
usage: r6_step_hbm_estimate.py <bench.json> <noside_summary.txt> <family_pmc.json> [<family_pmc_attn.json>] [--sq step_sq.json] -> stdout JSON"""
import json
import re
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
bench, summary = json.load(open(args[0])), open(args[1]).read()
fam = {}
for p in args[2:]:
    fam.update(json.load(open(p)))


def rep(tag, sub=None):
    """(traffic / algorithmic ratio, bytes per launch, GB/s) of a profiled program's kernel."""
    ks = fam[tag]
    hit = [v for n, v in ks.items() if sub is not None and sub in n and v["read_bytes"] + v["write_bytes"] > 1e6]
    if hit:
        return hit[0]
    return max(ks.values(), key=lambda v: v["read_bytes"] + v["write_bytes"])      # the program's main kernel


def gemm_alg(M, N, K, tn, gelu=False):
    if tn:                                   # dW[M, N] (f32 slabs / accumulate) = A[K, M]^T B[K, N]
        return K * M * 2 + K * N * 2 + M * N * 4
    return M * K * 2 + N * K * 2 + M * N * 2 * (2 if gelu else 1)


out = {"method": __doc__.split("usage:")[0].strip(), "families": {}}
fams = bench["roofline"]["families"]
tot_r = tot_w = 0.0
# ---- GEMMs
ratio = {}
for tag, shape, tn, gelu in (("gemm_nt_1024", (147456, 1024, 1024), False, False), ("gemm_nt_gelu", (147456, 4096, 1024), False, True),
                             ("gemm_nt_k4096", (147456, 1024, 4096), False, False), ("gemm_nt_lmhead", (147456, 50304, 1024), False, False),
                             ("gemm_tn_1024", (1024, 1024, 147456), True, False), ("gemm_tn_4096", (4096, 1024, 147456), True, False)):
    k = rep(tag)
    a = gemm_alg(*shape, tn, gelu)
    ratio[tag] = {"read_over_alg": k["read_bytes"] / a, "write_over_alg": k["write_bytes"] / a, "algorithmic_bytes": a,
                  "read_bytes": k["read_bytes"], "write_bytes": k["write_bytes"], "gbps_under_profiler": k["gbps"]}
out["gemm_representatives"] = ratio
for key, tn in (("gemm_nt", False), ("gemm_tn", True)):
    r = w = alg = ms_listed = 0.0
    for s in fams[key]["by_shape"]:
        M, N, K = s["MNK"]
        if tn:
            cls = "gemm_tn_4096" if max(M, N) >= 3072 else "gemm_tn_1024"
            gelu = False
        else:
            gelu = N >= 4096 and K == 1024
            cls = "gemm_nt_lmhead" if N > 8192 else ("gemm_nt_gelu" if N >= 3072 else ("gemm_nt_k4096" if K >= 2048 else "gemm_nt_1024"))
        a = gemm_alg(M, N, K, tn, gelu) * s["launches"]
        alg += a
        r += a * ratio[cls]["read_over_alg"]
        w += a * ratio[cls]["write_over_alg"]
        ms_listed += s["ms"]
    scale = fams[key]["ms"] / ms_listed                    # the by_shape table lists the ten largest shapes: the rest by time
    out["families"][key] = {"read_bytes": r * scale, "write_bytes": w * scale, "algorithmic_bytes": alg * scale, "kernel_ms": fams[key]["ms"],
                            "shapes_listed_ms": ms_listed}
    tot_r += r * scale
    tot_w += w * scale
# ---- everything else: rate x time
rows = []
for line in summary.split("\n"):
    m = re.match(r"\s*([\d.]+) ms\s+[\d.]+%\s+calls\s+([\d.]+)\s+avg\s+([\d.]+) us\s+(.*)", line)
    if m:
        rows.append((float(m.group(1)), float(m.group(2)), m.group(4)))
CLASSES = [("add_ln_fwd", "ln_fwd", "add_ln_fwd"), ("embed_ln_fwd", "ln_fwd", "add_ln_fwd"), ("add_ln_bwd", "ln_bwd", "add_ln_bwd"), ("embed_ln_bwd", "ln_bwd", "add_ln_bwd"),
           ("gate_fwd", "gate_fwd", None), ("gate_bwd", "gate_bwd", None), ("ls_loss", "loss", None), ("adamw", "adamw", None),
           ("bn_bwd_apply", "bn_bwd", "bn_bwd_apply"), ("bn_apply", "bn_apply", "bn_apply"), ("bn_", "bn_apply", "bn_apply"), ("slab_reduce", "slab_reduce", None),
           ("attn_tr_bwd_dq_kernel<7", "attn_cross_img4", "bwd_dq"), ("attn_tr_bwd_dkv_kernel<7", "attn_cross_img4", "bwd_dkv"), ("attn_tr_fwd_chunk", "attn_cross_img4", "fwd"),
           ("attn_tr_bwd_self", "attn_self_causal", "bwd_self"), ("attn_tr_fwd_kernel<4, true", "attn_self_causal", "fwd"),
           ("attn_tr_bwd_dq", "attn_cross_text", "bwd_dq"), ("attn_tr_bwd_dkv", "attn_cross_text", "bwd_dkv"), ("attn_tr_fwd", "attn_cross_text", "fwd")]
acc = {}
for ms, calls, name in rows:
    if "gemm" in name:
        continue
    hit = next(((tag, sub) for pat, tag, sub in CLASSES if pat in name), None)
    if hit is None or hit[0] not in fam:
        continue
    try:
        k = rep(*hit)
    except StopIteration:
        continue
    f = hit[0].split("_")[0] if hit[0].startswith("attn") else hit[0]
    d = acc.setdefault({"attn": "attention"}.get(f, f), {"read_bytes": 0.0, "write_bytes": 0.0, "kernel_ms": 0.0})
    share = k["read_bytes"] / (k["read_bytes"] + k["write_bytes"])
    b = k["gbps"] * 1e9 * ms * 1e-3
    d["read_bytes"] += b * share
    d["write_bytes"] += b * (1 - share)
    d["kernel_ms"] += ms
for f, d in acc.items():
    out["families"][f] = d
    tot_r += d["read_bytes"]
    tot_w += d["write_bytes"]
covered = sum(d["kernel_ms"] for d in out["families"].values())
out["hbm_traffic"] = {"read_bytes_per_step": tot_r, "write_bytes_per_step": tot_w, "bytes_per_step": tot_r + tot_w, "kernel_ms_covered": covered,
                      "gbps_over_covered_kernel_time": (tot_r + tot_w) / covered / 1e6,
                      "gbps_over_step_time": (tot_r + tot_w) / (bench["ms_per_step"] * 1e-3) / 1e9,
                      "note": "FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 of isolated representative launches (L2-side counters: reads served by the "
                              "memory-side cache count too), scaled to the step as described in `method`; an estimate, not a whole-step counter pass"}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Benchmark of the hot path: training samples/sec (businesses/sec) of the full multimodal
leave-one-out training step (forward + backward + grad clip + AdamW) on synthetic Yelp-shaped data.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  `value` is whole-job businesses/s with inputs resident in HBM;
`roofline` prices the step against the dense bf16 MFMA peak using the canonical algorithmic FLOP
count of SURVEY.md section 8d (K/V projections counted once per step), plus a live HIP-event
measurement of the dominant kernel (the bf16 MFMA GEMM); `cpu_baseline` times the CPU oracle (the
restated reference algorithm) on the host cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3


def flops_per_business(D, F, V, L_enc, L_dec, NR, S, T, I, P=196, Ft=47, multimodal=True, with_resnet=True):
    """Algorithmic FLOPs (2*MAC) of ONE training step for ONE business, de-duplicated count of
    SURVEY.md section 8d: forward x3 for everything with weights + input grads, ResNet stage 1-2 forward only."""
    R = NR * S
    enc = L_enc * (8 * R * D * D + 4 * R * D * F + 4 * S * D * R)
    nproj_out = 3 if multimodal else 1
    per_pass_layer = (8 * T * D * D + 4 * T * T * D) + 2 * T * D * D + nproj_out * 2 * T * D * D + 4 * T * D * F
    per_pass_layer += (NR - 1) * 4 * T * S * D
    rmem = NR * S
    if multimodal:
        per_pass_layer += 8 * T * D * D + 4 * T * Ft * D + I * 4 * T * P * D
        rmem += Ft + I * P
    dec = L_dec * NR * per_pass_layer + L_dec * 4 * rmem * D * D + NR * 2 * T * D * V
    total = 3.0 * (enc + dec)
    if multimodal:
        total += 0.9e9                                           # table encoder (fwd+bwd)
        if with_resnet:
            total += I * (3.63e9 + 3 * 10.35e9 + 3 * 0.41e9)      # stage1-2 fwd, stage3 + projection fwd+bwd
    return total


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=56,
                    help="businesses per GPU per step (56: 9*56*128 decoder rows = 252 x 256-row GEMM tiles, four full rounds of 256 CUs per "
                         "N=1024 product; ~96 GB of the 288 GB.  28 is 4 %% slower per business, 8 is BASELINE C4's reference-style batch)")
    ap.add_argument("--workload", default="multimodal", choices=["multimodal", "text"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-probe", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="issue every kernel from Python instead of replaying captured HIP graphs")
    return ap.parse_args()


def build(args, device):
    import multimodalsum_amd as mm
    cfg = mm.BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.workload == "multimodal":
        model = mm.MultimodalSum(config=cfg, label_smoothing=0.1, device=device, dtype=dtype)
    else:
        model = mm.TextSupervised(config=cfg, label_smoothing=None, device=device, dtype=dtype)
    model.train()
    return cfg, model


def make_batches(args, cfg, device, rank, n=2):
    from multimodalsum_amd import synthetic as syn
    out = []
    for i in range(n):
        b = syn.yelp_batch(args.batch, 9, 128, 4 if args.workload == "multimodal" else 1, cfg.vocab_size,
                           seed=1234 + 1000 * rank + i, img_hw=224 if args.workload == "multimodal" else 8)
        dev = syn.batch_to(b, device)
        # what a loader-side prefetcher does on the host copy (multimodalsum_amd/prefetch.py): the number of non-padding review tokens
        dev["reviews_mask"]._mmsum_valid_rows = int(b["reviews_mask"].ne(0).sum())
        dev["img_mask"]._mmsum_valid_rows = int(b["img_mask"].ne(0).sum())
        out.append(dev)
    return out


def run_step(args, model, opt, sch, b):
    from multimodalsum_amd import optim
    if args.workload == "multimodal":
        loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"], b["field_value"], b["img"], b["img_mask"])[0]
    else:
        loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"])[0]
    opt.zero_grad()
    loss.backward()
    optim.clip_grad_norm_(model.parameters(), 1.0, fused=True)
    opt.step()
    sch.step()
    return loss


def kernel_probe(dtype, batch=14):
    """Live HIP-event timing of the dominant kernel: the MFMA GEMM on the decoder FFN shape at the
    bench batch (M = 9*B*128 rows, fc1: N=4096, K=1024).  Events are recorded on the launch stream."""
    from multimodalsum_amd import kernels as kn
    M, N, K = 9 * batch * 128, 4096, 1024
    a = torch.randn(M, K, device="cuda").to(dtype)
    w = torch.randn(N, K, device="cuda").to(dtype)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    for _ in range(3):
        kn.gemm(a, w, out)
    iters = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        kn.gemm(a, w, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * M * N * K
    return {"kernel": "gemm_nt_ring_kernel<256,256,2,4> (bf16 NT GEMM, 4-stage LDS-DMA ring, 256x256x32 slabs)" if dtype == torch.bfloat16
            else "gemm_kernel<f32,NT>",
            "shape": [M, N, K], "avg_launch_ms": ms, "flops_per_launch": fl, "achieved": fl / ms / 1e9, "unit": "TFLOP/s"}


def padding_note(args, model, b):
    """What the padding-free encoder / K-V projections skip in this run (results identical; the FLOP count used for the
    roofline fraction stays the reference's padded one)."""
    from multimodalsum_amd.modules import _encoder_capacity
    batch = ((b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"], b["field_value"], b["img"], b["img_mask"])
             if args.workload == "multimodal" else (b["reviews"], b["reviews_mask"], b["reviews_rating"]))
    cap = _encoder_capacity(model, batch)
    if cap is None or cap[0] is None:
        return None
    R = b["reviews_mask"].numel()
    return {"encoder_rows_computed": cap[0], "encoder_rows_padded": R, "memory_rows_projected": cap[1],
            "note": "rows that are padding never reach a result; roofline.step keeps the padded FLOP count of SURVEY 8d"}


def pmc_traffic(shape):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (FETCH_SIZE doubled per
    the gfx950 correction + WRITE_SIZE); None when no profile of this exact shape is committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r01_dominant_gemm_pmc*.json"))):
        try:
            with open(path) as f:
                prof = json.load(f)
            if list(prof["shape"]) == list(shape):
                return prof["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def cpu_baseline(args, cfg):
    """The reference algorithm (CPU oracle, literal: 9 sequential passes, K/V re-projected per pass)
    on the host cores.  Bounded sample: B=1, BART-large width/vocab, 1 encoder + 1 decoder layer,
    1 image; businesses/s for the full model is extrapolated by the literal FLOP ratio."""
    from multimodalsum_amd import synthetic as syn
    from multimodalsum_amd.formula_init import formula_state_dict
    from oracle import bart_oracle as bo, encoders_oracle as eo, step_oracle as so
    try:
        cores = len(os.sched_getaffinity(0))       # cores this process may actually run on (cgroup/affinity aware)
    except AttributeError:
        cores = os.cpu_count() or 1
    try:       # cgroup v2 CPU quota (containers often expose every host core but only a slice of CPU time)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    cores = max(1, min(cores, 32))
    torch.set_num_threads(cores)
    L = 1
    ocfg = bo.BartCfg(vocab_size=cfg.vocab_size, d_model=cfg.d_model, ffn_dim=cfg.encoder_ffn_dim, encoder_layers=L,
                      decoder_layers=L, heads=cfg.heads, max_position_embeddings=cfg.max_position_embeddings, dropout=0.1)
    multimodal = args.workload == "multimodal"
    shapes = bo.bart_param_shapes(ocfg, multimodal, prefix="bart_model.")
    if multimodal:
        shapes.update(eo.table_param_shapes())
    sd = formula_state_dict(shapes, std=0.02)
    if multimodal:
        sd.update(formula_state_dict(eo.resnet_param_shapes(cfg.d_model), std=0.05))
    for k, v in sd.items():
        if v.is_floating_point() and v.dim() > 0 and "running" not in k:
            v.requires_grad_(True)
    I = 1
    b = syn.yelp_batch(1, 9, 128, I, cfg.vocab_size, seed=1234, img_hw=224 if multimodal else 8)

    def step():
        if multimodal:
            loss = so.multimodal_step_loss(sd, ocfg, b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"],
                                           b["field_value"], b["img"], b["img_mask"], 0.1, training=True)
        else:
            loss = so.text_step_loss(sd, ocfg, b["reviews"], b["reviews_mask"], b["reviews_rating"], None, training=True)
        loss.backward()

    step()                       # warm-up (allocator, thread pool)
    t0 = time.time()
    nstep = 0
    while nstep < 12 and (nstep == 0 or time.time() - t0 < 12.0):      # ~10-30 s of CPU work
        step()
        nstep += 1
    total = time.time() - t0
    dt = total / nstep
    D, F, V = cfg.d_model, cfg.encoder_ffn_dim, cfg.vocab_size

    def literal(Lx, Ix):         # reference-literal FLOPs: K/V re-projected in all 9 passes, q x3
        f = flops_per_business(D, F, V, Lx, Lx, 9, 128, 128, Ix, multimodal=multimodal)
        kv = Lx * 4 * (9 * 128 + (47 + Ix * 196 if multimodal else 0)) * D * D
        kv_lit = Lx * 9 * 4 * (8 * 128 + (47 + Ix * 196 if multimodal else 0)) * D * D
        q_extra = Lx * 9 * (2 if multimodal else 0) * 2 * 128 * D * D
        return f + 3.0 * (kv_lit - kv + q_extra)

    sample_flops = literal(L, I)
    full_flops = literal(cfg.encoder_layers, 4 if multimodal else 1)
    return {"value": (sample_flops / dt) / full_flops, "unit": "businesses/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle (PyTorch fp32, literal reference algorithm) fwd+bwd of one B=1 step with 1+1 layers, "
                      "BART-large width/vocab, %d image: %d steps in %.1f s (%.0f GFLOP/s); extrapolated to the 12+12-layer, "
                      "%d-image step by the reference-literal FLOP count" % (I, nstep, total, sample_flops / dt / 1e9, 4 if multimodal else 1),
            "sample_seconds": total}


def cpu_baseline_bounded(args, budget_s=240):
    """Runs cpu_baseline() in a child process (no GPU touched there) under a hard wall-clock budget."""
    import subprocess
    code = ("import sys, json, types; sys.path.insert(0, %r); import bench; from multimodalsum_amd.config import BartConfig; "
            "cfg = BartConfig.from_json_file(%r); "
            "print('CPUBASE ' + json.dumps(bench.cpu_baseline(types.SimpleNamespace(workload=%r), cfg)))"
            % (ROOT, os.path.join(ROOT, "cfg", "bart-large.json"), args.workload))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=budget_s, env=env)
        for line in r.stdout.splitlines():
            if line.startswith("CPUBASE "):
                return json.loads(line[8:])
        return {"value": None, "unit": "businesses/s", "cores": None, "kind": "port", "sample": "cpu baseline failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "businesses/s", "cores": None, "kind": "port",
                "sample": "cpu baseline exceeded its %d s budget on this host" % budget_s}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_ddp = os.environ.get("MMSUM_FORCE_DDP") == "1"          # debugging aid: run the RCCL gradient path at world size 1
    if world > 1 or force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", init_method="env://")
    import multimodalsum_amd as mm
    from multimodalsum_amd import optim
    cfg, model = build(args, device)
    runner = mm.DistributedDataParallel(model, delay_allreduce=True, always_reduce=force_ddp) if (world > 1 or force_ddp) else model
    opt = optim.get_optimizer(1e-5, ('bias', 'bn1.weight', 'bn2.weight', 'bn3.weight', 'layer_norm.weight', 'layernorm_embedding.weight'),
                              model.named_parameters(), None)
    sch = optim.get_linear_schedule_with_warmup(opt, 100, 100000)
    batches = make_batches(args, cfg, device, rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    priming = 0
    if not args.no_graphs:
        model.enable_step_graphs()
        priming = 2                         # un-timed: one eager step, one that captures the HIP graphs
        for i in range(priming):
            run_step(args, runner, opt, sch, batches[i % len(batches)])
    for i in range(args.warmup):
        run_step(args, runner, opt, sch, batches[i % len(batches)])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = run_step(args, runner, opt, sch, batches[i % len(batches)])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        multimodal = args.workload == "multimodal"
        fpb = flops_per_business(cfg.d_model, cfg.encoder_ffn_dim, cfg.vocab_size, cfg.encoder_layers, cfg.decoder_layers, 9, 128, 128,
                                 4 if multimodal else 1, multimodal=multimodal)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        achieved = value / world * fpb / 1e12
        step_roof = {"achieved": achieved, "frac": achieved / peak, "flops_per_business": fpb,
                     "scope": "whole training step per GPU (algorithmic FLOPs of SURVEY.md 8d / step time)"}
        if args.no_kernel_probe:
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                    "scope": step_roof["scope"], "flops_per_business": fpb}
        else:
            # the dominant kernel, timed live with HIP events on its launch stream; traffic from the committed PMC profile
            dk = kernel_probe(torch.bfloat16 if args.dtype == "bf16" else torch.float32, args.batch)
            roof = {"bound": "mfma", "achieved": dk["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": dk["achieved"] / peak,
                    "traffic": pmc_traffic(dk["shape"]), "kernel": dk["kernel"], "shape": dk["shape"], "avg_launch_ms": dk["avg_launch_ms"],
                    "flops_per_launch": dk["flops_per_launch"], "step": step_roof}
        out = {"metric": "training samples/sec (businesses/sec) BART-large multimodal" if multimodal else
               "training samples/sec (businesses/sec) BART-large text-only",
               "value": value, "unit": "businesses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic Yelp-shaped batches (seeded), formula-initialised BART-large/ResNet101 weights",
               "config": {"workload": "multimodal_train.py full text+img(4x224^2)+table step (fwd+bwd+clip+AdamW), 9 reviews x 128 tok"
                          if multimodal else "text_pretrain.py BART-large text-only step, 9 reviews x 128 tok",
                          "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "dropout": cfg.dropout},
               "padding_free": padding_note(args, model, batches[0]),
               "launch": "eager" if args.no_graphs else "hip-graph replay (1 forward graph + 1 graph per backward gradient segment), %d priming steps before warmup" % priming,
               "final_loss": loss_val, "peak_hbm_gb": round(torch.cuda.max_memory_reserved() / 2**30, 1), "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_bounded(args)
    if world > 1 or force_ddp:
        dist.destroy_process_group()          # RCCL prints its library banner on teardown: keep the JSON line last
    if rank == 0:
        sys.stdout.flush()
        try:                                   # RCCL's banner sits in the C stdio buffer until exit: push it out first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

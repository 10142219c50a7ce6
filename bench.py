#!/usr/bin/env python3
"""Benchmark of the hot path: training samples/sec (businesses/sec) of the full multimodal
leave-one-out training step (forward + backward + grad clip + AdamW) on synthetic Yelp-shaped data.

    python bench.py --gpus N --steps K --warmup W

N > 1: bench.py starts the N ranks itself (`python -m torch.distributed.run ... bench.py`, one process per GPU,
RCCL) before it touches a GPU, relays rank 0's JSON line and exits with the child's code; when it is ALREADY
running under torch.distributed.run (WORLD_SIZE set, the driver's form of the launch) it is one of the ranks.

Prints ONE JSON line (rank 0).  `value` is whole-job businesses/s with inputs resident in HBM -- a fresh batch with
fresh review lengths and image counts is generated ON THE DEVICE inside every timed step.
`roofline` describes the step's dominant kernel as it runs IN the step: after the timed region one more step is
issued eagerly with HIP events around every GEMM and attention launch (on the launch stream), so `achieved` is the
kernel's algorithmic FLOPs / its in-step duration, time-weighted over all its launches of the step, and
`roofline.families` gives the same for the two kernel families that make the step (every GEMM; every attention
launch), so the step number can be decomposed from the JSON line; `roofline.step` prices the whole step against the
dense bf16 MFMA peak with the canonical algorithmic FLOP count of SURVEY.md section 8d (padded rows, K/V projections
once per step) and, beside it, with the FLOPs actually executed (valid rows only).  `cpu_baseline` times the CPU
oracle (the restated reference algorithm, literal) on the host cores: whole B=1 steps of the same configuration
(forward, backward, clip, AdamW), no extrapolation.  `also` carries the other BASELINE configurations measured in
the same run (text-only step, per-GPU batches of 56 -- the headline of rounds 1-2 -- and 8, beam-search generation), a few
steps each.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ROCm runtime switch, read when the HIP runtime loads (i.e. before `import torch`): kernel arguments are written to device memory instead of
# host-coherent memory, so a small kernel starts about a microsecond earlier.  Same results; same box, interleaved (profiles/r05_kernarg_ab.txt):
# B = 1 32.4 -> 34.1 businesses/s, B = 8 142.9 -> 146.0, B = 128 and the decode step unchanged.  An exported value wins; the line reports it.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

PEAK_BF16_TFLOPS = 2500.0     # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
NO_DECAY = ('bias', 'bn1.weight', 'bn2.weight', 'bn3.weight', 'layer_norm.weight', 'layernorm_embedding.weight')


TELEMETRY_CODE = r"""
import glob, os, re, subprocess, sys, time
path = sys.argv[1]
parent = int(sys.argv[2])
deadline = time.time() + 3600.0
def alive():
    # the sampler must never outlive the bench process (killed by a timeout, an exception before read_telemetry, ...)
    return os.getppid() == parent and time.time() < deadline
def cards():
    out = []
    for h in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
        p = [os.path.join(h, n) for n in ('power1_average', 'power1_input') if os.path.exists(os.path.join(h, n))]
        f = os.path.join(h, 'freq1_input')
        if p and os.path.exists(f):
            bdf = os.path.basename(os.path.realpath(os.path.join(h, '..', '..')))
            out.append((bdf, p[0], f, os.path.join(h, 'power1_cap')))
    return out
def rd(p):
    with open(p) as f:
        return float(f.read().strip())
src = cards()
with open(path, 'w') as f:
    if src:
        for i, (bdf, p, fr, cap) in enumerate(src):
            try:
                c = rd(cap) / 1e6
            except Exception:
                c = None
            f.write('# card %d %s cap %s\n' % (i, bdf, c))
    else:
        cap = None
        try:
            out = subprocess.run(['rocm-smi', '--showmaxpower'], capture_output=True, text=True, timeout=20).stdout
            m = re.search(r'Power \(W\): ([\d.]+)', out)
            cap = float(m.group(1)) if m else None
        except Exception:
            pass
        f.write('# card 0 rocm-smi cap %s\n' % cap)
    f.flush()
    while alive():
        try:
            if src:
                t = time.time()
                for i, (bdf, p, fr, cap) in enumerate(src):
                    f.write('%.3f %d %.1f %.0f\n' % (t, i, rd(p) / 1e6, rd(fr) / 1e6))
                f.flush()
                time.sleep(0.05)
            else:
                out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=10).stdout
                p = re.search(r'Power \(W\): ([\d.]+)', out)
                c = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
                if p and c:
                    f.write('%.3f 0 %.1f %.0f\n' % (time.time(), float(p.group(1)), float(c.group(1))))
                    f.flush()
                time.sleep(0.1)
        except Exception:
            time.sleep(0.2)
try:
    if not alive():
        os.unlink(path)
except Exception:
    pass
"""


def start_telemetry():
    """Socket power and shader clock, sampled by a CHILD process started before this process touches a GPU (read-only: the amdgpu
    hwmon files of every card the node shows, or rocm-smi where they are absent).  Returns (process, sample file) or (None, None)."""
    import tempfile
    try:
        fd, path = tempfile.mkstemp(prefix="mmsum_telemetry_", suffix=".txt")
        os.close(fd)
        proc = subprocess.Popen([sys.executable, "-c", TELEMETRY_CODE, path, str(os.getpid())], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))

        def cleanup():                 # any exit path of this process (SystemExit without a GPU, an exception in the step, ...): the
            try:                       # child also watches its parent's pid itself, which covers SIGKILL / a `timeout` wrapper
                if proc.poll() is None:
                    proc.terminate()
                    proc.wait(timeout=5)
            except Exception:
                pass
            try:
                os.unlink(path)
            except OSError:
                pass
        import atexit
        atexit.register(cleanup)
        return proc, path
    except Exception:
        return None, None


def read_telemetry(proc, path, t0, t1, bdf=None):
    """Mean / extreme power and clock of the samples taken inside [t0, t1] (the timed region; wall clock) on the card whose PCI address
    is `bdf` (this process's device; the node may show other GPUs, idle or busy with somebody else's work).  Without a match: the
    card that drew the most power inside the window, and the line says so."""
    if proc is None:
        return None
    try:
        proc.terminate()
        proc.wait(timeout=5)
    except Exception:
        pass
    try:
        with open(path) as f:
            lines = f.read().splitlines()
        os.unlink(path)
        cards = {}
        for ln in lines:
            if ln.startswith("# card"):
                _, _, idx, name, _, cap = ln.split()
                cards[int(idx)] = (name, None if cap == "None" else float(cap))
        rows = [tuple(float(x) for x in ln.split()) for ln in lines if ln and not ln.startswith("#")]
        mine = [r for r in rows if t0 <= r[0] <= t1]
        if not mine:
            return {"samples": 0, "cards": len(cards), "note": "no sample fell inside the timed region"}
        pick, how = None, "pci address of the process's device"
        if bdf is not None:
            pick = next((i for i, (name, _) in cards.items() if name.lower() == bdf.lower()), None)
        if pick is None:
            means = {}
            for r in mine:
                means.setdefault(int(r[1]), []).append(r[2])
            pick = max(means, key=lambda i: sum(means[i]) / len(means[i]))
            how = "the card with the highest mean power inside the window (no PCI address match)" if len(cards) > 1 else "the only card shown"
        sel = [r for r in mine if int(r[1]) == pick]
        pw, ck = [r[2] for r in sel], [r[3] for r in sel]
        name, cap = cards.get(pick, (None, None))
        return {"power_w_mean": sum(pw) / len(pw), "power_w_max": max(pw), "sclk_mhz_mean": sum(ck) / len(ck), "sclk_mhz_min": min(ck),
                "sclk_mhz_max": max(ck), "power_cap_w": cap, "samples": len(sel), "card": name, "cards_shown": len(cards), "selected_by": how,
                "window": "the timed region (barrier to barrier), sampled by a child process started before the first GPU call"}
    except Exception as exc:
        return {"samples": 0, "error": repr(exc)[:200]}


RESNET_FLOPS_PER_IMAGE = 3.63e9 + 3 * 10.35e9 + 3 * 0.41e9      # stage 1-2 forward, stage 3 + projection forward + backward (224 x 224)


def flops_per_business(D, F, V, L_enc, L_dec, NR, S, T, I, P=196, Ft=47, multimodal=True, with_resnet=True, enc_rows=None,
                       mem_rows=None, img_run=None):
    """Algorithmic FLOPs (2*MAC) of ONE training step for ONE business, de-duplicated count of
    SURVEY.md section 8d: forward x3 for everything with weights + input grads, ResNet stage 1-2 forward only.
    enc_rows / mem_rows (per business): rows the padding-free encoder layers / K-V projections really work on -- the
    'executed' count; None = all padded rows, the canonical count.  img_run (per business): images the image branch really runs (the filled
    slots + the one representative of the batch's empty slots); None = all I slots."""
    R = NR * S
    Re = R if enc_rows is None else enc_rows
    enc = L_enc * (8 * Re * D * D + 4 * Re * D * F + 4 * S * D * R)
    nproj_out = 3 if multimodal else 1
    per_pass_layer = (8 * T * D * D + 4 * T * T * D) + 2 * T * D * D + nproj_out * 2 * T * D * D + 4 * T * D * F
    per_pass_layer += (NR - 1) * 4 * T * S * D
    rmem = NR * S
    if multimodal:
        per_pass_layer += 8 * T * D * D + 4 * T * Ft * D + I * 4 * T * P * D
        rmem += Ft + I * P
    if mem_rows is not None:
        rmem = mem_rows
    dec = L_dec * NR * per_pass_layer + L_dec * 4 * rmem * D * D + NR * 2 * T * D * V
    total = 3.0 * (enc + dec)
    if multimodal:
        total += 0.9e9                                           # table encoder (fwd+bwd)
        if with_resnet:
            total += (I if img_run is None else img_run) * RESNET_FLOPS_PER_IMAGE
    return total


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="businesses per GPU per step; default 128 for the training workloads (9*128*128 decoder rows = 576 x 256-row GEMM "
                         "tiles: exactly 9 rounds of the 256 CUs per N=1024 product; 228 GB of the 288 GB.  Measured on one box: 128 -> 239.2, "
                         "112 -> 235.8, 56 -> 225.7 businesses/s, 28 is 4 %% below 56; 8 is BASELINE C4's reference-style batch) and 8 for "
                         "--workload generate (test.py:176)")
    ap.add_argument("--workload", default="multimodal", choices=["multimodal", "text", "text_table", "generate"],
                    help="multimodal / text / text_table: the training step (BASELINE configs 4 / 2 / 3: text + table, one all-zero image slot per "
                         "business with img_mask False everywhere, multimodal_train.py:165-193); generate: test.py's beam search (BASELINE config 5)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-probe", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="issue every kernel from Python instead of replaying captured HIP graphs")
    ap.add_argument("--fixed-batches", action="store_true", help="alternate two pre-generated batches instead of generating one per step")
    ap.add_argument("--host-batches", action="store_true",
                    help="PCIe-inclusive rate: the step's inputs start in pinned HOST memory and reach the HBM through yelp_data_prefetcher "
                         "(side-stream copies under the previous step, multimodal_train.py:196-268); never the headline `value`")
    ap.add_argument("--grad-dtype", default="f32", choices=["f32", "bf16"], help="N > 1: dtype of the gradient buckets on the wire")
    ap.add_argument("--ddp-mode", default="all_reduce", choices=["all_reduce", "reduce_scatter"],
                    help="N > 1: one all-reduce per gradient bucket, or reduce-scatter + all-gather")
    ap.add_argument("--no-also", action="store_true", help="skip the extra configurations (text-only step, batch 8, generation)")
    ap.add_argument("--diag-stub-resnet", action="store_true",
                    help="DIAGNOSTIC (the line is marked invalid): the image encoder's forward / backward are replaced by a zero fill, to "
                         "measure how much of the step's wall time the ResNet branch is responsible for")
    ap.add_argument("--master-port", type=int, default=29517)
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = 8 if args.workload == "generate" else 128
    return args


# ------------------------------------------------------------------------------------------------
# N > 1 from a plain `python bench.py --gpus N`: start the ranks as children BEFORE any GPU call
# ------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    last_json = None
    for line in proc.stdout:
        if line.startswith("{") and '"metric"' in line:
            last_json = line.strip()
        else:
            sys.stderr.write(line)                # RCCL banners etc.: keep stdout to the one JSON line
    rc = proc.wait()
    if last_json is not None:
        print(last_json, flush=True)
    sys.exit(rc if rc != 0 or last_json is not None else 1)


def build(args, device):
    import torch
    import multimodalsum_amd as mm
    cfg = mm.BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if args.workload in ("multimodal", "text_table"):
        model = mm.MultimodalSum(config=cfg, label_smoothing=0.1, device=device, dtype=dtype)
    else:
        model = mm.TextSupervised(config=cfg, label_smoothing=None, device=device, dtype=dtype)
    model.train()
    if getattr(args, "diag_stub_resnet", False) and args.workload in ("multimodal", "text_table"):
        e = model._engine

        def img_fwd(img, out=None, img_mask=None):
            y = out if out is not None else e.empty(img.shape[0] * 196, cfg.d_model)
            y.zero_()
            return y, None
        e.img_fwd, e.img_bwd = img_fwd, (lambda c, dy: None)
    return cfg, model


def batch_source(args, cfg, device, rank):
    """A callable returning the next step's inputs (device tensors)."""
    from multimodalsum_amd import synthetic as syn
    multimodal = args.workload == "multimodal"
    I, hw = (4, 224) if multimodal else ((1, 224) if args.workload == "text_table" else (1, 8))
    if args.workload == "text_table":      # BASELINE config 3: the reference cannot run with I = 0 (multimodal_train.py:189), so one zero image, masked out
        gen = syn.DeviceBatches(args.batch, 9, 128, 1, cfg.vocab_size, device, seed=1234 + rank, img_hw=224, no_images=True)
        return gen.next
    if getattr(args, "host_batches", False):
        # two host batches (pinned), served in turn for ever through the reference's prefetcher protocol: every step's inputs cross PCIe
        import itertools
        from multimodalsum_amd import yelp_data_prefetcher
        hosts = []
        for i in range(2):
            hb = syn.yelp_batch(args.batch, 9, 128, I, cfg.vocab_size, seed=1234 + 1000 * rank + i, img_hw=hw)
            fv = hb["field_value"]
            hosts.append(tuple(t.pin_memory() for t in (hb["reviews"], hb["reviews_mask"], hb["reviews_rating"], fv[0], fv[1], fv[2], fv[3], fv[4], fv[5],
                                                         hb["img"], hb["img_mask"])))
        field = hosts and syn.yelp_batch(1, 9, 128, I, cfg.vocab_size, seed=1234, img_hw=8)["field"].to(device)
        pf = yelp_data_prefetcher(itertools.cycle(hosts), device)
        args.host_bytes_per_step = sum(t.numel() * t.element_size() for t in hosts[0])

        def nxt_host():
            reviews, mask, rating, fvd, img, img_mask = pf.next()
            return {"reviews": reviews, "reviews_mask": mask, "reviews_rating": rating, "field": field, "field_value": fvd, "img": img, "img_mask": img_mask}
        return nxt_host
    if args.fixed_batches:
        fixed = [syn.batch_to(syn.yelp_batch(args.batch, 9, 128, I, cfg.vocab_size, seed=1234 + 1000 * rank + i, img_hw=hw), device)
                 for i in range(2)]
        state = {"i": 0}

        def nxt():
            state["i"] += 1
            return fixed[state["i"] % 2]
        return nxt
    gen = syn.DeviceBatches(args.batch, 9, 128, I, cfg.vocab_size, device, seed=1234 + rank, img_hw=hw)
    return gen.next


def run_step(args, model, opt, sch, b):
    from multimodalsum_amd import optim
    if args.workload in ("multimodal", "text_table"):
        loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"], b["field_value"], b["img"], b["img_mask"])[0]
    else:
        loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"])[0]
    opt.zero_grad()
    loss.backward()
    optim.clip_grad_norm_(model.parameters(), 1.0, fused=True)
    opt.step()
    sch.step()
    return loss


# ------------------------------------------------------------------------------------------------
# the dominant kernel and the two kernel families, timed where they run: inside a step
# ------------------------------------------------------------------------------------------------
def probe_step_kernels(args, model, runner, opt, sch, b, cfg):
    """One more training step, issued eagerly (no graph replay, image / table branch on the main stream so that nothing runs
    beside the kernel being timed), with a HIP event pair around EVERY GEMM launch and every attention launch.  Events are
    recorded on the stream the kernel is launched on (torch's current stream at the call).
    Returns (dominant, families):
      dominant: the FFN up-projection + bias + GELU (+ saved pre-activation) NT GEMM, `gemm_nt_w4_kernel<EPI_GELU, OUT_T>` --
                24 launches per step (12 decoder layers on all 9*B*128 rows, 12 encoder layers on the valid rows); `achieved`
                is time-weighted over all of them, the decoder / encoder split is kept beside it;
      families: {"gemm": ..., "gemm_nt": ..., "gemm_tn": ..., "attention": ...}: launches, milliseconds and FLOPs of the step."""
    import torch
    from multimodalsum_amd import engine as eng_mod, kernels as kn
    e = model._engine
    Fd, D = cfg.decoder_ffn_dim, cfg.d_model
    real_gemm, real_fwd, real_bwd = kn.gemm, kn.attn_fwd, kn.attn_bwd
    gemms, attns = [], []

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def timed_gemm(a, w, out, *pos, **kw):
        a_t, b_t = bool(kw.get("a_t")), bool(kw.get("b_t"))
        M, K = (a.shape[1], a.shape[0]) if a_t else (a.shape[0], a.shape[1])
        N = w.shape[1] if b_t else w.shape[0]
        if kw.get("a2") is not None:
            K += kw["a2"].shape[1]
        e0, e1 = ev(), ev()
        e0.record()
        r = real_gemm(a, w, out, *pos, **kw)
        e1.record()
        gemms.append((e0, e1, M, N, K, a_t and b_t, kw.get("live"), kw.get("epi") == kn.EPI_GELU and not a_t and w.shape == (Fd, D) and M >= 4096))
        return r

    def attn_flops(d, mult):      # canonical: every padded key, the full T x S rectangle also under the causal mask; head_dim 64
        return mult * 4.0 * d.n_qblocks * d.T * d.H * 64 * d.S * (d.N - (1 if d.exclude_self else 0))

    def timed_fwd(desc, t):
        e0, e1 = ev(), ev()
        e0.record()
        real_fwd(desc, t)
        e1.record()
        attns.append((e0, e1, attn_flops(desc, 1.0)))

    def timed_bwd(desc, *a):
        e0, e1 = ev(), ev()
        e0.record()
        real_bwd(desc, *a)
        e1.record()
        attns.append((e0, e1, attn_flops(desc, 2.5)))       # five products against the forward's two

    # the image branch as a whole (ResNet101 stages 1-3 + projection, forward and layer3's backward): on the main stream in this step, so the
    # event pairs around img_fwd / img_bwd bracket exactly its kernels
    resnet = {"fwd": None, "bwd": None, "plan": None, "n": 0}
    real_img_fwd, real_img_bwd = getattr(e, "img_fwd", None), getattr(e, "img_bwd", None)

    def timed_img_fwd(img, *a, **kw):
        e0, e1 = ev(), ev()
        e0.record()
        r = real_img_fwd(img, *a, **kw)
        e1.record()
        resnet["fwd"], resnet["n"] = (e0, e1), img.shape[0]
        ip = getattr(r[1], "ip", None)
        resnet["plan"] = ip.plan if ip is not None else None
        return r

    def timed_img_bwd(*a, **kw):
        e0, e1 = ev(), ev()
        e0.record()
        r = real_img_bwd(*a, **kw)
        e1.record()
        resnet["bwd"] = (e0, e1)
        return r

    graphs = getattr(model, "_step_graphs", None)
    object.__setattr__(model, "_step_graphs", None)
    had_side = hasattr(e, "_side_stream")
    side = getattr(e, "_side_stream", None)
    e._side_stream = None
    kn.gemm, kn.attn_fwd, kn.attn_bwd = timed_gemm, timed_fwd, timed_bwd
    if e.with_img and not getattr(args, "diag_stub_resnet", False):
        e.img_fwd, e.img_bwd = timed_img_fwd, timed_img_bwd
    try:
        run_step(args, runner, opt, sch, b)
        torch.cuda.synchronize()
    finally:
        kn.gemm, kn.attn_fwd, kn.attn_bwd = real_gemm, real_fwd, real_bwd
        if "img_fwd" in e.__dict__ and e.__dict__["img_fwd"] is timed_img_fwd:
            del e.__dict__["img_fwd"], e.__dict__["img_bwd"]
        object.__setattr__(model, "_step_graphs", graphs)
        if had_side:
            e._side_stream = side
        else:
            del e._side_stream
    fam = {k: {"launches": 0, "ms": 0.0, "flops": 0.0} for k in ("gemm_nt", "gemm_tn", "attention")}
    shapes = {"gemm_nt": {}, "gemm_tn": {}}          # per launch shape [M, N, K] as called (row CAPACITY; live rows go into the FLOPs)
    dec, enc = [], []
    for e0, e1, M, N, K, tn, live, dominant in gemms:
        ms = e0.elapsed_time(e1)
        cap = (M, N, K)
        lv = int(live.item()) if live is not None else None
        if lv is not None:
            M, K = (M, min(K, lv)) if tn else (min(M, lv), K)
        name = "gemm_tn" if tn else "gemm_nt"
        f = fam[name]
        f["launches"] += 1
        f["ms"] += ms
        f["flops"] += 2.0 * M * N * K
        sh = shapes[name].setdefault(cap + (lv is not None,), [0, 0.0, 0.0])
        sh[0] += 1
        sh[1] += ms
        sh[2] += 2.0 * M * N * K
        if dominant:
            (dec if lv is None else enc).append((ms, M))
    for e0, e1, fl in attns:
        f = fam["attention"]
        f["launches"] += 1
        f["ms"] += e0.elapsed_time(e1)
        f["flops"] += fl
    fam["gemm"] = {k: fam["gemm_nt"][k] + fam["gemm_tn"][k] for k in ("launches", "ms", "flops")}
    if resnet["fwd"] is not None and resnet["bwd"] is not None:
        n = resnet["n"]
        plan = resnet["plan"].cpu().tolist() if resnet["plan"] is not None else None
        n_run = plan[0] if plan is not None else n
        ms_f, ms_b = resnet["fwd"][0].elapsed_time(resnet["fwd"][1]), resnet["bwd"][0].elapsed_time(resnet["bwd"][1])
        fam["resnet"] = {"launches": 2, "ms": ms_f + ms_b, "ms_forward": ms_f, "ms_backward": ms_b, "flops": n_run * RESNET_FLOPS_PER_IMAGE,
                         "flops_all_slots": n * RESNET_FLOPS_PER_IMAGE, "image_slots": n, "images_run": n_run,
                         "empty_slots": (plan[2] if plan is not None and plan[1] >= 0 else 0),
                         "scope": "engine.img_fwd + engine.img_bwd as issued on the main stream (every kernel of the image branch: layout, im2col, "
                                  "convolution GEMMs, BatchNorm, pooling, projection; its GEMMs are also counted in the gemm families); flops = the "
                                  "images that run (filled slots + one representative of the empty ones) x the per-image count of SURVEY 8d"}
    for name, table in shapes.items():        # the ten shapes that take the most time, per family
        rows = sorted(table.items(), key=lambda kv: -kv[1][1])[:10]
        fam[name]["by_shape"] = [{"MNK": list(k[:3]), "live_rows": k[3], "launches": v[0], "ms": round(v[1], 3),
                                  "tflops": round(v[2] / v[1] / 1e9, 1) if v[1] > 0 else None} for k, v in rows]
    peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
    for f in fam.values():
        f["achieved"] = f["flops"] / f["ms"] / 1e9 if f["ms"] > 0 else None
        f["frac"] = f["achieved"] / peak if f["achieved"] else None
    if "resnet" in fam:
        fam["resnet"]["frac_all_slots"] = fam["resnet"]["flops_all_slots"] / fam["resnet"]["ms"] / 1e9 / peak
    fam["note"] = ("one eager step after the timed region, HIP events around every launch; gemm = every mmsum_gemm launch at its live row "
                   "count (executed FLOPs), attention = every mmsum_attn_fwd / mmsum_attn_bwd launch priced at the canonical count (all "
                   "padded keys, full rectangle under the causal mask, backward = 2.5 x forward)")
    if not dec:
        return None, fam
    M = dec[0][1]
    plan = kn.gemm_plan(torch.empty(M, D, device=e.device, dtype=e.dtype), torch.empty(Fd, D, device=e.device, dtype=e.dtype),
                        torch.empty(M, Fd, device=e.device, dtype=e.dtype), epi=kn.EPI_GELU,
                        bias=torch.empty(Fd, device=e.device), aux=torch.empty(M, Fd, device=e.device, dtype=e.dtype))
    every = dec + enc
    tot_ms = sum(ms for ms, _ in every)
    tot_fl = sum(2.0 * rows * Fd * D for _, rows in every)
    davg = sum(ms for ms, _ in dec) / len(dec)
    out = {"kernel": "gemm_nt_w4_kernel<EPI_GELU,OUT_T> (bf16 NT GEMM x W^T + bias, GELU, pre-activation saved; %dx%d tile, four waves of 128x128, "
                     "64-deep LDS-DMA stages; %d persistent workgroups)" % (plan[1], plan[2], plan[3]) if args.dtype == "bf16" else "gemm_kernel<f32,NT,EPI_GELU>",
           "shape": [M, Fd, D], "launches_timed": len(every), "avg_launch_ms": tot_ms / len(every), "flops_per_launch": tot_fl / len(every),
           "achieved": tot_fl / tot_ms / 1e9,
           "algorithmic_bytes_per_launch": 2.0 * (M * D + Fd * D + 2 * M * Fd) + 4.0 * Fd,
           "measured": "HIP events around each in-step launch (one eager step after the timed region); achieved = FLOPs of all launches / their time",
           "decoder_calls": {"launches_timed": len(dec), "rows": M, "avg_launch_ms": davg, "min_launch_ms": min(ms for ms, _ in dec),
                             "max_launch_ms": max(ms for ms, _ in dec), "achieved": 2.0 * M * Fd * D / davg / 1e9}}
    if enc:
        eavg = sum(ms for ms, _ in enc) / len(enc)
        erows = sum(r for _, r in enc) / len(enc)
        out["encoder_calls"] = {"launches_timed": len(enc), "live_rows": erows, "avg_launch_ms": eavg,
                                "achieved": 2.0 * erows * Fd * D / eavg / 1e9}
    return out, fam


def pmc_traffic(shape):
    """HBM-side bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary of this exact shape
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, profiles/*dominant_gemm_pmc*.json, newest round first)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_dominant_gemm_pmc*.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
            if list(prof["shape"]) == list(shape):
                # a file that records the sha1 of the kernel's sources (tools/pmc_dominant_json.py, round 4 on) is checked against the
                # sources on disk: counters of another build are reported, but named as such
                src = os.path.basename(path)
                want = prof.get("source_sha1")
                if want is None:
                    src += " (kernel sources not recorded in this file)"
                else:
                    import hashlib
                    cur = {f: hashlib.sha1(open(os.path.join(ROOT, "multimodalsum_amd", "csrc", f), "rb").read()).hexdigest() for f in want}
                    if cur != want:
                        src += " (STALE: the kernel's sources changed since these counters were collected)"
                return prof["hbm_bytes_per_launch"], src, prof
        except Exception:
            continue
    return None, None, None


def step_pmc(workload, batch):
    """rocprofv3 counters of the WHOLE step from the newest committed profiles/r*_step_pmc.json of this workload and batch (a separate, serialised
    profiler run -- tools/r5_step_pmc.sh -- not this run: named as such): MFMA pipe busy fraction of all kernels' cycles, per family, and the
    MFMA FLOPs the hardware counted per business (to hold against roofline.step.executed.flops_per_business)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_step_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                prof = json.load(f)
            if prof.get("workload") == workload and int(prof.get("per_gpu_batch", -1)) == int(batch):
                return {"source": os.path.basename(path) + " (a separate rocprofv3 --pmc run of one eager step, kernels serialised; not collected by this run)",
                        "mfma_busy_frac": prof["mfma_busy_frac"], "by_family": {k: v.get("mfma_busy_frac") for k, v in prof.get("by_family", {}).items()},
                        "mfma_flops_per_business": prof.get("mfma_flops_per_business"),
                        # HBM-side bytes of the step from the TCC counters (round 6: assembled from isolated per-family passes -- the whole-step
                        # FETCH_SIZE / WRITE_SIZE passes do not finish on this pool; the file has the method and the representatives)
                        "hbm_traffic": ({k: v for k, v in prof["hbm_traffic"].items() if k not in ("gemm_representatives", "method")}
                                        if isinstance(prof.get("hbm_traffic"), dict) else prof.get("hbm_traffic"))}
        except Exception:
            continue
    return None


# ------------------------------------------------------------------------------------------------
# CPU baseline: whole steps of the restated reference algorithm on the host cores
# ------------------------------------------------------------------------------------------------
def host_cores():
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
    return max(1, cores)


def cpu_baseline(workload, budget_s=75.0, max_steps=3):
    """the reference algorithm (CPU oracle, literal: 9 sequential decoder passes, K/V re-projected per
    pass, unfused loss, PyTorch fp32) on the host cores -- one B=1 step of the SAME configuration as the GPU run (BART-large
    12+12 layers, 9 reviews x 128 tokens, 4 images of 224x224, table): forward + backward + clip_grad_norm_(1.0) + HF-AdamW over
    the decay group (quirk Q1: the no-decay group is empty), i.e. the step the GPU side times.  1 warm-up + up to `max_steps`
    timed steps within `budget_s`.
    Weights are random (uniform, the formula init's spread): the closed-form init costs ~50 s at this size and the timing
    does not depend on the values."""
    import torch
    from multimodalsum_amd import synthetic as syn
    from multimodalsum_amd.config import BartConfig
    from oracle import bart_oracle as bo, encoders_oracle as eo, step_oracle as so
    cfg = BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
    cores = min(host_cores(), 64)
    torch.set_num_threads(cores)
    multimodal = workload == "multimodal"
    L = cfg.encoder_layers
    ocfg = bo.BartCfg(vocab_size=cfg.vocab_size, d_model=cfg.d_model, ffn_dim=cfg.encoder_ffn_dim, encoder_layers=L,
                      decoder_layers=cfg.decoder_layers, heads=cfg.heads, max_position_embeddings=cfg.max_position_embeddings, dropout=0.1)
    shapes = bo.bart_param_shapes(ocfg, multimodal, prefix="bart_model.")
    if multimodal:
        shapes.update(eo.table_param_shapes())
        shapes.update(eo.resnet_param_shapes(cfg.d_model))
    g = torch.Generator().manual_seed(7)
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var") or ((".bn" in k or "layer_norm" in k or "layernorm_embedding" in k or "downsample.1" in k) and k.endswith(".weight")):
            sd[k] = torch.ones(shp)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shp)
        else:
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * (0.05 if "img_encoder" in k else 0.035)
    for k, v in sd.items():
        if v.is_floating_point() and v.dim() > 0 and "running" not in k:
            v.requires_grad_(True)
    I = 4 if multimodal else 1
    b = syn.yelp_batch(1, 9, 128, I, cfg.vocab_size, seed=1234, img_hw=224 if multimodal else 8)

    decay = [k for k, v in sd.items() if v.requires_grad and not any(nd in k for nd in so.NO_DECAY)]     # train_utils.py:49-57 with Q1
    moments = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in decay}
    nstep = [0]

    def step():
        for v in sd.values():
            v.grad = None
        if multimodal:
            loss = so.multimodal_step_loss(sd, ocfg, b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"],
                                           b["field_value"], b["img"], b["img_mask"], 0.1, training=True)
        else:
            loss = so.text_step_loss(sd, ocfg, b["reviews"], b["reviews_mask"], b["reviews_rating"], None, training=True)
        loss.backward()
        nstep[0] += 1
        with torch.no_grad():
            so.clip_grad_norm([v.grad for v in sd.values() if getattr(v, "grad", None) is not None], 1.0)     # multimodal_train.py:361-362
            for k in decay:
                if sd[k].grad is not None:
                    so.adamw_step(sd[k], sd[k].grad, moments[k][0], moments[k][1], nstep[0], 1e-5, weight_decay=0.01)

    t_w = time.time()
    step()                       # warm-up (allocator, thread pool)
    t_w = time.time() - t_w
    times = []
    t0 = time.time()
    while len(times) < max_steps and (not times or (time.time() - t0) + times[-1] < budget_s):
        t1 = time.time()
        step()
        times.append(time.time() - t1)
    dt = sum(times) / len(times)
    D, F, V = cfg.d_model, cfg.encoder_ffn_dim, cfg.vocab_size
    f = flops_per_business(D, F, V, L, L, 9, 128, 128, I, multimodal=multimodal)
    kv = L * 4 * (9 * 128 + (47 + I * 196 if multimodal else 0)) * D * D
    kv_lit = L * 9 * 4 * (8 * 128 + (47 + I * 196 if multimodal else 0)) * D * D
    q_extra = L * 9 * (2 if multimodal else 0) * 2 * 128 * D * D
    literal = f + 3.0 * (kv_lit - kv + q_extra)
    return {"value": 1.0 / dt, "unit": "businesses/s", "cores": cores, "kind": "port",
            "sample": "CPU oracle (PyTorch fp32, the reference algorithm restated literally) on %d host cores: whole training steps (forward, "
                      "backward, clip_grad_norm_ 1.0, HF-AdamW on the Q1 decay group; uniform-random weights of the formula init's spread) of "
                      "the bench configuration at B=1 (BART-large 12+12 layers, 9 reviews x 128 tokens%s): 1 warm-up (%.1f s) + %d timed steps, "
                      "%.1f s each (%.0f GFLOP/s on the reference-literal %.2f TFLOP/step); measured, not extrapolated"
                      % (cores, ", 4 images 224x224 through ResNet101, 47-field table" if multimodal else "", t_w, len(times), dt,
                         literal / dt / 1e9, literal / 1e12),
            "step_seconds": times, "warmup_seconds": t_w}


def cpu_baseline_bounded(args, budget_s=420):
    """Runs cpu_baseline() in a child process (no GPU touched there) under a hard wall-clock budget."""
    code = ("import sys, json; sys.path.insert(0, %r); import bench; print('CPUBASE ' + json.dumps(bench.cpu_baseline(%r)))"
            % (ROOT, args.workload))
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


def run_generate(model, cfg, device, B, steps, warmup, dtype_name):
    """BASELINE config 5 (test.py:137-164): B businesses (test.py:176: 8) x 8 reviews x 128 tokens + table + 4 images through the three
    encoders, then 4-beam search, max_length 128, no_repeat_ngram_size 3, early_stopping; random-init weights (they never emit EOS
    early, so every summary runs to max_length: the worst case).  One 'step' = one generate() call of the batch."""
    import torch
    from multimodalsum_amd import synthetic as syn
    was_training = model.training
    model.eval()
    max_length, beams = 128, 4
    b = syn.batch_to(syn.yelp_batch(B, 8, 128, 4, cfg.vocab_size, seed=7, img_hw=224), device)

    def run():
        with torch.no_grad():
            _, th, tm, tabh, tabm, ih, im = model.get_multimodal_outputs(b["reviews"], b["reviews_mask"], b["field"], b["field_value"], b["img"], b["img_mask"])
            rd = torch.zeros(B, 1, device=device)                                  # test.py:155
            return model.bart_model.generate(th, tm, tabh, tabm, ih, im, rating_diff=rd, num_beams=beams, length_penalty=1.0, max_length=max_length,
                                             no_repeat_ngram_size=3, early_stopping=True)
    for _ in range(max(1, warmup)):       # the first call captures the per-position decode graphs
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    model.train(was_training)
    nsteps = out.shape[1] - 1
    # algorithmic HBM bytes of ONE decode step: the decoder's weights once (per layer 3 + 1 self-attention, 1 + 1
    # cross-attention q / out, 2 + 2 alpha / beta, 4 + 4 FFN = 18 D^2; + the tied LM head V D), the cached cross-attention K / V of the
    # un-expanded memory (B businesses x (8 x 128 + 47 + 4 x 196) rows x 2 D per layer), the self-attention caches of the rows so far
    # (mean length max_length / 2); bf16 = 2 bytes, f32 = 4
    es = 2 if dtype_name == "bf16" else 4
    D, Ld, V = cfg.d_model, cfg.decoder_layers, cfg.vocab_size
    step_bytes = es * (Ld * 18 * D * D + V * D + Ld * B * (8 * 128 + 47 + 4 * 196) * 2 * D + Ld * B * beams * (max_length // 2) * 2 * D)
    step_s = dt / max(nsteps, 1)
    return {"metric": "generated summaries/sec (4-beam search, max_length 128) BART-large multimodal", "value": B / dt, "unit": "summaries/s",
            "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic Yelp-shaped test batch (seeded), formula-initialised weights",
            "config": {"workload": "test.py beam-search generation (beam=4, max_len=128, no_repeat_ngram_size=3, early_stopping) full multimodal: "
                                   "8 reviews x 128 tok + table + 4 images per business", "per_gpu_batch": B, "num_beams": beams},
            "decode_steps": nsteps, "ms_per_decode_step": dt * 1e3 / max(nsteps, 1), "tokens_per_s": B * nsteps / dt,
            "decode_step_bytes": step_bytes, "decode_hbm_frac": step_bytes / step_s / 8e12}


def bench_generate(args):
    import torch
    import multimodalsum_amd as mm
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    device = torch.device("cuda", 0)
    cfg = mm.BartConfig.from_json_file(os.path.join(ROOT, "cfg", "bart-large.json"))
    model = mm.MultimodalSum(config=cfg, label_smoothing=0.1, device=device, dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    out = run_generate(model, cfg, device, args.batch, args.steps, args.warmup, args.dtype)
    out["peak_hbm_gb"] = round(torch.cuda.max_memory_reserved() / 2**30, 1)
    print(json.dumps(out), flush=True)


def timed_steps(args, model, runner, opt, sch, next_batch, steps, warmup, sync):
    """`warmup` un-timed and `steps` timed training steps (after two priming steps when the step graphs are on: one eager, one
    that captures).  Returns (seconds of the timed region, last loss, the timed steps' device-side live counts, priming steps)."""
    priming = 0
    if not args.no_graphs:
        model.enable_step_graphs()
        priming = 2
        for _ in range(priming):
            run_step(args, runner, opt, sch, next_batch())
    for _ in range(warmup):
        run_step(args, runner, opt, sch, next_batch())
    live_rows = []
    import torch
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]     # per-step boundaries on the step's launch stream
    sync()
    w0 = time.time()
    t0 = time.perf_counter()
    for i in range(steps):
        marks[i].record()
        b = next_batch()                   # generated on the device inside the timed step
        loss = run_step(args, runner, opt, sch, b)
        live_rows.append((b["reviews_mask"].sum(), b["img_mask"].sum() if args.workload in ("multimodal", "text_table") else None))   # device scalars, read after the region
    marks[steps].record()
    sync()
    dt = time.perf_counter() - t0
    args.timed_window = (w0, time.time())          # wall-clock bounds of the timed region (the telemetry child's samples are cut to it)
    args.step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]     # device time of every timed step (HIP events)
    return dt, loss, live_rows, priming, b


def step_percentiles(step_ms):
    """p10 / p50 / p90 / min / max of the per-step device times (nearest-rank on the sorted list)."""
    v = sorted(step_ms)
    n = len(v)

    def q(f):
        return v[min(n - 1, max(0, int(round(f * (n - 1)))))]
    return {"ms_per_step_p10": q(0.1), "ms_per_step_p50": q(0.5), "ms_per_step_p90": q(0.9), "ms_per_step_min": v[0], "ms_per_step_max": v[-1],
            "per_step_timing": "HIP event at every step boundary on the launch stream, %d timed steps" % n}


def also_configs(args, cfg, model, device):
    """The other BASELINE configurations, measured in the same run so that the one driver-run line carries them (a few steps each):
    the reference-style per-GPU batch of 8 (multimodal_train.py:420 default is 1 per GPU; SURVEY.md 8d C4: {1, 8}) and test.py's
    beam-search generation on the headline's model; the text-only step (text_pretrain.py:66-113, BASELINE config 2) follows on a
    TextSupervised model built after this one is released (also_text_only)."""
    import copy
    import gc
    import torch
    from multimodalsum_amd import optim
    out = {}
    object.__setattr__(model, "_step_graphs", None)        # the headline's captured graph set pins its activations: release it first
    gc.collect()
    torch.cuda.empty_cache()

    def train_cfg(name, workload, batch, mdl, steps=4, warmup=1):
        a = copy.copy(args)
        a.workload, a.batch = workload, batch
        opt = optim.get_optimizer(1e-5, NO_DECAY, mdl.named_parameters(), None)
        sch = optim.get_linear_schedule_with_warmup(opt, 100, 100000)
        dt, loss, _, _, _ = timed_steps(a, mdl, mdl, opt, sch, batch_source(a, cfg, device, 0), steps, warmup, torch.cuda.synchronize)
        out[name] = {"value": batch * steps / dt, "unit": "businesses/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
                     "per_gpu_batch": batch, "final_loss": float(loss.item()),
                     "workload": {"multimodal": "multimodal_train.py full step", "text": "text_pretrain.py text-only step",
                                  "text_table": "multimodal_train.py text + table step (BASELINE config 3: one all-zero image slot per business, "
                                                "img_mask False everywhere; the ResNet still runs on it, as in the reference)"}[workload]
                                 + " (fwd+bwd+clip+AdamW), 9 reviews x 128 tok, hip-graph replay"}
        if workload == "text_table":
            fpb = flops_per_business(cfg.d_model, cfg.encoder_ffn_dim, cfg.vocab_size, cfg.encoder_layers, cfg.decoder_layers, 9, 128, 128, 1)
            out[name]["step_roofline_frac"] = out[name]["value"] * fpb / 1e12 / PEAK_BF16_TFLOPS
    try:
        if args.batch != 56:
            train_cfg("multimodal_B56", "multimodal", 56, model)     # the batch of rounds 1-2's headline, for continuity
            object.__setattr__(model, "_step_graphs", None)
            gc.collect()
            torch.cuda.empty_cache()
        fpb4 = flops_per_business(cfg.d_model, cfg.encoder_ffn_dim, cfg.vocab_size, cfg.encoder_layers, cfg.decoder_layers, 9, 128, 128, 4)
        train_cfg("multimodal_B8", "multimodal", 8, model)
        out["multimodal_B8"]["step_roofline_frac"] = out["multimodal_B8"]["value"] * fpb4 / 1e12 / PEAK_BF16_TFLOPS
        object.__setattr__(model, "_step_graphs", None)
        gc.collect()
        torch.cuda.empty_cache()
        # the reference's own default: one business per GPU and step (multimodal_train.py:420, --batch_size 1)
        train_cfg("multimodal_B1", "multimodal", 1, model, steps=8, warmup=2)
        out["multimodal_B1"]["step_roofline_frac"] = out["multimodal_B1"]["value"] * fpb4 / 1e12 / PEAK_BF16_TFLOPS
        g = run_generate(model, cfg, device, 8, 2, 1, args.dtype)
        out["generate_B8"] = {k: g[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "decode_steps", "ms_per_decode_step", "tokens_per_s",
                                                "decode_step_bytes", "decode_hbm_frac")}
        out["generate_B8"]["workload"] = g["config"]["workload"]
        object.__setattr__(model, "_step_graphs", None)
        gc.collect()
        torch.cuda.empty_cache()
        train_cfg("text_table_B%d" % args.batch, "text_table", args.batch, model)     # BASELINE config 3 at the headline batch
    except Exception as exc:                   # the headline number must survive a failure here
        out["error_multimodal"] = repr(exc)[:300]
    return out


def also_generate_f32(cfg, device):
    """BASELINE config 5 in the f32 compute mode -- on the committed fixture (tests/golden/g2_generate_full.npz: the widest-margin one of six input seeds, smallest decision gap 5.5e-4 nats) its token ids equal the reference's own generate() at this size (12 + 12 layers, 4 beams,
    max_length 128: tests/test_generation_gpu.py::test_generation_ids_equal_the_reference_at_config5_size): f32 weights and kernels
    (weight-streaming v_mfma_f32_16x16x4_f32 products, f32 decode attention), the same 8 businesses x 4 beams, beside the bf16 line."""
    import torch
    import multimodalsum_amd as mm
    try:
        mdl = mm.MultimodalSum(config=cfg, label_smoothing=0.1, device=device, dtype=torch.float32, deterministic=True)
        g = run_generate(mdl, cfg, device, 8, 1, 1, "f32")
        out = {k: g[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "decode_steps", "ms_per_decode_step", "tokens_per_s")}
        out["workload"] = g["config"]["workload"] + " -- f32 compute mode (deterministic kernels): the mode whose ids equal the reference's generate() on the committed fixture of this size (decision margin 5.5e-4 nats)"
        return out
    except Exception as exc:
        return {"error": repr(exc)[:300]}


def also_text_only(args, cfg, device):
    """BASELINE config 2: BART-large text-only (8 source reviews x 128 tok per pass) bf16 on one GPU, at the headline batch."""
    import copy
    import torch
    from multimodalsum_amd import optim
    try:
        a = copy.copy(args)
        a.workload = "text"                          # at the headline's batch
        _, mdl = build(a, device)
        opt = optim.get_optimizer(1e-5, NO_DECAY, mdl.named_parameters(), None)
        sch = optim.get_linear_schedule_with_warmup(opt, 100, 100000)
        steps, warmup = 4, 1
        dt, loss, _, _, _ = timed_steps(a, mdl, mdl, opt, sch, batch_source(a, cfg, device, 0), steps, warmup, torch.cuda.synchronize)
        fpb = flops_per_business(cfg.d_model, cfg.encoder_ffn_dim, cfg.vocab_size, cfg.encoder_layers, cfg.decoder_layers, 9, 128, 128, 1, multimodal=False)
        val = a.batch * steps / dt
        return {"value": val, "unit": "businesses/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "per_gpu_batch": a.batch,
                "final_loss": float(loss.item()), "step_roofline_frac": val * fpb / 1e12 / PEAK_BF16_TFLOPS,
                "workload": "text_pretrain.py BART-large text-only step (fwd+bwd+clip+AdamW), 9 reviews x 128 tok, hip-graph replay"}
    except Exception as exc:
        return {"error": repr(exc)[:300]}


def main():
    args = parse()
    if args.workload == "generate":
        return bench_generate(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)                       # does not return
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    tele_proc, tele_path = start_telemetry() if rank == 0 else (None, None)      # a child process, before this one touches a GPU
    # stdout carries the ONE JSON line and nothing else: while the ranks run, file descriptor 1 points at stderr (RCCL writes its
    # library banner to the C stdout at init and at teardown); it is put back for the final print
    saved_stdout = None
    if world > 1 or os.environ.get("MMSUM_FORCE_DDP") == "1":
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_ddp = os.environ.get("MMSUM_FORCE_DDP") == "1"          # debugging aid: run the RCCL gradient path at world size 1
    dist = None
    if world > 1 or force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
    import multimodalsum_amd as mm
    from multimodalsum_amd import optim
    cfg, model = build(args, device)
    ddp = None
    if world > 1 or force_ddp:
        ddp = mm.DistributedDataParallel(model, delay_allreduce=True, always_reduce=force_ddp, collect_stats=True, mode=args.ddp_mode,
                                         grad_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else None)
    runner = ddp if ddp is not None else model
    opt = optim.get_optimizer(1e-5, NO_DECAY, model.named_parameters(), None)
    sch = optim.get_linear_schedule_with_warmup(opt, 100, 100000)
    next_batch = batch_source(args, cfg, device, rank)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    stats_skip = None
    dt, loss, live_rows, priming, b = timed_steps(args, model, runner, opt, sch, next_batch, args.steps, args.warmup, sync)
    if ddp is not None:
        stats_skip = max(0, len(ddp.stats) - args.steps)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())
    telemetry = None
    if rank == 0:
        bdf = None
        try:
            pr = torch.cuda.get_device_properties(device)
            bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        telemetry = read_telemetry(tele_proc, tele_path, *args.timed_window, bdf=bdf)
    live_rows = [(float(t), float(im) if im is not None else 0.0) for t, im in live_rows]      # the device scalars, now that the region is over
    comm = ddp.comm_stats(skip=stats_skip) if ddp is not None else None
    if world > 1:                       # every rank: the exchange's two forms on the wrapper's bucket sizes, outside the step
        try:
            from multimodalsum_amd.parallel import bus_microbench
            mb = bus_microbench(device, iters=3)
            if comm is not None:
                comm["microbench"] = mb
                comm["xgmi_links"] = "7 links x ~153 GB/s per GPU (point-to-point)"
        except Exception as exc:
            sys.stderr.write("bench.py: bus microbench failed: %r\n" % (exc,))
    graphs = getattr(model, "_step_graphs", None)
    captures = graphs.captures if graphs is not None else 0
    peak_gb = round(torch.cuda.max_memory_reserved() / 2**30, 1)
    probe, families, also = None, None, None
    if rank == 0 and world == 1 and not args.no_graphs:
        # the timed region is over: hand the captured graph set's activations (about half of the HBM at the default batch) back
        # before anything else allocates -- the probe's eager step needs as much again
        import gc
        object.__setattr__(model, "_step_graphs", None)
        graphs = loss = None                  # (the loss tensor's autograd node holds the captured set too)
        gc.collect()
        torch.cuda.empty_cache()
    try:
        if rank == 0 and not args.no_kernel_probe and world == 1:
            probe, families = probe_step_kernels(args, model, runner, opt, sch, b, cfg)
    except Exception as exc:                   # (an out-of-memory here must not cost the headline number)
        sys.stderr.write("bench.py: kernel probe failed: %r\n" % (exc,))
    if rank == 0 and world == 1 and not args.no_also and args.workload == "multimodal" and args.dtype == "bf16" and not args.no_graphs:
        try:
            del runner, opt, sch, next_batch, b
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            also = also_configs(args, cfg, model, device)
            object.__setattr__(model, "_step_graphs", None)        # release the headline model before the text-only one is built
            model = None
            gc.collect()
            torch.cuda.empty_cache()
            also["text_only_B%d" % args.batch] = also_text_only(args, cfg, device)
            gc.collect()
            torch.cuda.empty_cache()
            also["generate_B8_f32"] = also_generate_f32(cfg, device)
        except Exception as exc:
            sys.stderr.write("bench.py: extra configurations failed: %r\n" % (exc,))
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * args.steps / dt
        multimodal = args.workload in ("multimodal", "text_table")
        I = 4 if args.workload == "multimodal" else 1
        dims = (cfg.d_model, cfg.encoder_ffn_dim, cfg.vocab_size, cfg.encoder_layers, cfg.decoder_layers, 9, 128, 128, I)
        fpb = flops_per_business(*dims, multimodal=multimodal)
        text_rows = sum(t for t, _ in live_rows) / len(live_rows) / args.batch
        filled = (sum(im for _, im in live_rows) / len(live_rows) / args.batch) if multimodal else 0.0          # filled image slots per business
        mem_rows = text_rows + ((47 + 196 * filled) if multimodal else 0)
        # images the branch runs per business: the filled slots + ONE representative per batch of its empty slots (engine.img_fwd's live-image window)
        windowed = multimodal and os.environ.get("MMSUM_IMAGE_DEDUPE") != "0"
        # (a batch runs ONE extra image -- the representative -- only when it has an empty slot at all)
        img_run = (sum(im + (1 if im < args.batch * I else 0) for _, im in live_rows) / len(live_rows) / args.batch) if windowed else None
        fpb_exec = flops_per_business(*dims, multimodal=multimodal, enc_rows=text_rows, mem_rows=mem_rows, img_run=img_run)
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        achieved = value / world * fpb / 1e12
        achieved_exec = value / world * fpb_exec / 1e12
        # the step's PRIMARY figure counts the FLOPs that are executed; `canonical` is SURVEY.md 8d's padded count (every padded encoder row, every
        # masked memory row and all I image slots per business, the skipped ones included), kept for comparison with earlier rounds
        step_roof = {"achieved": achieved_exec, "frac": achieved_exec / peak, "flops_per_business": fpb_exec,
                     "executed": {"achieved": achieved_exec, "frac": achieved_exec / peak,
                                  "flops_per_business": fpb_exec, "encoder_rows_per_business": text_rows, "memory_rows_per_business": mem_rows,
                                  "images_run_per_business": img_run,
                                  "note": "FLOPs of the rows the padding-free encoder layers / K-V projections really process and of the images the "
                                          "image branch really runs (filled slots + one representative of the empty ones); results identical"},
                     "canonical": {"achieved": achieved, "frac": achieved / peak, "flops_per_business": fpb,
                                   "note": "SURVEY.md 8d's algorithmic count on padded rows and all image slots: includes work the step skips"},
                     "scope": "whole training step per GPU: executed FLOPs / step time (canonical = the padded count of SURVEY.md 8d)"}
        if probe is None:
            roof = {"bound": "mfma", "achieved": achieved_exec, "peak": peak, "unit": "TFLOP/s", "frac": achieved_exec / peak, "traffic": None,
                    "scope": step_roof["scope"], "flops_per_business": fpb_exec, "executed": step_roof["executed"], "canonical": step_roof["canonical"]}
        else:
            traffic, src, pmc = pmc_traffic(probe["shape"])
            roof = {"bound": "mfma", "achieved": probe["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": probe["achieved"] / peak,
                    "traffic": traffic, "traffic_source": src, "step": step_roof}
            if pmc is not None:
                # rocprof-reported utilisation of the dominant kernel (north star): MFMA pipe busy cycles per SIMD (1,024 SIMDs) over the
                # kernel's cycles per XCD (8 XCDs), both from the same PMC pass; HBM rate = PMC bytes per launch / the launch time
                # measured live in this run
                try:
                    roof["mfma_busy_frac"] = (pmc["sq"]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (pmc["sq"]["GRBM_GUI_ACTIVE"] / 8.0)
                    roof["hbm_gbps"] = traffic / (probe["decoder_calls"]["avg_launch_ms"] * 1e-3) / 1e9     # the PMC shape = the decoder launches (all rows)
                    roof["hbm_frac_of_8tbps"] = roof["hbm_gbps"] / 8000.0
                except Exception:
                    pass
            roof.update({k: v for k, v in probe.items() if k != "achieved"})
        if families is not None:
            roof["families"] = families
        sp = step_pmc(args.workload, args.batch)
        if sp is not None and "step" in roof:
            roof["step"]["pmc"] = sp
        out = {"metric": "training samples/sec (businesses/sec) BART-large multimodal" if multimodal else
               "training samples/sec (businesses/sec) BART-large text-only",
               "value": value, "unit": "businesses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
               "data": "synthetic Yelp-shaped batches generated on the device inside every timed step (fresh review lengths and image counts "
                       "per step), formula-initialised BART-large/ResNet101 weights" if not (args.fixed_batches or getattr(args, "host_batches", False)) else
                       ("two synthetic Yelp-shaped batches in pinned HOST memory, copied to the HBM every step by yelp_data_prefetcher (%.0f MB per step "
                        "over PCIe, on a side stream under the previous step): the PCIe-INCLUSIVE rate, not the headline value" % (args.host_bytes_per_step / 1e6))
                       if getattr(args, "host_batches", False) else
                       "two fixed synthetic Yelp-shaped batches (seeded), formula-initialised BART-large/ResNet101 weights",
               "config": {"workload": {"multimodal": "multimodal_train.py full text+img(4x224^2)+table step (fwd+bwd+clip+AdamW), 9 reviews x 128 tok",
                                       "text_table": "multimodal_train.py text+table step (one all-zero image slot per business, img_mask False), 9 reviews x 128 tok",
                                       "text": "text_pretrain.py BART-large text-only step, 9 reviews x 128 tok"}[args.workload],
                          "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "dropout": cfg.dropout},
               "launch": "eager" if args.no_graphs else "hip-graph replay (1 forward graph + 1 graph per backward gradient segment, ONE set for all "
                         "batches: row counts are device-side), %d priming steps before warmup, %d capture(s) in the whole run" % (priming, captures),
               "graph_captures": captures, "final_loss": loss_val, "peak_hbm_gb": peak_gb, "roofline": roof}
        if getattr(args, "step_ms", None):
            out.update(step_percentiles(args.step_ms))
        out["runtime_env"] = {k: os.environ.get(k) for k in ("HIP_FORCE_DEV_KERNARG", "MMSUM_SIDE_STREAM", "MMSUM_IMAGE_DEDUPE", "MMSUM_IMPLICIT_CONV") if os.environ.get(k) is not None}
        if telemetry is not None:
            out["telemetry"] = telemetry
        if args.diag_stub_resnet:
            out["diagnostic"] = "INVALID as a benchmark: ResNet forward / backward stubbed (--diag-stub-resnet)"
        if comm is not None:
            out["comm"] = comm
        if also is not None:
            out["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_bounded(args)
    if dist is not None:
        dist.destroy_process_group()          # RCCL prints its library banner on teardown: keep the JSON line last
    sys.stdout.flush()
    try:                                       # RCCL's banner sits in the C stdio buffer until exit: push it out (to stderr) first
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if saved_stdout is not None:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

"""CPU, world_size 2, gloo: the data-parallel path (parameter broadcast, segment-wise gradient all-reduce
over the flat arena, mean over ranks) through the real DistributedDataParallel wrapper, with the kernel
emulator standing in for the HIP library.  rank r trains on batch r; the averaged gradients must equal
the mean of the two single-process gradients."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dtype=torch.float32, layers=1):
    from tests import cpu_kernel_emu as emu
    import multimodalsum_amd.engine as eng
    import multimodalsum_amd.modules as mods
    import multimodalsum_amd.optim as opt
    for m in (eng, mods, opt):
        m.kn = emu
    from multimodalsum_amd.modules import TextSupervised
    from multimodalsum_amd.formula_init import formula_state_dict
    from oracle import bart_oracle as bo
    from tests.test_host_logic_cpu import tiny_cfg, oracle_cfg
    cfg = tiny_cfg(vocab=60, d=256, ffn=64, layers=layers, heads=4, maxpos=40)
    sd = formula_state_dict(bo.bart_param_shapes(oracle_cfg(cfg), False, prefix="bart_model."), std=0.08)
    model = TextSupervised(config=cfg, label_smoothing=0.1, device="cpu", dtype=dtype)
    model.load_state_dict(sd)
    model.train()
    return cfg, model


def _batch(cfg, rank):
    from multimodalsum_amd import synthetic as syn
    return syn.yelp_batch(2, 3, 16, 1, cfg.vocab_size, seed=70 + rank, img_hw=8)


# (name, DistributedDataParallel keyword arguments, model layers, layers per gradient segment)
CASES = [("allreduce_f32", dict(), 1, 3),
         ("rsag_f32_segments", dict(mode="reduce_scatter", bucket_elems=100003), 4, 2),       # odd bucket size: shards + a remainder
         ("allreduce_bf16", dict(grad_dtype=torch.bfloat16), 1, 3),
         ("rsag_bf16", dict(mode="reduce_scatter", grad_dtype=torch.bfloat16, bucket_elems=65536), 1, 3)]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from multimodalsum_amd.parallel import DistributedDataParallel, reduce_tensor
    for name, kw, layers, per in CASES:
        cfg, model = _build(layers=layers)
        model.grad_segment_layers = per
        if rank == 1:       # perturb rank 1: the wrapper must broadcast rank 0's parameters
            with torch.no_grad():
                model._engine.arena.data.add_(0.5)
        ddp = DistributedDataParallel(model, delay_allreduce=True, **kw)
        seen = []
        model._engine.segment_hooks.insert(0, lambda prefixes: seen.append(list(prefixes)))
        b = _batch(cfg, rank)
        loss = ddp(b["reviews"], b["reviews_mask"], b["reviews_rating"])[0]
        loss.backward()
        mean_loss = reduce_tensor(loss.detach().reshape(1), world)
        torch.save({"grad": model._engine.arena.grad.clone(), "data": model._engine.arena.data.clone(), "loss": mean_loss, "segments": seen,
                    "has_grad": [n for n, p in model.named_parameters() if p.grad is not None]}, os.path.join(out_dir, "%s_r%d.pt" % (name, rank)))
    from multimodalsum_amd.parallel import bus_microbench          # the table bench.py --gpus N attaches as comm.microbench
    mb = bus_microbench("cpu", sizes_elems=(4099,), dtypes=(torch.float32, torch.bfloat16), iters=1)
    if rank == 0:
        torch.save(mb, os.path.join(out_dir, "microbench.pt"))
    dist.destroy_process_group()


def test_ddp_two_ranks_gloo(tmp_path):
    """Every exchange mode of the wrapper at world size 2: one all-reduce per bucket and reduce-scatter + all-gather (bucket sizes
    that do not divide by the world size included), f32 and bf16 buckets, one and several gradient segments per stack."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    mb = torch.load(tmp_path / "microbench.pt")
    assert len(mb) == 2 and all(r["elements"] == 4098 and r["all_reduce_ms"] > 0 and r["reduce_scatter_all_gather_bus_gb_s"] > 0 for r in mb)
    for name, kw, layers, per in CASES:
        r0, r1 = torch.load(tmp_path / ("%s_r0.pt" % name)), torch.load(tmp_path / ("%s_r1.pt" % name))
        assert torch.equal(r0["data"], r1["data"]), "parameters were not broadcast from rank 0"
        assert torch.equal(r0["grad"], r1["grad"]), (name, "ranks disagree on the reduced gradient")
        assert r0["has_grad"] == r1["has_grad"] and len(r0["has_grad"]) > 30
        cfg, model = _build(layers=layers)
        grads, losses = [], []
        for rank in range(2):
            b = _batch(cfg, rank)
            for p in model.parameters():
                p.grad = None
            loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"])[0]
            loss.backward()
            grads.append(model._engine.arena.grad.clone())
            losses.append(loss.detach())
        bf16 = kw.get("grad_dtype") == torch.bfloat16
        if bf16:        # each rank rounds its bucket to bf16, the sum is taken in bf16, the mean is written back to the f32 arena
            ref = (grads[0].bfloat16() + grads[1].bfloat16()).float() / 2
            tol = 2.0 ** -7 * ref.abs().max().item()
        else:
            ref = (grads[0] + grads[1]) / 2
            tol = 1e-6 + 1e-5 * ref.abs().max().item()
        err = (r0["grad"] - ref).abs().max().item()
        assert err <= tol, (name, err, tol)
        assert abs(r0["loss"].item() - (losses[0] + losses[1]).item() / 2) < 1e-6
        # segments: top layers first, `per` layers each; the tied embedding only with the encoder's bottom layers (last)
        segs = r0["segments"]
        want = 2 * len(range(0, layers, per)) if per < layers else 2
        assert len(segs) == want, (name, segs)
        assert any("model.shared." in px for px in segs[-1]) and not any("model.shared." in px for sg in segs[:-1] for px in sg)
        if name == "rsag_f32_segments":
            assert segs[0] == ["bart_model.model.decoder.layers.2.", "bart_model.model.decoder.layers.3."], segs[0]
            assert segs[2] == ["bart_model.model.encoder.layers.2.", "bart_model.model.encoder.layers.3."], segs[2]

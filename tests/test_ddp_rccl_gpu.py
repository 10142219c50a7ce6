"""GPU, RCCL: the data-parallel path on real devices -- the twin of tests/test_ddp_gloo_cpu.py.

Starts one rank per visible GPU with torch.distributed.run (as README.md:140 of the reference starts its 8 ranks and as
bench.py --gpus N does), backend "nccl" (= RCCL over xGMI).  Each rank trains on its own batch through
multimodalsum_amd.DistributedDataParallel: parameter broadcast from rank 0 (apex DDP's constructor, multimodal_train.py:474),
segment-wise all-reduce on the comm stream overlapping the remaining backward, gradient MEAN over ranks.  The reduced
gradients must be identical on every rank and equal the mean of the single-process gradients of the same batches.
On a 1-GPU box the launch has one rank: the same code path through RCCL at world size 1 (always_reduce)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(n, out_dir, dtype, graphs, port, mode="all_reduce", wire="f32", bucket=1 << 20):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_rccl_worker.py"), str(out_dir), dtype, "1" if graphs else "0",
           mode, wire, str(bucket)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PYTHONPATH=ROOT)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, "ranks failed:\n" + r.stdout[-2000:] + "\n" + r.stderr[-4000:]


@pytest.mark.parametrize("dtype,graphs,mode,wire,bucket", [
    ("f32", False, "all_reduce", "f32", 1 << 20), ("bf16", True, "all_reduce", "f32", 1 << 20),
    # the reduce-scatter + all-gather form through RCCL (ReduceOp.AVG into the persistent shard buffer), with a bucket size that no
    # world size divides (1,000,003 elements: the remainder goes through all_reduce) and, once, bf16 buckets on the wire
    ("f32", False, "reduce_scatter", "f32", 1000003), ("bf16", True, "reduce_scatter", "bf16", 1000003)])
def test_ddp_rccl_gradient_mean(tmp_path, dtype, graphs, mode, wire, bucket):
    n = torch.cuda.device_count()
    assert n >= 1
    _launch(n, tmp_path, dtype, graphs, 29611 + (1 if graphs else 0) + (2 if mode == "reduce_scatter" else 0), mode, wire, bucket)
    ranks = [torch.load(tmp_path / ("r%d.pt" % r), weights_only=False) for r in range(n)]
    for r in ranks[1:]:
        assert torch.equal(ranks[0]["data"], r["data"]), "parameters were not broadcast from rank 0"
        assert torch.equal(ranks[0]["grad"], r["grad"]), "ranks disagree on the reduced gradient"
        assert ranks[0]["has_grad"] == r["has_grad"]
    assert len(ranks[0]["has_grad"]) > 100
    st = ranks[0]["stats"]
    assert st is not None and st["buckets_per_step"] >= 3 and st["bytes_per_step"] > 0     # decoder | encoders upper | lower + embeddings
    if graphs:
        assert ranks[0]["captures"] == 1
    # single-process reference: mean over the ranks' batches of the un-wrapped model's gradients
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_rccl_worker as w
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    cfg, model = w.build(tdt, "cuda")
    grads, losses = [], []
    for r in range(n):
        for p in model.parameters():
            p.grad = None
        loss = w.step(model, w.batch(cfg, r, "cuda"))
        torch.cuda.synchronize()
        grads.append(model._engine.arena.grad.double().cpu())
        losses.append(float(loss))
    ref = sum(grads) / n
    err = (ranks[0]["grad"].double() - ref).abs().max().item()
    tol = (1e-6 + 1e-5 * ref.abs().max().item()) if dtype == "f32" else (1e-5 + 2e-3 * ref.abs().max().item())
    if wire == "bf16":
        tol += 2.0 ** -8 * ref.abs().max().item()              # the buckets are rounded to bf16 for the exchange
    assert err <= tol, (err, ref.abs().max().item())
    assert abs(float(ranks[0]["loss"]) - sum(losses) / n) <= 1e-5 + 1e-3 * abs(sum(losses) / n)


def test_bench_gpus_flag_starts_the_ranks(tmp_path):
    """`python bench.py --gpus N` (no torch.distributed.run around it) starts N ranks itself and reports n_gpus = N: run at N =
    the number of visible GPUs (1 on a single-GPU box, where it exercises the WORLD_SIZE check and the single-process path;
    the driver's 8-GPU node gets 8 RCCL ranks and the comm block)."""
    import json
    n = torch.cuda.device_count()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "2", "--no-cpu-baseline",
           "--master-port", "29631"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == n and out["config"]["global_batch"] == 2 * n and out["graph_captures"] == 1
    if n > 1:
        assert out["comm"]["allreduce_ms"] > 0 and out["comm"]["bus_gb_s"] > 0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", PYTHONPATH=ROOT))
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stdout + bad.stderr)

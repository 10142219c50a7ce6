"""Worker of tests/test_ddp_rccl_gpu.py: one rank of a torch.distributed.run launch (backend "nccl" = RCCL).
Rank r trains one fused multimodal step on batch r through DistributedDataParallel (HIP graphs on: forward graph,
per-segment backward graphs with the all-reduce of a finished segment overlapping the next one) and saves its reduced
gradient arena, parameters and comm statistics; the test compares with the mean of single-process gradients."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(dtype, device):
    from multimodalsum_amd.modules import MultimodalSum
    from tests.test_host_logic_cpu import tiny_cfg, oracle_cfg, f3_state
    cfg = tiny_cfg(vocab=200, d=1024, ffn=128, layers=4, heads=16, maxpos=40)
    model = MultimodalSum(config=cfg, label_smoothing=0.1, device=device, dtype=dtype, deterministic=True)
    model.load_state_dict(f3_state(oracle_cfg(cfg)))
    model.train()
    return cfg, model


def batch(cfg, rank, device):
    from multimodalsum_amd import synthetic as syn
    return syn.batch_to(syn.yelp_batch(2, 3, 32, 2, cfg.vocab_size, seed=90 + rank, img_hw=64), device)


def step(model, b):
    loss = model(b["reviews"], b["reviews_mask"], b["reviews_rating"], b["field"], b["field_value"], b["img"], b["img_mask"])[0]
    loss.backward()
    return loss


def main():
    out_dir, dtype_name, graphs = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
    mode = sys.argv[4] if len(sys.argv) > 4 else "all_reduce"
    wire = torch.bfloat16 if (len(sys.argv) > 5 and sys.argv[5] == "bf16") else None
    bucket = int(sys.argv[6]) if len(sys.argv) > 6 else 1 << 20
    dtype = torch.float32 if dtype_name == "f32" else torch.bfloat16
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
    cfg, model = build(dtype, device)
    if rank != 0:       # perturb the other ranks: the wrapper must broadcast rank 0's parameters
        with torch.no_grad():
            model._engine.arena.data.add_(0.25)
    from multimodalsum_amd.parallel import DistributedDataParallel, reduce_tensor
    ddp = DistributedDataParallel(model, delay_allreduce=True, always_reduce=True, collect_stats=True, bucket_elems=bucket, mode=mode,
                                  grad_dtype=wire)
    if graphs:
        model.enable_step_graphs()
    b = batch(cfg, rank, device)
    reps = 3 if graphs else 1                   # eager warm-up, capture, replay
    for _ in range(reps):
        for p in model.parameters():
            p.grad = None
        loss = step(ddp, b)
        torch.cuda.synchronize()
    mean_loss = reduce_tensor(loss.detach().reshape(1), world)
    torch.save({"grad": model._engine.arena.grad.cpu(), "data": model._engine.arena.data.cpu(), "loss": mean_loss.cpu(),
                "stats": ddp.comm_stats(skip=reps - 1), "has_grad": [n for n, p in model.named_parameters() if p.grad is not None],
                "captures": model._step_graphs.captures if graphs else 0},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

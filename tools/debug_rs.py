import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
g = torch.randn(5_000_011, device=dev)
ref = g.clone()
shard = torch.empty(1_000_003, device=dev)
for op in (dist.ReduceOp.AVG, dist.ReduceOp.SUM):
    g.copy_(ref)
    for b0 in range(0, g.numel(), 1_000_003):
        chunk = g[b0:min(g.numel(), b0 + 1_000_003)]
        n = chunk.numel()
        dist.reduce_scatter_tensor(shard[:n], chunk, op=op)
        mid = (shard[:n] - ref[b0:b0 + n]).abs().max().item()
        dist.all_gather_into_tensor(chunk, shard[:n])
        print(op, b0, "after reduce_scatter: shard err", mid, "after all_gather: err", (chunk - ref[b0:b0 + n]).abs().max().item(), flush=True)
    # in place (the round-3 form)
    g.copy_(ref)
    chunk = g[:1_000_000]
    dist.reduce_scatter_tensor(chunk.view(1, -1)[0], chunk, op=op)
    print(op, "in place rs err", (chunk - ref[:1_000_000]).abs().max().item())
dist.destroy_process_group()

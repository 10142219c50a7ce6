import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
N = 3_000_000
g = torch.zeros(1024 + N + 1024, device=dev)
shard = torch.empty(N, device=dev)
comm = torch.cuda.Stream()
ones = torch.ones(1024, device=dev)
for mode in ("rs_ag", "ar", "ag_only", "rs_only"):
    g.zero_()
    g[1024:1024 + N].fill_(2.0)
    torch.cuda.synchronize()
    B = g[1024:1024 + N]
    for it in range(200):
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            if mode == "rs_ag":
                dist.reduce_scatter_tensor(shard, B, op=dist.ReduceOp.AVG); dist.all_gather_into_tensor(B, shard)
            elif mode == "ar":
                dist.all_reduce(B, op=dist.ReduceOp.AVG)
            elif mode == "ag_only":
                dist.all_gather_into_tensor(B, shard if it else B.clone())
            else:
                dist.reduce_scatter_tensor(shard, B, op=dist.ReduceOp.AVG)
        for _ in range(5):                      # the compute stream keeps updating the neighbours of the range while the collective runs
            g[:1024].add_(ones)
            g[1024 + N:].add_(ones)
    torch.cuda.current_stream().wait_stream(comm)
    torch.cuda.synchronize()
    print(mode, "below:", g[:1024].min().item(), g[:1024].max().item(), "above:", g[1024 + N:].min().item(), g[1024 + N:].max().item(), "(expect 1000)",
          "range:", B.min().item(), B.max().item(), flush=True)
dist.destroy_process_group()

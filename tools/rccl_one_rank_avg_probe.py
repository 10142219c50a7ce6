"""Which RCCL call writes below its output at world size 1?  The arena offsets of the failing DDP buckets, each op alone, sentinels around."""
import os, sys, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
G = 118206016 + 52608 + 4096
for op_name, op in (("AVG", dist.ReduceOp.AVG), ("SUM", dist.ReduceOp.SUM)):
    for o, n in ((118206016, 52608), (118186432, 19584), (4096, 1048576), (4096, 52608), (4100, 52608), (4096, 19584)):
        for shard_off in (0, 4):
            g = torch.arange(G, device=dev, dtype=torch.float32) * 1e-3
            want = g.clone()
            shard_buf = torch.full(((1 << 20) + 64,), -7.0, device=dev)
            shard = shard_buf[shard_off:shard_off + n]
            head = g[o:o + n]
            dist.reduce_scatter_tensor(shard, head, op=op)
            torch.cuda.synchronize()
            bad_g1 = (g != want).nonzero().reshape(-1).tolist()[:8]
            sb = shard_buf.clone(); sb[shard_off:shard_off + n] = -7.0
            bad_s1 = (sb != -7.0).nonzero().reshape(-1).tolist()[:8]
            ok_s = bool(torch.equal(shard, want[o:o + n]))
            g2 = torch.zeros_like(g)
            dist.all_gather_into_tensor(g2[o:o + n], shard)
            torch.cuda.synchronize()
            exp2 = torch.zeros_like(g); exp2[o:o + n] = want[o:o + n]
            bad_g2 = (g2 != exp2).nonzero().reshape(-1).tolist()[:8]
            print(op_name, "o", o, "n", n, "shard_off", shard_off, "| reduce_scatter: input changed at", bad_g1, "shard buffer outside changed at", bad_s1, "shard ok", ok_s,
                  "| all_gather: wrong at", bad_g2, [x - o for x in bad_g2])
dist.destroy_process_group()

#!/usr/bin/env python3
"""Bus bandwidth of the data-parallel exchange on this node, outside any training step: all_reduce against reduce_scatter + all_gather
on DistributedDataParallel's bucket sizes (256 MB and 64 MB of f32, and the same element counts in bf16), one rank per GPU over RCCL.

    python tools/rccl_bus_bench.py [N]          # N ranks (default: every visible GPU); starts them itself

Prints one line per (dtype, size): milliseconds and bus GB/s = 2 (N-1)/N * bytes / time of both forms, to be read against the
7 x ~153 GB/s of xGMI links per MI355X (a ring uses one link per direction per neighbour; all-to-all-connected collectives can use all seven).
`python bench.py --gpus N` attaches the same table to its JSON line as comm.microbench."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import torch.distributed as dist
    local = int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist.init_process_group(backend="nccl", init_method="env://", device_id=device)
    from multimodalsum_amd.parallel import bus_microbench
    rows = bus_microbench(device)
    if dist.get_rank() == 0:
        W = dist.get_world_size()
        print("world size %d; xGMI: 7 links x ~153 GB/s per GPU" % W)
        for r in rows:
            print("%-8s %6.0f MB   all_reduce %8.2f ms %7.1f GB/s   reduce_scatter+all_gather %8.2f ms %7.1f GB/s"
                  % (r["dtype"], r["bytes"] / 2**20, r["all_reduce_ms"], r["all_reduce_bus_gb_s"], r["reduce_scatter_all_gather_ms"],
                     r["reduce_scatter_all_gather_bus_gb_s"]), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if "LOCAL_RANK" in os.environ:
        return worker()
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else torch.cuda.device_count()      # device_count() does not initialise the GPU
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.abspath(__file__)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    main()

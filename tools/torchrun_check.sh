#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 3 --warmup 1 --no-also --no-cpu-baseline --no-kernel-probe > gpurun_out/torchrun_bench.json 2> gpurun_out/torchrun_bench.err; echo "rc $?"
tail -1 gpurun_out/torchrun_bench.json | cut -c1-400; tail -3 gpurun_out/torchrun_bench.err | cut -c1-300
MMSUM_FORCE_DDP=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 1 --steps 3 --warmup 1 --no-also --no-cpu-baseline --no-kernel-probe --batch 8 > gpurun_out/torchrun_bench_ddp.json 2> gpurun_out/torchrun_bench_ddp.err; echo "rc $?"
tail -1 gpurun_out/torchrun_bench_ddp.json | cut -c1-700; tail -3 gpurun_out/torchrun_bench_ddp.err | cut -c1-300

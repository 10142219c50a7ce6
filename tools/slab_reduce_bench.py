import sys, os, torch
sys.path.insert(0, os.getcwd())
from multimodalsum_amd import kernels as kn
from tools.gemm_bench import timeit
for M, N, sk in [(1024, 1024, 12), (3072, 1024, 4), (4096, 1024, 3), (256, 1024, 32), (256, 2304, 22), (1024, 256, 32)]:
    ws = torch.randn(sk * M, N, device="cuda"); out = torch.zeros(M, N, device="cuda")
    us = timeit(lambda: kn.slab_reduce(ws, sk, out, accumulate=True)) * 1e3
    print("slab_reduce M=%5d N=%5d sk=%2d  %6.1f us  %5.2f TB/s" % (M, N, sk, us, (sk + 2) * M * N * 4 / us / 1e6), flush=True)

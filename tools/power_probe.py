#!/usr/bin/env python3
"""Socket power and shader clock (rocm-smi, read-only queries) while ONE kind of work runs back to back for a few seconds:
the K = 4096 NT GEMM (MFMA-dense), the K = 1024 one with the GELU epilogue, the vendor GEMM on the same shape (torch.matmul ->
hipBLASLt: calibration of what the chip sustains under its power limit, not a dependency), and an HBM-bound LayerNorm pass.
Prints per phase: TFLOP/s (or TB/s) of the phase, mean / max power, mean / min clock.   usage: python tools/power_probe.py [seconds]"""
import os
import re
import subprocess
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalsum_amd import kernels as kn
from multimodalsum_amd import _lib

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\): ([\d.]+)", out)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            if p and c:
                samples.append((time.time(), float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.25)


def phase(name, fn, work, unit):
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < SECS:
        for i in range(20):
            fn(n + i)
        n += 20
        torch.cuda.synchronize()
    t1 = time.time()
    mine = [s for s in samples if t0 + 1.0 <= s[0] <= t1]          # the first second: the power filter's ramp
    pw, ck = [s[1] for s in mine], [s[2] for s in mine]
    rate = work * n / (t1 - t0)
    print("%-46s %8.1f %s | %d samples: power mean %6.0f W max %6.0f W, shader clock mean %5.0f MHz min %5.0f MHz"
          % (name, rate, unit, len(mine), sum(pw) / max(1, len(pw)), max(pw) if pw else 0, sum(ck) / max(1, len(ck)), min(ck) if ck else 0), flush=True)


def main():
    try:
        print(subprocess.run(["rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout.strip().replace("\n", " | ")[:300])
    except Exception as exc:
        print("rocm-smi --showmaxpower:", exc)
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    dt, M, nbuf = torch.bfloat16, 129024, 4
    time.sleep(2.0)
    idle = samples[-4:]
    print("idle: power %.0f W, shader clock %d MHz" % (sum(s[1] for s in idle) / max(1, len(idle)), idle[-1][2] if idle else 0))
    for name, N, K, kind in (("mmsum_gemm x W^T + bias      [129024,1024,4096]", 1024, 4096, "bias"), ("mmsum_gemm + bias + GELU + aux [129024,4096,1024]", 4096, 1024, "gelu"),
                             ("torch.matmul (hipBLASLt)     [129024,1024,4096]", 1024, 4096, "blas")):
        a = [torch.randn(M, K, device="cuda").to(dt) for _ in range(nbuf)]
        b = [torch.randn(N, K, device="cuda").to(dt) * 0.03 for _ in range(nbuf)]
        out = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nbuf)]
        aux = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(nbuf)] if kind == "gelu" else None
        bias = torch.randn(N, device="cuda")

        def run(i, a=a, b=b, out=out, aux=aux, bias=bias, kind=kind):
            j = i % nbuf
            if kind == "bias":
                kn.gemm(a[j], b[j], out[j], bias=bias)
            elif kind == "gelu":
                kn.gemm(a[j], b[j], out[j], bias=bias, epi=_lib.EPI_GELU, aux=aux[j])
            else:
                torch.matmul(a[j], b[j].t(), out=out[j])
        phase(name, run, 2.0 * M * N * K / 1e12, "TFLOP/s")
        del a, b, out, aux
        torch.cuda.empty_cache()
    x = [torch.randn(M, 1024, device="cuda").to(dt) for _ in range(nbuf)]
    y = [torch.empty(M, 1024, device="cuda", dtype=dt) for _ in range(nbuf)]

    def copy(i):
        y[i % nbuf].copy_(x[i % nbuf])
    phase("device copy [129024,1024] bf16 (HBM-bound)", copy, 2.0 * M * 1024 * 2 / 1e12, "TB/s   ")
    global stop
    stop = True


main()

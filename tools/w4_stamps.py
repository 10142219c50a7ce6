#!/usr/bin/env python3
"""Where a tile of the four-wave NT GEMM kernel spends its time: in-kernel stamps (s_memtime), in a DIAGNOSTIC build.

    python tools/w4_stamps.py build          # CPU: copies csrc/ to tools/build/stamps/, patches gemm_fast.hip, builds libmmsum_hip.so there
    python tools/w4_stamps.py run [M N K]    # GPU: runs x W^T (+bias) through that library and prints the breakdown per tile

The shipped kernel carries no stamp: the patch below adds, to a copy of the source, five stamps per tile taken by wave 0 --
tile start, first fragments read (prologue done: the first DMA round trip), main loop done, epilogue done (all stores issued),
tile end -- accumulated per workgroup in registers and written to a __device__ array at kernel exit (nothing the kernel computes
reads them).  Cycles are shader cycles (s_memtime); microseconds use the clock measured over the kernel (s_memrealtime, 100 MHz).
"""
import ctypes
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "multimodalsum_amd", "csrc")
DST = os.path.join(ROOT, "tools", "build", "stamps")


def patch(text):
    def rep(old, new, count=1):
        nonlocal text
        assert text.count(old) >= count, old[:70]
        text = text.replace(old, new, count)

    # storage + accessor
    rep("namespace {\n", "__device__ unsigned long long g_w4_stamps[256][12];\n"
        "extern \"C\" __attribute__((visibility(\"default\"))) int mmsum_w4_stamps(unsigned long long* host256x12) {\n"
        "    return hipMemcpyFromSymbol(host256x12, HIP_SYMBOL(g_w4_stamps), sizeof(g_w4_stamps)) == hipSuccess ? 0 : -5;\n}\n"
        "extern \"C\" __attribute__((visibility(\"default\"))) int mmsum_w4_stamps_clear(void) {\n"
        "    static unsigned long long z[256][12];\n"
        "    return hipMemcpyToSymbol(HIP_SYMBOL(g_w4_stamps), z, sizeof(z)) == hipSuccess ? 0 : -5;\n}\n"
        "namespace {\n")
    # per-workgroup accumulators, inside the w4 NT kernel only (anchors that occur once, in that kernel)
    rep("    const int ydelta = (fo ^ 64) - fo;\n",
        "    const int ydelta = (fo ^ 64) - fo;\n"
        "    unsigned long long st_pro = 0, st_main = 0, st_epi = 0, st_end = 0, st_tiles = 0, st_gap = 0, st_prev = 0, st_head = 0, st_sync = 0, st_vm = 0;\n"
        "    const unsigned long long st_k0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();\n")
    rep("        // prologue, in the order the steady state issues: A(0) B(0) A(1) B(1) A(2)\n",
        "        const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();\n"
        "        if (st_prev != 0) st_gap += st_t0 - st_prev; else st_head = st_t0 - st_k0;\n"
        "        // prologue, in the order the steady state issues: A(0) B(0) A(1) B(1) A(2)\n")
    rep("            if constexpr (LEFT >= 2) wait_vmcnt<8>(); else if constexpr (LEFT == 1) wait_vmcnt<0>();     // stage st+1 has landed; A of st+2 may be in flight\n"
        "            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
        "            __builtin_amdgcn_s_barrier();\n",
        "            asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
        "            const unsigned long long st_s0 = __builtin_amdgcn_s_memtime();\n"
        "            if constexpr (LEFT >= 2) wait_vmcnt<8>(); else if constexpr (LEFT == 1) wait_vmcnt<0>();\n"
        "            const unsigned long long st_s1 = __builtin_amdgcn_s_memtime();\n"
        "            __builtin_amdgcn_s_barrier();\n"
        "            { const unsigned long long st_s2 = __builtin_amdgcn_s_memtime(); st_vm += st_s1 - st_s0; st_sync += st_s2 - st_s1; }\n")
    rep("        // four MFMAs: A block I x B block J (quarters q = 2 si + sj)\n",
        "        asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n"
        "        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();\n"
        "        // four MFMAs: A block I x B block J (quarters q = 2 si + sj)\n")
    rep("        if (st < nst) stage(st, abuf, std::integral_constant<int, 0>{}, std::false_type{});\n        asm volatile(\"s_nop 15\\n s_nop 15\" ::: \"memory\");",
        "        if (st < nst) stage(st, abuf, std::integral_constant<int, 0>{}, std::false_type{});\n        asm volatile(\"s_nop 15\\n s_nop 15\" ::: \"memory\");\n"
        "        const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();\n"
        "        st_pro += st_t1 - st_t0; st_main += st_t2 - st_t1; st_epi -= st_t2; st_end -= st_t2; ++st_tiles;")
    rep("                                                                                                                     wave * 64 + lane_e, lane_e);\n    }\n    lds_barrier();\n    }\n}\n",
        "                                                                                                                     wave * 64 + lane_e, lane_e);\n    }\n"
        "    { const unsigned long long t3 = __builtin_amdgcn_s_memtime(); st_epi += t3; }\n"
        "    asm volatile(\"s_waitcnt vmcnt(0)\" ::: \"memory\");      // DIAGNOSTIC ONLY: how long the tile's stores take to be acknowledged\n"
        "    { const unsigned long long t4 = __builtin_amdgcn_s_memtime(); st_end += t4; st_prev = t4; }\n"
        "    lds_barrier();\n    }\n"
        "    if (threadIdx.x == 0 && blockIdx.x < 256) {\n"
        "        unsigned long long* o = g_w4_stamps[blockIdx.x];\n"
        "        o[0] = st_tiles; o[1] = st_pro; o[2] = st_main; o[3] = st_epi; o[4] = st_end;\n"
        "        o[5] = __builtin_amdgcn_s_memtime() - st_k0; o[6] = __builtin_amdgcn_s_memrealtime() - st_r0; o[7] = st_gap;\n"
        "        g_w4_stamps[blockIdx.x][8] = st_head; g_w4_stamps[blockIdx.x][9] = __builtin_amdgcn_s_memtime() - st_prev; g_w4_stamps[blockIdx.x][10] = st_vm; g_w4_stamps[blockIdx.x][11] = st_sync;\n    }\n}\n")
    return text


def build():
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    os.makedirs(os.path.dirname(DST), exist_ok=True)
    shutil.copytree(SRC, DST, ignore=shutil.ignore_patterns("*.o", "*.so", "*.txt"))
    p = os.path.join(DST, "gemm_fast.hip")
    patched = patch(open(p).read())
    open(p, "w").write(patched)
    mk = os.path.join(DST, "Makefile")
    m = open(mk).read()
    m = m.replace("../../include/mmsum_hip.h", os.path.join(ROOT, "include", "mmsum_hip.h"))
    m = m.replace("\tpython3 check_resources.py gemm_fast.resources.txt || { rm -f $@; exit 1; }\n", "")       # the stamps cost registers: diagnostic build
    open(mk, "w").write(m)
    for f in os.listdir(DST):                               # includes of the public header by relative path
        if f.endswith((".h", ".hip")):
            q = os.path.join(DST, f)
            t = open(q).read()
            if "../../include/mmsum_hip.h" in t:
                t = t.replace("../../include/mmsum_hip.h", os.path.join(ROOT, "include", "mmsum_hip.h"))
                open(q, "w").write(t)
    subprocess.check_call(["make", "-C", DST, "-j4"])
    print("built", os.path.join(DST, "libmmsum_hip.so"))


def run(M, N, K, variants=("plain", "bias", "gelu", "acc")):
    os.environ["MMSUM_LIB"] = os.path.join(DST, "libmmsum_hip.so")
    sys.path.insert(0, ROOT)
    import torch
    from multimodalsum_amd import kernels as kn, _lib
    lib = _lib.lib
    dt = torch.bfloat16
    a = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") * 0.03).to(dt)
    out = torch.randn(M, N, device="cuda").to(dt)
    aux = torch.empty(M, N, device="cuda", dtype=dt)
    bias = torch.randn(N, device="cuda")
    buf = (ctypes.c_ulonglong * (256 * 12))()
    for var in variants:
        def launch():
            if var == "plain":
                kn.gemm(a, w, out)
            elif var == "bias":
                kn.gemm(a, w, out, bias=bias)
            elif var == "gelu":
                kn.gemm(a, w, out, bias=bias, epi=kn.EPI_GELU, aux=aux)
            else:
                kn.gemm(a, w, out, accumulate=True)
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        lib.mmsum_w4_stamps_clear()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        assert lib.mmsum_w4_stamps(buf) == 0
        rows = [[buf[i * 12 + j] for j in range(12)] for i in range(256)]
        rows = [r for r in rows if r[0] > 0]
        tiles = sum(r[0] for r in rows)
        ghz = sum(r[5] for r in rows) / (sum(r[6] for r in rows) * 10.0)       # cycles per 10 ns tick -> GHz
        cyc = [sum(r[j] for r in rows) / tiles for j in (1, 2, 3, 4)]
        us = [c / (ghz * 1e3) for c in cyc]
        gaps = sum(r[7] for r in rows) / max(1, tiles - len(rows)) / (ghz * 1e3)          # between a tile's end and the next one's first DMA
        head = sum(r[8] for r in rows) / len(rows) / (ghz * 1e3)
        tail = sum(r[9] for r in rows) / len(rows) / (ghz * 1e3)
        inker = max(r[6] for r in rows) / 100.0                                          # the longest workgroup, in us (100 MHz ticks)
        print("%-5s M=%d N=%d K=%d: %.1f us launch (longest workgroup %.1f us), %d tiles on %d workgroups, %.2f GHz in-kernel | per tile: prologue %.2f us, "
              "main loop %.2f us (%d stages), epilogue issue %.2f us, + until its stores are acknowledged %.2f us, tile end -> next tile's first DMA %.2f us "
              "(sum %.2f us) | per workgroup: kernel start -> first DMA %.2f us, last store -> exit %.2f us | inside the main loop, per stage: "
              "waiting for the own DMA of the next stage (vmcnt) %.0f cycles, at the barrier %.0f cycles, of %.0f cycles per stage (2,048 = MFMA issue)"
              % (var, M, N, K, e0.elapsed_time(e1) * 1e3, inker, tiles, len(rows), ghz, us[0], us[1], K // 64, us[2], us[3] - us[2], gaps,
                 us[0] + us[1] + us[3] + gaps, head, tail, sum(r[10] for r in rows) / tiles / (K // 64), sum(r[11] for r in rows) / tiles / (K // 64),
                 cyc[1] / (K // 64)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        args = [int(x) for x in sys.argv[2:5]] if len(sys.argv) >= 5 else None
        shapes = [tuple(args)] if args else [(64512, 1024, 1024), (64512, 1024, 4096), (64512, 4096, 1024), (64512, 1024, 64)]
        for M, N, K in shapes:
            run(M, N, K)

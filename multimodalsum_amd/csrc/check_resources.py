"""Build-time guard (csrc/Makefile runs it on the compiler's -Rpass-analysis=kernel-resource-usage remarks of gemm_fast.hip).

gemm_tn_w4_kernel keeps its 256 accumulators in AGPRs that it names in inline asm (gemm_tn_w4_acc.inc); the compiler does not
know they are live between the zero-fill and the read-out, so it must never need an AGPR or a scratch slot of its own in that
kernel: the build fails if the kernel spills, uses scratch, or asks for more than 256 architectural VGPRs (beyond that the
allocator would take AGPRs).  The four-wave NT kernel binds its accumulators as "+a" operands (the compiler knows them), but a
spill inside its main loop would put scratch traffic on the vmcnt counter the DMA pipeline is counted on: its allowance is a
handful of registers, which the allocator parks around the main loop, never in it (its tile header and its epilogue re-read the
lane index opaquely so that nothing lane-constant is carried across; tools/asm_scratch_report.py shows where spills sit)."""
import re
import sys

LIMITS = {                    # kernel-name substring -> (max arch VGPRs, max scratch bytes / lane, max VGPR spills)
    "gemm_tn_w4_kernel": (256, 0, 0),
    # NT: a few registers may be parked in scratch AROUND the main loop (stored in the tile header, reloaded for the epilogue:
    # two to four instructions per tile); tools/asm_scratch_report.py must show none between the first and the last MFMA of a tile
    "gemm_nt_w4_kernel": (256, 48, 10),
    # the f32-atomic output form (split-K without slabs: not on the training step's path) stores element by element from the
    # accumulator layout and spills a few more registers in that epilogue; more specific entries win
    "gemm_nt_w4_kernelILi0ELi3E": (256, 96, 24),
}


def main(path):
    text = open(path).read()
    blocks = re.split(r"remark: Function Name: ", text)[1:]
    seen, bad = set(), []
    for b in blocks:
        name = b.split()[0]
        keys = [k for k in LIMITS if k in name]
        for key in sorted(keys, key=len)[-1:]:                       # the longest (most specific) matching entry
            max_v, max_scratch, max_spill = LIMITS[key]
            seen.add(key)
            seen.update(k for k in keys)

            def field(k):
                return int(re.search(k + r": (\d+)", b).group(1))
            v, a, scratch, spill = field("VGPRs"), field("AGPRs"), field(r"ScratchSize \[bytes/lane\]"), field("VGPRs Spill")
            if v > max_v or scratch > max_scratch or spill > max_spill or a > 256:
                bad.append("%s: VGPRs %d (max %d), AGPRs %d, scratch %d B/lane (max %d), VGPR spills %d (max %d)"
                           % (name, v, max_v, a, scratch, max_scratch, spill, max_spill))
    missing = set(LIMITS) - seen
    if missing:
        bad.append("no resource remarks for %s: was -Rpass-analysis=kernel-resource-usage dropped from the Makefile?" % sorted(missing))
    if bad:
        sys.stderr.write("check_resources: kernel resource limits violated\n  " + "\n  ".join(bad) + "\n")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))

#!/usr/bin/env python3
"""Build-time guard of the fence-free in-launch hand-offs (gemm_skinny.hip: dec_gemm_kernel's split-K slabs; decode.hip:
decode_cross_attn_kernel / decode_cross_attn_f32_kernel's entity mean; gemm_fast.hip: the fused split of the small NT products).

Those kernels pass partial results between workgroups of ONE launch through memory without release / acquire fences (an agent-scope
release writes back an XCD's whole L2: 14 us per product when 256 - 512 workgroups each pay it, profiles/NOTES_r04.md): payload words
are stored write-through (relaxed agent-scope atomic stores = `global_store ... sc1`), the storing wave drains them (`s_waitcnt
vmcnt(0)`), one lane takes a ticket (relaxed agent-scope `global_atomic_add`), and the last arriver reads every payload word past its
L1 / the XCD's L2 (`global_load ... sc1`).  That is sound on gfx950 AS LONG AS the compiler lowers the relaxed atomics that way; a
toolchain that drops the sc1 bit, moves the wait, or merges the payload loads into plain ones would corrupt the decode step silently.
This script reads the device assembly the object was built from (hipcc -save-temps) and fails the build unless, in every such kernel:

  1. there is a ticket `global_atomic_add`;
  2. every `global_store` before the first ticket carries sc1 (they are the payload), and there is at least one;
  3. an `s_waitcnt` with vmcnt(0) lies between the last payload store and the ticket;
  4. after the ticket at least one `global_load_dword*` carries sc1 and at most ALLOW[kernel] dword loads do not (dec_gemm: the bias).

usage: check_handoff.py <device .s file> [...]
"""
import re
import sys

# kernel family (regular expression on the mangled name) -> plain dword loads allowed after the ticket (dec_gemm: the bias; the fused
# split of the small NT products, gemm_nt_ring_kernel<..., FS = true>: whatever the ordinary epilogue behind the reduction reads --
# bias, saved pre-activation, the C of an accumulating store)
KERNELS = {"dec_gemm_kernel": 1, "decode_cross_attn_kernel": 0, "decode_cross_attn_f32_kernel": 0, r"gemm_nt_ring_kernel\w*Lb1EEEv8GemmArgs": 256}
LABEL = re.compile(r"^(_Z\w+):")


def functions(path):
    name, body = None, []
    with open(path) as f:
        for line in f:
            m = LABEL.match(line)
            if m:
                name, body = m.group(1), []
                continue
            if name is None:
                continue
            t = line.strip()
            if t and not t.startswith((";", ".")):
                body.append(t.split(";")[0].strip())
            if t.startswith("s_endpgm"):
                pass
            if t.startswith(".Lfunc_end"):
                yield name, body
                name = None


def check(name, allow, body):
    errs = []
    atom = [i for i, t in enumerate(body) if t.startswith("global_atomic_add")]
    if not atom:
        return ["no ticket global_atomic_add"]
    a = atom[0]
    stores = [i for i in range(a) if body[i].startswith("global_store")]
    if not stores:
        errs.append("no payload store before the ticket")
    for i in stores:
        if " sc1" not in body[i]:
            errs.append("payload store without sc1: %s" % body[i])
    if stores and not any(body[i].startswith("s_waitcnt") and "vmcnt(0)" in body[i] for i in range(stores[-1] + 1, a)):
        errs.append("no s_waitcnt vmcnt(0) between the last payload store and the ticket")
    loads = [body[i] for i in range(a + 1, len(body)) if body[i].startswith("global_load_dword")]
    plain = [t for t in loads if " sc1" not in t]
    if len(loads) == len(plain):
        errs.append("no sc1 payload load after the ticket")
    if len(plain) > allow:
        errs.append("%d plain dword loads after the ticket (allowed %d): %s" % (len(plain), allow, plain[:3]))
    return errs


def main(paths):
    seen, bad = {k: 0 for k in KERNELS}, 0
    for path in paths:
        for name, body in functions(path):
            fam = next((k for k in sorted(KERNELS, key=len, reverse=True) if re.search(k, name)), None)
            if fam is None or not any(t.startswith("s_endpgm") for t in body):
                continue
            seen[fam] += 1
            for e in check(name, KERNELS[fam], body):
                bad += 1
                print("check_handoff: %s: %s" % (name, e), file=sys.stderr)
    present = [k for k, n in seen.items() if n]
    if not present:
        print("check_handoff: none of %s found in %s" % (sorted(KERNELS), paths), file=sys.stderr)
        return 1
    if bad:
        return 1
    print("check_handoff: ok (%s)" % ", ".join("%s x%d" % (k, seen[k]) for k in present))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

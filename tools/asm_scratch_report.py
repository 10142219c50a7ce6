"""Where a kernel's scratch (spill) traffic sits relative to its MFMA main loop.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S file.hip -o file.s; python tools/asm_scratch_report.py file.s [name-filter]
For every kernel: lines, MFMAs, scratch loads/stores, and how many of them lie inside a loop that contains MFMAs
(the innermost backward branches whose span holds an MFMA): spills there cost every iteration, spills outside cost once per tile."""
import re
import sys


def report(path, flt=""):
    t = open(path).read()
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm", t, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        lines = body.split("\n")
        labels = {l.split(":")[0].strip(): i for i, l in enumerate(lines) if re.match(r"^\.?\w+:", l)}
        mf = [i for i, l in enumerate(lines) if "v_mfma" in l]
        sc = [i for i, l in enumerate(lines) if re.search(r"\bscratch_(load|store)", l)]
        loops = []
        for i, l in enumerate(lines):
            b = re.search(r"s_cbranch_\w+\s+(\.?\w+)|s_branch\s+(\.?\w+)", l)
            if b:
                tgt = labels.get(b.group(1) or b.group(2))
                if tgt is not None and tgt < i and any(tgt <= x <= i for x in mf):
                    loops.append((tgt, i))
        inner = [(a, b) for a, b in loops if not any((c, d) != (a, b) and a <= c and d <= b for c, d in loops)]
        hot = [i for i in sc if any(a <= i <= b for a, b in inner)]
        print("%-70s lines %6d mfma %5d scratch %4d in-innermost-mfma-loops %4d" % (name[-70:], len(lines), len(mf), len(sc), len(hot)))


if __name__ == "__main__":
    report(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

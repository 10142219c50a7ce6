#!/usr/bin/env python3
"""Instruction-class pattern of a kernel's basic blocks in a device .s file (hipcc --cuda-device-only -S): one character per instruction
(M mfma, E v_exp, v other VALU, r LDS read, w LDS write, L global/buffer load, S store, W s_waitcnt, B s_barrier, s other scalar), for the blocks with at
least MIN_MFMA MFMAs.  usage: isa_pattern.py file.s <substring of the kernel's mangled name> [MIN_MFMA]"""
import re
import sys


def cls(op):
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('v_exp'): return 'E'
    if op.startswith(('ds_read', 'ds_load')): return 'r'
    if op.startswith(('ds_write', 'ds_store')): return 'w'
    if op.startswith(('buffer_load', 'global_load')): return 'L'
    if op.startswith(('buffer_store', 'global_store')): return 'S'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_waitcnt'): return 'W'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith('s_'): return 's'
    return '?'


def main(path, name, min_mfma=16):
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and name in l]
    for start in starts:
        end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
        print(lines[start].split(':')[0], end - start, 'lines')
        blocks, cur = [], []
        for l in lines[start:end]:
            t = l.strip()
            if re.match(r'^\.LBB\d+_\d+:', t):
                blocks.append(cur)
                cur = []
            elif t and not t.startswith((';', '.')):
                cur.append(t.split()[0])
        blocks.append(cur)
        for bi, b in enumerate(blocks):
            nm = sum(1 for o in b if o.startswith('v_mfma'))
            if nm >= min_mfma:
                print(' block %d: %d instructions, %d mfma, %d v_exp, %d other VALU' % (bi, len(b), nm, sum(1 for o in b if o.startswith('v_exp')),
                                                                                     sum(1 for o in b if cls(o) == 'v')))
                print('  ' + ''.join(cls(o) for o in b))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 16)

#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv by kernel family.  usage: prof_summary.py file.csv nsteps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
if n <= 0:        # 0 = count the steps: one adamw_kernel launch per optimizer step
    n = float(sum(int(r['Calls']) for r in rows if 'adamw_kernel' in r['Name']) or 1)
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step: %.2f" % (tot / n / 1e6))
groups = {}
for r in rows:
    nm = r['Name']
    key = ('gemm_fast' if ('gemm_nt_' in nm or 'gemm_tn_' in nm) else 'gemm_skinny' if 'gemm_skinny' in nm else 'rows_gather' if 'rows_gather' in nm else 'slab_reduce' if 'slab_reduce' in nm else 'gemm_generic' if 'gemm_kernel' in nm else 'attn_fwd' if ('attn_fwd' in nm or 'attn_tr_fwd' in nm)
           else 'attn_dq' if 'bwd_dq' in nm else 'attn_dkv' if 'bwd_dkv' in nm else 'transpose' if 'transpose' in nm
           else 'bn' if 'bn_' in nm else 'im2col_col2im' if ('im2col' in nm or 'col2im' in nm) else 'colsum' if 'colsum' in nm else 'ln' if '_ln_' in nm else 'adamw' if 'adamw' in nm
           else 'loss' if 'ls_loss' in nm else 'torch' if ('at::native' in nm or 'rocclr' in nm) else 'other')
    groups[key] = groups.get(key, 0) + int(r['TotalDurationNs'])
for k, v in sorted(groups.items(), key=lambda x: -x[1]):
    print("%-14s %8.2f ms/step  %5.1f%%" % (k, v / n / 1e6, 100.0 * v / tot))
if len(sys.argv) > 3:
    for r in rows:
        if sys.argv[3] in r['Name']:
            print(r['Name'][-70:], r['Calls'], "avg %.0f us" % (float(r['AverageNs']) / 1e3))

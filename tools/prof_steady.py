#!/usr/bin/env python3
"""Per-step kernel table of the STEADY-STATE steps from a rocprofv3 kernel trace.  usage: prof_steady.py <kernel_trace.csv> <steps> [top]
The window is everything between the (steps+1)-th last and the last adamw_kernel launch, i.e. the last `steps` training steps: model
construction (formula-init hash kernels), graph capture and priming steps are outside it, so one-time launches no longer show up as
per-step "torch" time (the round-4 summary divided the whole trace by the optimizer launches)."""
import collections, csv, re, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
steps = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
assert len(ad) > steps, "trace holds %d optimizer launches, need more than %d" % (len(ad), steps)
rows = rows[ad[-steps - 1] + 1:ad[-1] + 1]
tot = collections.Counter()
cnt = collections.Counter()
for s, e, n in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    tot[n] += e - s
    cnt[n] += 1
total = sum(tot.values())
span = rows[-1][1] - rows[0][0]
print("steady-state window: last %d steps, %d kernels; span %.2f ms/step, kernel time %.2f ms/step" % (steps, len(rows), span / 1e6 / steps, total / 1e6 / steps))
for n, v in tot.most_common(top):
    print("%7.2f ms %5.1f%%  calls %7.1f  avg %8.1f us  %s" % (v / steps / 1e6, 100.0 * v / total, cnt[n] / steps, v / cnt[n] / 1e3, n[:110]))


def family(nm):
    return ('gemm_fast' if ('gemm_nt_' in nm or 'gemm_tn_' in nm) else 'gemm_skinny' if 'gemm_skinny' in nm else 'rows_gather' if 'rows_gather' in nm
            else 'slab_reduce' if 'slab_reduce' in nm else 'gemm_generic' if 'gemm_kernel' in nm else 'attn_fwd' if ('attn_fwd' in nm or 'attn_tr_fwd' in nm)
            else 'attn_dq' if 'bwd_dq' in nm else 'attn_dkv' if 'bwd_dkv' in nm else 'attn_self_bwd' if 'bwd_self' in nm else 'transpose' if 'transpose' in nm
            else 'bn' if 'bn_' in nm else 'im2col_col2im' if ('im2col' in nm or 'col2im' in nm) else 'colsum' if 'colsum' in nm else 'ln' if '_ln_' in nm
            else 'gate' if 'gate_' in nm else 'adamw' if 'adamw' in nm else 'loss' if 'ls_loss' in nm else 'image_plan' if 'image_' in nm
            else 'torch' if ('at::native' in nm or 'rocclr' in nm) else 'other')


groups = collections.Counter()
gcnt = collections.Counter()
for n, v in tot.items():
    groups[family(n)] += v
    gcnt[family(n)] += cnt[n]
print("families (steady state):")
for k, v in groups.most_common():
    print("%-14s %8.2f ms/step  %5.1f%%  launches/step %7.1f" % (k, v / steps / 1e6, 100.0 * v / total, gcnt[k] / steps))

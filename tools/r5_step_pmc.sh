#!/usr/bin/env bash
# Whole-step hardware counters: one --pmc pass (--kernel-trace only) over two eager steps of the bench configuration; the summary counts the
# SECOND step only (tools/step_pmc_summary.py --after-first-adamw).  (The FETCH_SIZE / WRITE_SIZE passes over a whole B = 128 step did not
# finish on this pool -- rocprofv3 segfaulted in one, the other ran into its 20-minute limit -- so the step's HBM traffic is not collected;
# the dominant kernel's is: tools/pmc_dominant.sh.)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d "$R"/gpurun_out/spmc_s -o s -- python3 "$R"/bench.py --no-graphs --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-probe --no-also > "$R"/gpurun_out/spmc_s.log 2>&1
echo "pass s rc $?"
cd "$R"
python tools/step_pmc_summary.py 1 --after-first-adamw gpurun_out/spmc_s/s_counter_collection.csv
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/spmc_s/s_counter_collection.csv")))
first = min(int(r["Dispatch_Id"]) for r in rows if "adamw_kernel" in r["Kernel_Name"])
fam = collections.defaultdict(lambda: collections.Counter())
def family(n):
    if "gemm" in n: return "gemm"
    if "attn" in n: return "attention"
    return "other"
for r in rows:
    if int(r["Dispatch_Id"]) > first:
        fam[family(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
for f, c in fam.items():
    if c["GRBM_GUI_ACTIVE"]:
        print("%-10s MFMA pipe busy %.1f %% of its kernels' cycles (busy %.4g, GUI_ACTIVE %.4g)" % (f, 100.0 * (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0), c["SQ_VALU_MFMA_BUSY_CYCLES"], c["GRBM_GUI_ACTIVE"]))
PY
rm -rf gpurun_out/spmc_s

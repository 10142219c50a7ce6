#!/usr/bin/env bash
# Whole-step SQ counters (MFMA pipe busy) of the bench configuration: one --pmc pass (--kernel-trace only) over two eager steps, the SECOND
# step's dispatches counted (those after the first optimizer kernel) -> gpurun_out/r6_step_sq.json (tools/r5_step_pmc.sh with a JSON result).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d "$R"/gpurun_out/spmc_s -o s -- python3 "$R"/bench.py --no-graphs --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-probe --no-also > "$R"/gpurun_out/spmc_s.log 2>&1
echo "pass s rc $?"
cd "$R"
python - <<'PY'
import csv, collections, json
rows = list(csv.DictReader(open("gpurun_out/spmc_s/s_counter_collection.csv")))
first = min(int(r["Dispatch_Id"]) for r in rows if "adamw_kernel" in r["Kernel_Name"])
fam = collections.defaultdict(collections.Counter)
tot = collections.Counter()
ms, seen = 0.0, set()
def family(n):
    if "gemm" in n: return "gemm"
    if "attn" in n: return "attention"
    return "other"
for r in rows:
    if int(r["Dispatch_Id"]) > first:
        fam[family(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            ms += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
busy = lambda c: (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0) if c["GRBM_GUI_ACTIVE"] else 0.0
# one v_mfma_f32_32x32x16_bf16 = 32 busy cycles = 32,768 FLOP; 16x16x32 = 16 cycles = 16,384 FLOP: 1,024 FLOP per busy cycle either way
out = {"round": 6, "workload": "multimodal", "per_gpu_batch": 128,
       "command": "tools/r6_step_sq.sh (rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python3 bench.py --no-graphs --steps 1 --warmup 1 ...; the second step's dispatches)",
       "kernel_ms_under_profiler": ms, "GRBM_GUI_ACTIVE": tot["GRBM_GUI_ACTIVE"], "SQ_VALU_MFMA_BUSY_CYCLES": tot["SQ_VALU_MFMA_BUSY_CYCLES"],
       "mfma_busy_frac": busy(tot),
       "by_family": {f: {"mfma_busy_frac": busy(c), "SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": c["GRBM_GUI_ACTIVE"]} for f, c in fam.items()},
       "mfma_flops_per_step": tot["SQ_VALU_MFMA_BUSY_CYCLES"] * 1024.0, "mfma_flops_per_business": tot["SQ_VALU_MFMA_BUSY_CYCLES"] * 1024.0 / 128}
json.dump(out, open("gpurun_out/r6_step_sq.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("mfma_busy_frac", "kernel_ms_under_profiler", "mfma_flops_per_business")}), {f: round(v["mfma_busy_frac"], 3) for f, v in out["by_family"].items()})
PY
rm -rf gpurun_out/spmc_s

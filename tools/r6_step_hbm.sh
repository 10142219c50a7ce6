#!/usr/bin/env bash
# HBM-side bytes of a whole training step from the TCC counters: one rocprofv3 --pmc pass per counter (FETCH_SIZE, WRITE_SIZE; --kernel-trace
# only) over `bench.py --no-graphs --steps 1 --warmup 1` at batch $STEP_B (default 128); only the SECOND step's dispatches are counted
# (those after the first optimizer kernel).  Writes gpurun_out/r6_step_hbm_B<batch>.json: per family and for the step, bytes read / written
# (FETCH_SIZE doubled: the guide's gfx950 correction), kernel time under the profiler, GB/s.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
B="${STEP_B:-128}"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout ${STEP_PMC_LIMIT:-1500} rocprofv3 --output-format csv --kernel-trace --pmc $c -d "$R"/gpurun_out/shbm_$c -o p -- python3 "$R"/bench.py --batch $B --no-graphs --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-probe --no-also > "$R"/gpurun_out/shbm_$c.log 2>&1
  echo "pass $c rc $?"
done
cd "$R"
python - "$B" <<'PY'
import csv, collections, json, os, sys
B = int(sys.argv[1])
def family(n):
    if "gemm" in n: return "gemm"
    if "attn" in n: return "attention"
    if "add_ln" in n or "embed_ln" in n: return "layernorm"
    if "bn_" in n or "im2col" in n or "col2im" in n or "maxpool" in n or "nchw" in n or "image_" in n: return "image_rowwise"
    if "ls_loss" in n: return "loss"
    if "gate_" in n: return "gate"
    if "adamw" in n or "l2_" in n: return "optimizer"
    if "slab_reduce" in n: return "slab_reduce"
    return "other"
out = {"round": 6, "workload": "multimodal", "per_gpu_batch": B,
       "command": "STEP_B=%d bash tools/r6_step_hbm.sh (rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --batch %d --no-graphs --steps 1 --warmup 1 ...; the second step's dispatches; FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024 bytes)" % (B, B),
       "by_family": {}}
tot = collections.Counter()
for c, scale in (("FETCH_SIZE", 2048.0), ("WRITE_SIZE", 1024.0)):
    path = "gpurun_out/shbm_%s/p_counter_collection.csv" % c
    if not os.path.exists(path):
        out[c] = None
        continue
    rows = list(csv.DictReader(open(path)))
    first = min((int(r["Dispatch_Id"]) for r in rows if "adamw_kernel" in r["Kernel_Name"]), default=-1)
    seen = set()
    for r in rows:
        if int(r["Dispatch_Id"]) <= first or r["Counter_Name"] != c:
            continue
        f = family(r["Kernel_Name"])
        d = out["by_family"].setdefault(f, {"read_bytes": 0.0, "write_bytes": 0.0, "kernel_ms": 0.0, "launches": 0})
        d["read_bytes" if c == "FETCH_SIZE" else "write_bytes"] += float(r["Counter_Value"]) * scale
        if c == "FETCH_SIZE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            d["kernel_ms"] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
            d["launches"] += 1
    out[c] = True
rd = sum(d["read_bytes"] for d in out["by_family"].values())
wr = sum(d["write_bytes"] for d in out["by_family"].values())
ms = sum(d["kernel_ms"] for d in out["by_family"].values())
for d in out["by_family"].values():
    d["gbps"] = (d["read_bytes"] + d["write_bytes"]) / max(d["kernel_ms"], 1e-9) / 1e6
out["hbm_traffic"] = {"read_bytes_per_step": rd, "write_bytes_per_step": wr, "kernel_ms_under_profiler": ms,
                      "gbps_over_kernel_time": (rd + wr) / max(ms, 1e-9) / 1e6} if out.get("FETCH_SIZE") and out.get("WRITE_SIZE") else None
json.dump(out, open("gpurun_out/r6_step_hbm_B%d.json" % B, "w"), indent=1)
print(json.dumps(out["hbm_traffic"]))
for f, d in sorted(out["by_family"].items(), key=lambda kv: -kv[1]["kernel_ms"]):
    print("%-14s %7.1f ms  launches %5d  read %8.2f GB  written %8.2f GB  %7.0f GB/s" % (f, d["kernel_ms"], d["launches"], d["read_bytes"] / 1e9, d["write_bytes"] / 1e9, d["gbps"]))
PY
rm -rf gpurun_out/shbm_FETCH_SIZE gpurun_out/shbm_WRITE_SIZE
tail -3 gpurun_out/shbm_FETCH_SIZE.log gpurun_out/shbm_WRITE_SIZE.log 2>/dev/null | cut -c1-300

#!/usr/bin/env python3
"""The dominant kernel's PMC passes (tools/pmc_dominant.sh) -> the JSON bench.py reads roofline.traffic from.
usage: pmc_dominant_json.py <round> M N K out.json f.csv w.csv s.csv t.csv
Records the sha1 of the kernel's sources and the library's build id, so that bench.py can tell a stale file from a current one."""
import collections, csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("gemm_fast.hip", "gemm_common.h", "mmsum_device.h")


def source_sha1():
    return {f: hashlib.sha1(open(os.path.join(ROOT, "multimodalsum_amd", "csrc", f), "rb").read()).hexdigest() for f in SOURCES}


def main():
    rnd, M, N, K = (int(x) for x in sys.argv[1:5])
    out = sys.argv[5]
    acc = collections.defaultdict(list)
    for f in sys.argv[6:]:
        for r in csv.DictReader(open(f)):
            if "gemm_nt_w4" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    rd = mean["FETCH_SIZE"] * 1024 * 2.0             # KiB; doubled per the guide's gfx950 correction (64-byte units counted as 32)
    wr = mean["WRITE_SIZE"] * 1024
    alg = M * K * 2 + N * K * 2 + 2 * M * N * 2 + N * 4
    try:
        sys.path.insert(0, ROOT)
        from multimodalsum_amd import _lib
        build = _lib.lib.mmsum_build_id().decode()
    except Exception:
        build = None
    doc = {"round": rnd, "kernel": "gemm_nt_w4_kernel<EPI_GELU,OUT_T> (FFN up-projection + bias + GELU, saves the pre-activation)",
           "command": "PMC_M=%d bash tools/pmc_dominant.sh  (rocprofv3 --kernel-trace --pmc <group> -- python3 tools/gemm_one.py %d %d %d gelu 6; one pass per counter group)" % (M, M, N, K),
           "shape": [M, N, K], "launches_averaged": len(acc["FETCH_SIZE"]), "fetch_size_kib": mean["FETCH_SIZE"], "write_size_kib": mean["WRITE_SIZE"],
           "gfx950_fetch_correction": 2.0, "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch": float(alg), "traffic_over_algorithmic": (rd + wr) / alg, "duration_ns_under_pmc": mean["_dur_ns"],
           "sq": {k: mean[k] for k in mean if k.startswith("SQ_") or k.startswith("GRBM_")}, "tcc": {k: mean[k] for k in mean if k.startswith("TCC_")},
           "mmsum_build_id": build, "source_sha1": source_sha1()}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("shape", "hbm_bytes_per_launch", "traffic_over_algorithmic", "duration_ns_under_pmc", "mmsum_build_id")}))


if __name__ == "__main__":
    main()

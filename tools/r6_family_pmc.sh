#!/usr/bin/env bash
# Counter-based HBM bytes per kernel family.  The whole-step FETCH_SIZE / WRITE_SIZE passes do not finish on this pool at
# B = 128 (25-minute limit / a rocprofv3 segfault, again in round 6: tools/r6_step_hbm.sh), so ONE representative launch of every family is
# profiled in isolation at the bench batch's sizes -- rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes -- and the
# step's traffic is put together from bytes per launch x launches per step (profiles/r06_step_B128*_summary.txt).
# Output: gpurun_out/r6_family_pmc.txt + .json (per family: measured read / write bytes per launch, algorithmic bytes, ratio, GB/s under the profiler).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R"/gpurun_out/fam
run() {   # tag, program args...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc $c -d "$R"/gpurun_out/fam/${tag}_$c -o p -- python3 "$@" > "$R"/gpurun_out/fam/${tag}_$c.log 2>&1
  done
}
ONLY="${FAMILY_ONLY:-all}"        # FAMILY_ONLY=attn: the attention programs only
if [ "$ONLY" = all ]; then
for f in ln_fwd ln_bwd gate_fwd gate_bwd loss adamw bn_apply bn_bwd slab_reduce; do run $f "$R"/tools/family_one.py $f 3; done
run gemm_nt_1024 "$R"/tools/gemm_one.py 147456 1024 1024 nt 3
run gemm_nt_gelu "$R"/tools/gemm_one.py 147456 4096 1024 gelu 3
run gemm_nt_k4096 "$R"/tools/gemm_one.py 147456 1024 4096 nt 3
run gemm_nt_lmhead "$R"/tools/gemm_one.py 147456 50304 1024 nt 2
run gemm_tn_1024 "$R"/tools/gemm_one.py 1024 1024 147456 tn 3
run gemm_tn_4096 "$R"/tools/gemm_one.py 4096 1024 147456 tn 3
fi
export ATTN_BENCH_B=128 ATTN_BENCH_PADS=1 ATTN_BENCH_MAPS=1
run attn_cross_text "$R"/tools/attn_bench.py cross_text
run attn_cross_img4 "$R"/tools/attn_bench.py cross_img4
run attn_self_causal "$R"/tools/attn_bench.py self_causal
cd "$R"
out=r6_family_pmc; [ "$ONLY" = all ] || out=r6_family_pmc_$ONLY
python tools/family_pmc_summary.py gpurun_out/fam $out > gpurun_out/$out.txt 2>&1
cat gpurun_out/$out.txt
rm -rf gpurun_out/fam/*_FETCH_SIZE gpurun_out/fam/*_WRITE_SIZE

cd $GRAFT_REPO_ROOT
EXP=$GRAFT_REPO_ROOT/tools/build/dmastag/libmmsum_hip.so
for rep in 1 2; do
  echo "== base $rep"; timeout 300 python tools/gemm_epi_bench.py 129024 2>&1 | grep -v amdgpu
  echo "== dma pieces staggered by wave parity $rep"; MMSUM_LIB=$EXP timeout 300 python tools/gemm_epi_bench.py 129024 2>&1 | grep -v amdgpu
done
MMSUM_LIB=$EXP timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -q -x -k "nt_" 2>&1 | tail -2

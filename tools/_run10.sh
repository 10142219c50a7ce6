cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/tools/build/prev/multimodalsum_amd/csrc/libmmsum_hip.so
for rep in 1 2; do
  for which in old new; do
    if [ $which = old ]; then export MMSUM_LIB=$OLD; else unset MMSUM_LIB; fi
    timeout 600 python bench.py --no-cpu-baseline --no-also --no-kernel-probe > gpurun_out/r03j_ab_${which}${rep}.json 2> gpurun_out/r03j_ab_${which}${rep}.err
    python -c "
import json,sys
d=json.loads(open('gpurun_out/r03j_ab_${which}${rep}.json').read().strip().splitlines()[-1]); print('${which}${rep}', round(d['value'],2), round(d['ms_per_step'],2))"
  done
done
unset MMSUM_LIB
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > gpurun_out/r03j_tests.log 2>&1; echo "rc $?" >> gpurun_out/r03j_tests.log
tail -3 gpurun_out/r03j_tests.log
ATTN_BENCH_B=112 ATTN_BENCH_PADS=1 timeout 600 python tools/attn_bench.py > gpurun_out/r03j_attn_new.txt 2>&1
MMSUM_LIB=$OLD ATTN_BENCH_B=112 ATTN_BENCH_PADS=1 timeout 600 python tools/attn_bench.py > gpurun_out/r03j_attn_old.txt 2>&1
paste -d'|' <(grep -v amdgpu gpurun_out/r03j_attn_old.txt | cut -c1-110) <(grep -v amdgpu gpurun_out/r03j_attn_new.txt | cut -c60-110) | head -40

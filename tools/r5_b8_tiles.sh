#!/usr/bin/env bash
# Small-batch tile rule: the B = 8 step with the shipped tile rule and with every product of <= 32768 rows forced to 256x128 / 128x128 tiles.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
F="--batch 8 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-probe --no-also"
for rep in 1 2; do
  for v in ship tile1 tile2; do
    if [ $v = ship ]; then unset MMSUM_LIB; else export MMSUM_LIB=tools/build/$v/libmmsum_hip.so; fi
    python bench.py $F > gpurun_out/b8_${v}_$rep.json 2> gpurun_out/b8_${v}_$rep.err
    echo "$v $rep $(python -c "import json; d=json.load(open('gpurun_out/b8_${v}_$rep.json')); print(round(d['value'],2), round(d['ms_per_step'],2))")"
  done
done
unset MMSUM_LIB
python tools/gemm_bench.py > gpurun_out/b8_gemm_ship.txt 2>&1; MMSUM_LIB=tools/build/tile1/libmmsum_hip.so python tools/gemm_bench.py > gpurun_out/b8_gemm_tile1.txt 2>&1; MMSUM_LIB=tools/build/tile2/libmmsum_hip.so python tools/gemm_bench.py > gpurun_out/b8_gemm_tile2.txt 2>&1
paste gpurun_out/b8_gemm_ship.txt gpurun_out/b8_gemm_tile1.txt gpurun_out/b8_gemm_tile2.txt | cut -c1-64,100-160,200-260 | grep NT

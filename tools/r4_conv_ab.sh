#!/usr/bin/env bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python tools/debug_ddp_rs.py > gpurun_out/r4h_debug_ddp.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv3x3 or batchnorm or conv_im2col" > gpurun_out/r4h_kernel_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4h_kernel_tests.log
timeout 1200 python -m pytest tests/test_modules_gpu.py tests/test_bench_shapes_gpu.py tests/test_timed_path_gpu.py -m gpu -q -x -k "multimodal_step or wide_step or full_depth or img_supervised" > gpurun_out/r4h_step_tests.log 2>&1; echo "rc $?" >> gpurun_out/r4h_step_tests.log
F="--steps 6 --warmup 2 --no-cpu-baseline --no-kernel-probe --no-also"
for rep in 1 2; do
  python bench.py $F > gpurun_out/r4h_implicit_$rep.json 2> gpurun_out/r4h_implicit_$rep.err
  MMSUM_IMPLICIT_CONV=fwd python bench.py $F > gpurun_out/r4h_fwdonly_$rep.json 2> gpurun_out/r4h_fwdonly_$rep.err
  MMSUM_IMPLICIT_CONV=0 python bench.py $F > gpurun_out/r4h_im2col_$rep.json 2> gpurun_out/r4h_im2col_$rep.err
done
grep -v "^\[\|RCCL\|HIP ver\|ROCm\|Hostname\|Librccl" gpurun_out/r4h_debug_ddp.txt | tail -24 | cut -c1-700; tail -12 gpurun_out/r4h_kernel_tests.log; tail -8 gpurun_out/r4h_step_tests.log
for f in gpurun_out/r4h_implicit_*.json gpurun_out/r4h_fwdonly_*.json gpurun_out/r4h_im2col_*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(round(d['value'],2), round(d['ms_per_step'],2))")"; done

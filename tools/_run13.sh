cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_timed_path_gpu.py -q -x -k "generation" -s 2>&1 | grep -v amdgpu | tail -4
timeout 300 python tools/power_probe.py 6 > gpurun_out/r03m_power_probe.txt 2>&1; grep -v amdgpu gpurun_out/r03m_power_probe.txt
timeout 300 python tools/blas_ref.py 129024 > gpurun_out/r03m_blas_ref.txt 2>&1; grep -v amdgpu gpurun_out/r03m_blas_ref.txt

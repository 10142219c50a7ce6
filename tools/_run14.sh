cd $GRAFT_REPO_ROOT
timeout 600 python tools/w4_stamps.py run 64512 1024 4096 > gpurun_out/r03n_stamps.txt 2>&1
timeout 600 python tools/w4_stamps.py run 64512 4096 1024 >> gpurun_out/r03n_stamps.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03n_stamps.txt | cut -c1-900

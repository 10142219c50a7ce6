cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -q -x -k "live_row" 2>&1 | grep -v amdgpu | tail -3

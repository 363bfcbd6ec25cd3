cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_golden.py tests/test_gpu_stream.py tests/test_gpu_dropin.py tests/test_gpu_pipes.py tests/test_gpu_records.py tests/test_gpu_split.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -15

cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_comm.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -5

cd $GRAFT_REPO_ROOT
F='^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version'
timeout 900 python -m pytest tests/test_gpu_split.py -m gpu -x -q 2>&1 | grep -v "$F" | tail -15

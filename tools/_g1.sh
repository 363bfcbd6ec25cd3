cd $GRAFT_REPO_ROOT
F='^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -15
timeout 600 python tools/family_bench.py 2>&1 | grep -v "$F" | grep "resample\|filter"

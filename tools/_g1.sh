cd $GRAFT_REPO_ROOT
F='^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version'
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -4
timeout 600 python tools/family_bench.py > gpurun_out/family_bench.txt 2>&1

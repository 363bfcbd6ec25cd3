cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -3
python tools/full_tiles_ab.py 2>&1 | grep "full tiles"
python tools/kbench.py 2>&1 | grep -E "2\^27"

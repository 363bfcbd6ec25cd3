cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -3
SDRHIP_SWEEP_SCALE=10 timeout 900 python -m pytest tests/test_gpu_pipes.py tests/test_gpu_records.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -2
SDRHIP_STREAM_SLOTS=3 timeout 900 python -m pytest tests/test_gpu_pipes.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -2
python tools/host_stream_native.py 2>&1 | grep "fir"

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_stream.py tests/test_gpu_chain.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -2
for i in 1 2 3; do for w in 1 0; do echo -n "TAP_PF=$w: "; SDRHIP_RESAMP_TAP_PF=$w python tools/stage_bench.py 26 2>&1 | grep -E "resample.*8192"; done; done

cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_examples.py tests/test_golden_haskell.py -m gpu -x -q 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version" | tail -3
python tools/host_stream_native.py 2>&1 | grep "sdrhip_fm_stream" | head -8
SDRHIP_STREAM_SLOTS=2 python tools/host_stream_native.py 2>&1 | grep "sdrhip_fm_stream" | head -4
SDRHIP_STREAM_SLOTS=3 python tools/host_stream_native.py 2>&1 | grep "sdrhip_fm_stream" | head -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

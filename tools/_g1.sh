cd $GRAFT_REPO_ROOT
F='^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|^RCCL\|^HIP version'
timeout 600 python tools/host_stream_native.py 2>&1 | grep -v "$F" > gpurun_out/host_stream.txt
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_err.txt
